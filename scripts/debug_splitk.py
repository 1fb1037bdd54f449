import os, sys, ctypes, torch, numpy as np
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops, _lib
dt = torch.float32 if sys.argv[1] == "f32" else torch.bfloat16
ksplit = int(sys.argv[2])
B, L, Cin, Cout, k = 3, 40, 192, 136, 3
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Cin, L, generator=g).to(dt); w = (torch.randn(Cout, Cin, k, generator=g) / 24).to(dt)
ref = F.conv1d(x.double(), w.double(), padding=1).permute(0, 2, 1).reshape(B * L, Cout)
M = B * L
xd = x.permute(0, 2, 1).contiguous().cuda(); wp = w.permute(0, 2, 1).contiguous().cuda()
C = torch.full((M, Cout), float("nan"), dtype=dt, device="cuda")
C2 = torch.full((M, Cout), float("nan"), dtype=dt, device="cuda")
bias = torch.randn(Cout, generator=g); gate = torch.randn(B, Cout, generator=g)
stats = torch.full(((M + 127) // 128, 2, Cout), float("nan"), device="cuda")
d = ops.gemm_desc(xd, wp, C, M, Cout, Cin, taps=k, pad=1, Lout=L, Lsrc=L, bias=bias.cuda(), gate=gate.cuda(), ldg=Cout, C2=C2, stats=stats)
ws = torch.full((ksplit * M * Cout,), float("nan"), dtype=torch.float32, device="cuda")
arr = (_lib.GemmDesc * 1)(d)
_lib.check(_lib.lib().drn_gemm_nt_splitk(arr, ksplit, ctypes.c_void_p(ws.data_ptr()), ops.dtype_code(xd), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "splitk")
torch.cuda.synchronize()
pre = ref + bias.double()
e2 = (C2.double().cpu() - pre).abs()
print("C2 max err", float(e2.max()), "rows", sorted(set((e2 > 1e-2).nonzero()[:, 0].tolist()))[:10], "cols", sorted(set((e2 > 1e-2).nonzero()[:, 1].tolist()))[:10])
print("C2 vs ref (no bias)", float((C2.double().cpu() - ref).abs().max()))
ref = pre * gate.double().repeat_interleave(L, 0)
err = (C.double().cpu() - ref).abs()
print("max err", float(err.max()), "nan in ws", int(torch.isnan(ws).sum()), "nan in C", int(torch.isnan(C).sum()))
bad = (err > 1e-2).nonzero()
print("bad rows", sorted(set(bad[:, 0].tolist()))[:20], "bad cols", sorted(set(bad[:, 1].tolist()))[:20], len(bad))
wsv = ws.view(ksplit, M, Cout).double().cpu()
print("sum-of-splits err", float((wsv.sum(0) - ref).abs().max()))
