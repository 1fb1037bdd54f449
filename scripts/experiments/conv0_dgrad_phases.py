"""conv0's data gradient with the gate backward in its epilogue (gemm_nt_w4c_kernel<false>, DrnGemmDesc::gb_*) at the benchmarked shape:
per-workgroup phases.  Library built with -DDRN_NT_PHASES (see w4_phases.py)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib
dev = torch.device("cuda:0")
big = torch.empty(1 << 28, device=dev)
g = torch.Generator(device="cuda").manual_seed(0)
B, L, Cout, D, P = 32, 256, 256, 4096, 256
M, Cin = B * L, D + P
dY = torch.randn(M, Cout, device=dev, generator=g).to(torch.bfloat16)
Wd = (torch.randn(D, 3 * Cout, device=dev, generator=g) * 0.05).to(torch.bfloat16)
Z = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
gate = torch.rand(B, D, device=dev, generator=g) + 0.5
dx = torch.zeros(M, Cin, device=dev, dtype=torch.bfloat16)
dZT = torch.empty(D, M, device=dev, dtype=torch.bfloat16)
dgate, dsum = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
for gb in (True, False):
    kw = dict(gate=gate, ldg=D, gate_bwd=dict(act=Z, ld_act=D, dct=dZT, ldt=M, dgate=dgate, dsum=dsum)) if gb else {}
    d = ops.gemm_desc(dY, Wd, dx, M, D, Cout, taps=3, pad=1, mode=1, Lout=L, Lsrc=L, ldc=Cin, **kw)
    for _ in range(3):
        big.add_(1.0)
        ops.gemm_nt([d], ops.BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big.add_(1.0)
    e0.record(); ops.gemm_nt([d], ops.BF16); e1.record()
    torch.cuda.synchronize()
    nb = 4096
    buf = (ctypes.c_longlong * (nb * 8))()
    lib().drn_debug_nt_phases_w4(buf, nb * 8)
    t = np.array(buf, dtype=np.int64).reshape(nb, 8)[:, :5] * 10.0 / 1e3
    t = t[t[:, 0] > 0]
    t = t[t[:, 0] >= t[:, 0].max() - 300.0]
    t0 = t[:, 0].min()
    d_ = np.diff(t, axis=1)
    q = lambda x: "%.1f/%.1f/%.1f" % (np.percentile(x, 10), np.median(x), np.percentile(x, 90))
    print("conv0 dgrad %-22s %4d wgs | events %.1f us | span %.1f | prologue %s  K loop %s  (stamp) %s  epilogue+drain %s" % (
        "+ gate backward (gb_*)" if gb else "plain (swapped tiles)", len(t), e0.elapsed_time(e1) * 1e3, t[:, 4].max() - t0, q(d_[:, 0]), q(d_[:, 1]), q(d_[:, 2]), q(d_[:, 3])), flush=True)
