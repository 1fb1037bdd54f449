"""Stress the one-launch split-K exchange of drn_gemm_nt_splitk: many launches, workspace poisoned with NaN in between,
neighbouring kernels that leave ws lines in every XCD's L2; any NaN / mismatch in the output is a protocol failure."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import _lib, ops
dev = torch.device("cuda", 0)
L = _lib.lib()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
counters = torch.zeros(2048, dtype=torch.int32, device=dev)
bad = 0
for (M, N, K, ks, dt) in [(64, 512, 3072, 8, torch.float32), (4096, 512, 3072, 4, torch.bfloat16),
                          (8192, 256, 13056, 1, torch.bfloat16), (8192, 256, 13056, 2, torch.bfloat16), (8192, 256, 13056, 4, torch.bfloat16),
                          (8192, 256, 6528, 4, torch.bfloat16), (4096, 256, 13056, 4, torch.bfloat16), (8192, 256, 13056, 3, torch.bfloat16)]:
    code = 0 if dt == torch.float32 else 1
    A = torch.randn(M, K, device=dev).to(dt)
    W = torch.randn(N, K, device=dev).to(dt)
    C = torch.empty(M, N, device=dev, dtype=dt)
    d = ops.gemm_desc(A, W, C, M, N, K)
    arr = (_lib.GemmDesc * 1)(d)
    n_ws = int(L.drn_gemm_nt_splitk_ws_elems(M, N, ks))
    ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
    ref = None
    for it in range(60):
        ws.fill_(float("nan"))
        if it % 3 == 1:
            (ws * 1.0).sum()                      # plain loads of the poisoned workspace on every XCD
        _lib.check(L.drn_gemm_nt_splitk(arr, ks, ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(counters.data_ptr()), code, stream), "x")
        out = C.float().clone()
        if ref is None:
            ref = out
            want = A.float() @ W.float().t()
            err = float((out - want).abs().max() / want.abs().max())
            print("M=%d N=%d K=%d ks=%d %s: rel err vs torch %.2e" % (M, N, K, ks, dt, err))
        elif not torch.equal(out, ref) or not bool(torch.isfinite(out).all()):
            bad += 1
            if bad < 5:
                print("  MISMATCH at iteration", it, "nan count", int((~torch.isfinite(out)).sum()), "max diff", float((out - ref).abs().max()))
    print("  counters zero:", int(counters.abs().sum()) == 0)
print("bad launches:", bad)
