"""Would the mid-size k = 3 convolutions of the pyramid (112 tiles of 256x256: too few for one workgroup per CU) gain from
gemm_nt_w4c_kernel with a 2-way split into fp32 planes?  Single problems of their total size, cold operands.
usage: python scripts/experiments/splitk256_mid.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops, _lib
dev = torch.device("cuda", 0)
bf = torch.bfloat16
big = torch.empty(1 << 28, device=dev)
def timeit(fn, reps=15):
    for _ in range(3):
        big.add_(1.0); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        big.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, L, N, Cin, mode, stats) in [(56, 256, 512, 1024, 1, False), (56, 256, 512, 512, 0, True), (56, 256, 512, 512, 1, False)]:
    M = B * L
    X = torch.randn(M, Cin, device=dev).to(bf)
    W = (torch.randn(N, 3 * Cin, device=dev) * 0.02).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    st = torch.zeros(M // 128, 2, N, device=dev) if stats else None
    d = ops.gemm_desc(X, W, C, M, N, Cin, taps=3, pad=1, mode=mode, Lout=L, Lsrc=L, stats=st)
    arr = (_lib.GemmDesc * 1)(d)
    ws = torch.empty(4 * M * N, device=dev)
    print("M=%d N=%d K=%d mode=%d: shipped path %7.1f us" % (M, N, 3 * Cin, mode, timeit(lambda: ops.gemm_nt([d], ops.BF16))), flush=True)
    for ks in (2, 3):
        try:
            us = timeit(lambda: _lib.check(_lib.lib().drn_gemm_nt_splitk256(arr, ks, ctypes.c_void_p(ws.data_ptr()), ops.BF16, stream), "splitk256"))
            print("   splitk256 x%d %7.1f us" % (ks, us), flush=True)
        except Exception as e:
            print("   ksplit=%d: %s" % (ks, str(e)[:90]))
