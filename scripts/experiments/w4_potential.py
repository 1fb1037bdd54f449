"""Upper bound for moving the step's conv launches on 256x256 tiles to gemm_nt_w4_kernel: PLAIN products of the same M, N, K
(towers forward 14336 x 1024 x 1536 with BatchNorm-free epilogue, conv0 data gradient 8192 x 4096 x 768), nt_w4 = 0 / 1, cold."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib, check
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
for (M, N, K) in [(14336, 1024, 1536), (8192, 4096, 768), (14336, 512, 3072), (8192, 256, 13056)]:
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    d = ops.gemm_desc(A, W, C, M, N, K)
    for rnd in range(2):
        for mode in (0, 1):
            check(lib().drn_tune(b"nt_w4", mode), "tune")
            check(lib().drn_tune(b"exp0", 1), "tune")
            us = timeit(lambda: ops.gemm_nt([d], ops.BF16))
            print("M=%5d N=%5d K=%5d  nt_w4=%d  %7.1f us  %6.0f TFLOP/s" % (M, N, K, mode, us, 2.0 * M * N * K / us / 1e6), flush=True)
