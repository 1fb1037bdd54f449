"""Does the row stride of the operands matter (L2 channel conflicts at power-of-two strides)?  Same products with lda = ldb = K
and K + pad.  usage: python scripts/experiments/sweep_w4_pad.py"""
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
modes = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 6]
for (M, N, K, f32out) in [(4096, 4096, 8192, True), (8192, 4096, 4096, False)]:
    for pad in (0, 32, 64, 128, 256):
        A = torch.randn(M + 8, K + pad, device=dev).to(bf)
        W = (torch.randn(N + 8, K + pad, device=dev) * 0.05).to(bf)
        C = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else bf)
        d = ops.gemm_desc(A, W, C, M, N, K, lda=K + pad, ldb=K + pad, out_f32=f32out)
        for mode in modes:
            check(lib().drn_tune(b"nt_w4", mode), "tune")
            us = timeit(lambda: ops.gemm_nt([d], ops.BF16))
            print("M=%5d N=%5d K=%5d f32out=%d  pad=%4d nt_w4=%2d  %7.1f us  %6.0f TFLOP/s" % (M, N, K, f32out, pad, mode, us, 2.0 * M * N * K / us / 1e6), flush=True)
check(lib().drn_tune(b"nt_w4", 0), "tune")
