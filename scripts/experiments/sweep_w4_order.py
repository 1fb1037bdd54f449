"""Tile order (drn_tune exp3 = xcd_swizzle + 1: 1 none, 2 XCD-contiguous, 3 8-row groups only, 4 both = shipped) for the two
prop_fc products on gemm_nt_w4_kernel, cold operands.  usage: python scripts/experiments/sweep_w4_order.py"""
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
for (M, N, K, f32out) in [(4096, 4096, 8192, True), (8192, 4096, 4096, False)]:
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else bf)
    d = ops.gemm_desc(A, W, C, M, N, K, out_f32=f32out)
    for rnd in range(2):
        for order in (1, 2, 3, 4):
            check(lib().drn_tune(b"exp3", order), "tune")
            us = timeit(lambda: ops.gemm_nt([d], ops.BF16))
            print("M=%5d N=%5d K=%5d f32out=%d  exp3=%d  %7.1f us  %6.0f TFLOP/s" % (M, N, K, f32out, order, us, 2.0 * M * N * K / us / 1e6), flush=True)
check(lib().drn_tune(b"exp3", 0), "tune")
