"""Times the schedule / ablation variants of gemm_nt_w4_kernel (library built from `python scripts/gen_w4_loop.py --experiments`):
drn_tune nt_w4 = 0 (general 8-wave kernel), 1 + variant.  Cold operands.  usage: python scripts/experiments/sweep_w4.py [nvariants]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib, check
dev = torch.device("cuda", 0)
bf = torch.bfloat16
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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
for (B, Lo, N, K, f32out, full) in [(1, 4096, 4096, 8192, True, False), (32, 256, 4096, 4096, False, True)]:
    M = B * Lo
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else bf)
    C2 = torch.empty(M, N, device=dev, dtype=bf) if full else None
    bias_t = torch.randn(N, device=dev) if full else None
    gate_t = torch.rand(B, N, device=dev) if full else None
    d = ops.gemm_desc(A, W, C, M, N, K, Lout=Lo, Lsrc=Lo, bias=bias_t, gate=gate_t, ldg=N, C2=C2, out_f32=f32out)
    for rnd in range(2):
        for mode in range(0, nv + 1):
            check(lib().drn_tune(b"nt_w4", mode), "tune")
            us = timeit(lambda: ops.gemm_nt([d], ops.BF16))
            print("M=%5d N=%5d K=%5d f32out=%d  nt_w4=%2d  %7.1f us  %6.0f TFLOP/s" % (M, N, K, f32out, mode, us, 2.0 * M * N * K / us / 1e6), flush=True)
check(lib().drn_tune(b"nt_w4", 0), "tune")
