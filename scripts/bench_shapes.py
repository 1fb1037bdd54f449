"""Tile-count / K-length sweep of the big NT GEMM (per-tile overhead vs main-loop speed)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops

dev = "cuda:0"
dt = torch.bfloat16
code = ops.dtype_code(torch.empty(1, dtype=dt))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (M, N, K, f32) in [(4096, 4096, 4096, 0), (4096, 4096, 8192, 0), (4096, 4096, 8192, 1), (8192, 4096, 4096, 0), (8192, 4096, 8192, 0),
                       (8192, 8192, 4096, 0), (4096, 4096, 16384, 0), (8192, 4352, 768, 0)]:
    W = torch.randn(N, K, device=dev).to(dt)
    A = torch.randn(M, K, device=dev).to(dt)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else dt)
    d = ops.gemm_desc(A, W, C, M, N, K, out_f32=bool(f32))
    ms = timeit(lambda: ops.gemm_nt([d], code))
    print("M=%5d N=%5d K=%5d f32out=%d tiles=%4d  %8.3f ms %8.1f TFLOP/s" % (M, N, K, f32, (M // 256) * ((N + 255) // 256), ms, 2.0 * M * N * K / ms / 1e9))
