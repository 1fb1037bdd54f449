"""Isolated timing of the two transposing passes of the input stage at the benchmarked shape (B*T = 8192, D = 4096, bf16)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops
dev = torch.device("cuda", 0)
def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, T, D, P = 32, 256, 4096, 256
x = torch.randn(B * T, D, device=dev)
big = torch.empty(1 << 28, device=dev)      # 1 GB: flush the caches between calls
def flush(fn):
    def f():
        big.add_(1.0)
        fn()
    return f
t_flush = timeit(lambda: big.add_(1.0))
t = timeit(flush(lambda: ops.cast_transpose(x, ops.BF16))) - t_flush
print("cast_transpose  %.1f us  %.2f TB/s (268 MB)" % (t, 268.4e6 / t / 1e6))
dG = torch.randn(B, T, D + P, device=dev).to(torch.bfloat16)
Z = torch.randn(B, T, D, device=dev).to(torch.bfloat16)
gate = torch.randn(B, D, device=dev)
dZT = torch.empty(D, B * T, device=dev, dtype=torch.bfloat16)
dgate, dsum = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
t = timeit(flush(lambda: ops.gate_bwd_t(dG, D + P, Z, D, gate, dZT, dgate, B, T, D, ops.BF16, dsum=dsum))) - t_flush
print("gate_bwd_t      %.1f us  %.2f TB/s (201 MB)" % (t, 201.3e6 / t / 1e6))
