"""128x128 vs 256x256 tiles of the NT kernel on the step's mid-size GEMM shapes, operands flushed out of the caches before
every launch (as inside the step).  Needs the experiments build (DRN_LIB_PATH=scripts/experiments/libdrn_exp.so, DRN_NT_TILE)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops
dev = torch.device("cuda", 0)
big = torch.empty(1 << 28, device=dev)
def timeit(fn, reps=20):
    def f():
        big.add_(1.0)
        fn()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t_flush = timeit(lambda: None)
for (M, N, K) in [(8192, 4352, 768), (14336, 1024, 1536), (14336, 512, 3072), (14336, 512, 1536), (8192, 4096, 4096)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    d = ops.gemm_desc(A, W, C, M, N, K)
    t = timeit(lambda: ops.gemm_nt([d], ops.BF16)) - t_flush
    print("tile %s  M=%5d N=%5d K=%5d  %7.1f us  %6.1f TFLOP/s" % (os.environ.get("DRN_NT_TILE", "auto"), M, N, K, t, 2.0 * M * N * K / t / 1e6))
