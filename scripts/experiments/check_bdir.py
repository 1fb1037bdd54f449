"""B-operand-from-registers variant of the 256x256 NT tile (drn_tune exp1) against the shipped kernel: bit-for-bit and timing with
the operands flushed out of the caches before every launch (as inside the step)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib, check
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
for (M, N, K, f32out) in [(8192, 4096, 4096, False), (4096, 4096, 8192, True), (8192, 4096, 768, False), (14336, 1024, 1536, False)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    outs, ts = [], []
    for mode in (0, 1, 0, 1):
        check(lib().drn_tune(b"exp1", mode), "tune")
        C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
        d = ops.gemm_desc(A, W, C, M, N, K, out_f32=f32out)
        t = timeit(lambda: ops.gemm_nt([d], ops.BF16)) - t_flush
        outs.append(C)
        ts.append(t)
    print("M=%5d N=%5d K=%5d  shipped %7.1f / %7.1f us   B-direct %7.1f / %7.1f us  (%.0f -> %.0f TFLOP/s)  bit-identical: %s" % (
        M, N, K, ts[0], ts[2], ts[1], ts[3], 2.0 * M * N * K / min(ts[0], ts[2]) / 1e6, 2.0 * M * N * K / min(ts[1], ts[3]) / 1e6,
        bool(torch.equal(outs[0], outs[1]))))
