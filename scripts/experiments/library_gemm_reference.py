"""What do the vendor libraries reach on the two prop_fc products?  (Information only: the product path keeps its own kernels.)
forward: Y (8192 x 4096) = X (8192 x 4096) W^T (4096 x 4096); weight gradient: dW (4096 x 4096) = dZ^T (4096 x 8192) X (8192 x 4096).
Cold operands (a 1 GB buffer is touched between calls), 20 calls each, HIP events.  usage (GPU box): python scripts/experiments/library_gemm_reference.py"""
import torch
dev = torch.device("cuda:0")
big = torch.empty(1 << 28, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        big.add_(1.0); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        big.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


x = torch.randn(8192, 4096, device=dev).bfloat16()
w = torch.randn(4096, 4096, device=dev).bfloat16()
dzT = torch.randn(4096, 8192, device=dev).bfloat16()
xT = x.t().contiguous()
fl = 2.0 * 8192 * 4096 * 4096
for name, fn in (("forward  x @ w.T (bf16 out)", lambda: torch.mm(x, w.t())),
                 ("wgrad    dzT @ x (bf16 out)", lambda: torch.mm(dzT, x)),
                 ("wgrad NT dzT @ xT.T (bf16 out)", lambda: torch.mm(dzT, xT.t()))):
    us = timed(fn)
    print("%-34s %7.1f us  %6.0f TFLOP/s" % (name, us, fl / us / 1e6))
try:
    us = timed(lambda: torch.mm(dzT, x, out_dtype=torch.float32))
    print("%-34s %7.1f us  %6.0f TFLOP/s" % ("wgrad    dzT @ x (fp32 out)", us, fl / us / 1e6))
except Exception as e:
    print("fp32-out mm not available:", str(e).split("\n")[0])
