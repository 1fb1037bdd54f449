"""Is the 256 x 256-tile GEMM's speed set by what the chip did in the milliseconds before?  prop_fc's forward shape, timed per launch with
HIP events: back to back, after idle gaps, and after a stretch of a bandwidth-bound kernel (what the step does between its large GEMMs)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from drn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M, N, K, T = 8192, 4096, 4096, 256
g = torch.Generator(device="cpu").manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
B = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).to(dev)
C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
C2 = torch.empty_like(C)
bias = torch.zeros(N, device=dev)
gate = torch.rand(M // T, N, device=dev)
d = ops.gemm_desc(A, B, C, M, N, K, Lout=T, bias=bias, gate=gate, ldg=N, C2=C2)
big = torch.empty(256 << 20, dtype=torch.float32, device=dev)          # 1 GiB: a copy of it is ~0.35 ms of pure HBM traffic
big2 = torch.empty_like(big)


def one():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm_nt([d], ops.BF16)
    e1.record()
    return e0, e1


def series(n, between):
    ev = []
    for _ in range(n):
        between()
        ev.append(one())
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0], t[-1]


for _ in range(5):
    ops.gemm_nt([d], ops.BF16)
torch.cuda.synchronize()
print("back to back          : median %.1f  min %.1f  max %.1f us" % series(40, lambda: None))
for gap in (0.2, 0.5, 1, 2, 5, 20):
    def idle(gap=gap):
        torch.cuda.synchronize()
        time.sleep(gap / 1e3)
    print("after %4.1f ms idle     : median %.1f  min %.1f  max %.1f us" % ((gap,) + series(25, idle)))
for n in (1, 2, 4):
    def copies(n=n):
        for _ in range(n):
            big2.copy_(big)
    print("after %d x 1 GiB copy   : median %.1f  min %.1f  max %.1f us" % ((n,) + series(25, copies)))
print("back to back again    : median %.1f  min %.1f  max %.1f us" % series(40, lambda: None))
