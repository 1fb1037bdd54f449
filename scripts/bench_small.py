"""Why do the query side's small kernels take 10-18 us inside the step?  Times drn_qe_attn_fwd (32 workgroups, one memory round
trip) three ways: hot in a loop, after a 1 GB cache-flushing pass, and with its operands carved out of ONE allocation instead of
eight separate ones.  usage (GPU box): python scripts/bench_small.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops
dev = torch.device("cuda", 0)
B, L, C = 32, 8, 1024


def timeit(fn, reps=50, pre=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if pre is not None:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


def mk(alloc):
    out = alloc(B * L * C).view(B, L, C).normal_()
    qcmd = alloc(B * 3 * C).view(B, 3 * C).normal_()
    w = alloc(C).normal_()
    bias = alloc(4)[:1].zero_()
    att = alloc(B * 3 * L).view(B, 3, L)
    cmds = alloc(3 * B * C).view(3, B, C)
    return out, qcmd, w, bias, att, cmds


lens = torch.randint(3, L + 1, (B,), device=dev, dtype=torch.int64)
big = torch.empty(1 << 28, device=dev)
sep = mk(lambda n: torch.empty(n, device=dev))
pool = torch.empty(1 << 20, device=dev)
off = [0]


def carve(n):
    n4 = (n + 3) // 4 * 4
    t = pool[off[0]:off[0] + n]
    off[0] += n4
    return t


one = mk(carve)
for name, (out, qcmd, w, bias, att, cmds) in (("separate allocations", sep), ("one allocation", one)):
    f = lambda: ops.qe_attn_fwd(out, qcmd, w, bias, lens, att, cmds, B, L, C)
    print("%-22s hot %.1f us   after a 1 GB flush %.1f us   (event pair alone %.1f us)" % (
        name, timeit(f), timeit(f, pre=lambda: big.add_(1.0)), timeit(lambda: None)))
