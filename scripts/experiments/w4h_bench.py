"""gemm_nt_w4h_kernel vs the 128 x 128 kernel on the grouped pyramid launches of the benchmarked shape (B = 32, T = 256), operands
rotated through enough buffers that no launch finds its inputs in the caches.  usage: python scripts/experiments/w4h_bench.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from drn_amd import ops, _lib

dev = torch.device("cuda:0")
L = _lib.lib()


def tune(k, v):
    _lib.check(L.drn_tune(k.encode(), int(v)), "tune")


SHAPES = [  # name, N, Cin per level, taps, mode, stats
    ("laterals fwd", 512, [256, 512, 1024], 1, 0, True),
    ("fpn layers fwd", 512, [512] * 3, 3, 0, True),
    ("mix_fc fwd", 512, [1024] * 3, 1, 0, True),
    ("towers dgrad", 512, [1024] * 3, 3, 1, False),
    ("fpn layers dgrad", 512, [512] * 3, 3, 1, False),
    ("laterals dgrad", None, [512] * 3, 1, 1, False),        # N = 256 / 512 / 1024
]
B, T = 32, 256
NBUF = 6
for name, N, cins, taps, mode, stats in SHAPES:
    sets = []
    for b in range(NBUF):
        descs, keep = [], []
        for l in range(3):
            Lq = T >> l
            M = B * Lq
            n = N if N is not None else (256, 512, 1024)[l]
            A = torch.randn(M, cins[l], device=dev).to(torch.bfloat16)
            W = (torch.randn(n, taps * cins[l], device=dev) * 0.05).to(torch.bfloat16)
            C = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
            st = torch.empty(M // 128, 2, n, device=dev) if stats else None
            descs.append(ops.gemm_desc(A, W, C, M, n, cins[l], taps=taps, pad=1 if taps == 3 else 0, mode=mode, Lout=Lq, Lsrc=Lq, stats=st))
            keep.append((A, W, C, st))
        sets.append((descs, keep))
    flops = sum(2.0 * d.M * d.N * d.taps * d.Cin for d in sets[0][0])
    res = {}
    for flag in (0, 128, 0, 128):
        tune("nt_w4h", flag)
        kind = ops.gemm_nt_plan(sets[0][0], ops.BF16)
        for descs, _ in sets:
            ops.gemm_nt(descs, ops.BF16)
        torch.cuda.synchronize()
        evs = []
        for rep in range(5):
            for descs, _ in sets:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm_nt(descs, ops.BF16)
                e1.record()
                evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        res.setdefault(flag, []).append((kind, ts[len(ts) // 2], ts[0]))
    print("%-18s %6.1f GFLOP | 128-tile kind %d: median %5.1f / %5.1f us (min %5.1f) | w4h kind %d: median %5.1f / %5.1f us (min %5.1f) -> %4.0f TF/s"
          % (name, flops / 1e9, res[0][0][0], res[0][0][1], res[0][1][1], res[0][0][2], res[128][0][0], res[128][0][1], res[128][1][1], res[128][0][2],
             flops / (res[128][1][1] * 1e-6) / 1e12), flush=True)
