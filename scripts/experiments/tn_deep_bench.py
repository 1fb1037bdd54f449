"""The per-tap weight-gradient kernel's row-block ring, 2-deep vs 4-deep (drn_tune "tn_deep"), on the step's three small launches:
laterals (multi, Cin 256 / 512 / 1024), conv2 (M = 2048, 1024 x 1536, stride 2), conv1 (M = 4096, 512 x 768, stride 2).  Cold operands."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib
dev = torch.device("cuda:0")
dt = torch.bfloat16
big = torch.empty(1 << 28, device=dev)
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
B = 32
lat = [(rnd(B * L, 512), rnd(B * L, c), B * L, L, c) for L, c in ((256, 256), (128, 512), (64, 1024))]
conv2 = (rnd(B * 64, 1024), rnd(B * 128, 512), B * 64, 64, 128)
conv1 = (rnd(B * 128, 512), rnd(B * 256, 256), B * 128, 128, 256)


def run_lat():
    dWs = [torch.empty(512, c, 1, device=dev) for _, _, _, _, c in lat]
    ops.gemm_wgrad_multi([ops.wgrad_desc(dy, x, M, Lout=L, Lsrc=L, ldy=512, ldx=c) for dy, x, M, L, c in lat], dWs, 512, [c for *_, c in lat],
                         taps=1, w_layout=1, dtype=ops.BF16)
    return dWs


def run_conv(case, N, Cin):
    dy, x, M, Lo, Ls = case
    dW = torch.empty(N, Cin, 3, device=dev)
    ops.gemm_wgrad([ops.wgrad_desc(dy, x, M, Lout=Lo, Lsrc=Ls, ldy=N, ldx=Cin)], dW, N, Cin, taps=3, stride=2, pad=1, w_layout=1, dtype=ops.BF16)
    return [dW]


cases = [("laterals wgrad (multi)", run_lat), ("conv2 wgrad", lambda: run_conv(conv2, 1024, 512)), ("conv1 wgrad", lambda: run_conv(conv1, 512, 256))]
for name, fn in cases:
    res = {}
    for deep in (0, 320, 0, 320):
        lib().drn_tune(b"tn_deep", deep)
        ts = []
        for _ in range(12):
            big.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res.setdefault(deep, []).append(ts[len(ts) // 2])
        ref = res.setdefault("out%d" % deep, [o.clone() for o in out])
    same = all(torch.equal(a, b) for a, b in zip(res["out0"], res["out320"]))
    print("%-24s 2-deep %s us   4-deep %s us   (launch + reduce, events)  bit-identical: %s" % (name, res[0], res[320], same), flush=True)
