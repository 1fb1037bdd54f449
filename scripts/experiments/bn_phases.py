"""Where does a one-launch conv -> BN -> ReLU block (drn_conv_bn_train) spend its time, next to the two launches it replaces?
Library built with -DDRN_NT_PHASES (see nt_phases.py).  Per workgroup, 100 MHz wall clock: entry / staging ready / first tile /
K loop done / statistics + raw tile published / column complete / statistics merged / normalised tile stored.
usage (GPU box): DRN_LIB_PATH=scripts/experiments/libdrn_hip_phases.so python scripts/experiments/bn_phases.py"""
import ctypes, os, sys
import numpy as np
import torch
import torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import functional as DF, ops
from drn_amd._lib import lib
dev = torch.device("cuda:0")
dt = torch.bfloat16
big = torch.empty(1 << 28, device=dev)


def blk(Cin, Cout, k):
    return nn.Conv1d(Cin, Cout, k, padding=(k - 1) // 2, bias=False).to(dev), nn.BatchNorm1d(Cout).to(dev)


def stamps(fn_name):
    nb = 4096
    buf = (ctypes.c_longlong * (nb * 8))()
    getattr(lib(), fn_name)(buf, nb * 8)
    t = np.array(buf, dtype=np.int64).reshape(nb, 8) * 10.0 / 1e3
    t = t[t[:, 0] > 0]
    return t[t[:, 0] >= t[:, 0].max() - 400.0] if len(t) else t


B = 32
Ls = (256, 128, 64)
cases = [("laterals (chain) N=512 K=256..1024", lambda xs: DF.multi_conv_block(xs, lat, True, dt, chain_up=True), (256, 512, 1024)),
         ("level convs N=512 K=1536", lambda xs: DF.multi_conv_block(xs, lvl, True, dt), (512, 512, 512)),
         ("towers N=1024 K=1536", lambda xs: DF.conv_block(xs, tw[0], tw[1], True, dt)[0], (512, 512, 512)),
         ("mix_fc N=512 K=1024", lambda xs: DF.conv_block(xs, mx[0], mx[1], True, dt)[0], (1024, 1024, 1024)),
         ("iou conv N=256 K=1536", lambda xs: DF.conv_block(xs, io[0], io[1], True, dt)[0], (512, 512, 512))]
lat = [blk(c, 512, 1) for c in (256, 512, 1024)]
lvl = [blk(512, 512, 3) for _ in range(3)]
tw, mx, io = blk(512, 1024, 3), blk(1024, 512, 1), blk(512, 256, 3)
with torch.no_grad():
    for name, fn, cins in cases:
        xs = [torch.randn(B, L, c, device=dev).to(dt) for L, c in zip(Ls, cins)]
        for fuse in (True, False):
            ops.BN_FUSE = fuse
            for _ in range(3):
                big.add_(1.0)                    # cold operands, as inside the step
                fn(xs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            big.add_(1.0)
            e0.record(); fn(xs); e1.record()
            torch.cuda.synchronize()
            t = stamps("drn_debug_nt_phases_bn" if fuse else "drn_debug_nt_phases")
            if not len(t):
                print("%-38s %-8s no stamps (the instrumented variant declined: events %.1f us)" % (name, "fused" if fuse else "2 launch", e0.elapsed_time(e1) * 1e3))
                continue
            t0 = t[:, 0].min()
            ncol = 8 if fuse else 5
            d = np.diff(t[:, :ncol], axis=1)
            names = ["prologue", "first tile", "K loop", "publish", "wait", "merge", "normalise"] if fuse else ["prologue", "first tile", "K loop", "epilogue"]
            print("%-38s %-8s %4d wgs  events %.1f us  kernel span %.1f us | K-loop end: first %.1f last %.1f | median per wg: %s" % (
                name, "fused" if fuse else "2 launch", len(t), e0.elapsed_time(e1) * 1e3, t[:, ncol - 1].max() - t0, t[:, 3].min() - t0, t[:, 3].max() - t0,
                ", ".join("%s %.2f" % (n, np.median(d[:, i])) for i, n in enumerate(names))))
            if fuse:
                print("%48s p90 per wg: %s | wait+merge+normalise after the LAST K-loop end: %.1f us" % (
                    "", ", ".join("%s %.2f" % (n, np.percentile(d[:, i], 90)) for i, n in enumerate(names)), t[:, 7].max() - t[:, 3].max()))
