"""Where do the five forward GEMM launches of FPN + heads (w4h / w4c / general kernel) spend their time?  Per workgroup, 100 MHz wall
clock: entry / staging state ready / K loop done / tile stores issued / statistics written + stores drained.
Library built with -DDRN_NT_PHASES (every .hip):
  cd drn_amd/csrc && for f in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDRN_NT_PHASES -c $f -o /tmp/ph/${f%.hip}.o; done
  hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ph/*.o -o scripts/experiments/libdrn_hip_phases.so
usage (GPU box): DRN_LIB_PATH=scripts/experiments/libdrn_hip_phases.so python scripts/experiments/w4_phases.py"""
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


def stamps():
    best = None
    for fn in ("drn_debug_nt_phases_w4h", "drn_debug_nt_phases_w4", "drn_debug_nt_phases"):
        nb = 4096
        buf = (ctypes.c_longlong * (nb * 8))()
        getattr(lib(), fn)(buf, nb * 8)
        t = np.array(buf, dtype=np.int64).reshape(nb, 8)[:, :5] * 10.0 / 1e3
        t = t[t[:, 0] > 0]
        if len(t) and (best is None or t[:, 0].max() > best[1][:, 0].max()):
            best = (fn, t)
    fn, t = best
    return fn, t[t[:, 0] >= t[:, 0].max() - 300.0]


B = 32
Ls = (256, 128, 64)
lat = [blk(c, 512, 1) for c in (256, 512, 1024)]
lvl = [blk(512, 512, 3) for _ in range(3)]
tw, mx, io = blk(512, 1024, 3), blk(1024, 512, 1), blk(512, 256, 3)
cases = [("laterals (chain) N=512 K=256..1024", lambda xs: DF.multi_conv_block(xs, lat, True, dt, chain_up=True), (256, 512, 1024)),
         ("level convs N=512 K=1536", lambda xs: DF.multi_conv_block(xs, lvl, True, dt), (512, 512, 512)),
         ("towers N=1024 K=1536", lambda xs: DF.conv_block(xs, tw[0], tw[1], True, dt)[0], (512, 512, 512)),
         ("mix_fc N=512 K=1024", lambda xs: DF.conv_block(xs, mx[0], mx[1], True, dt)[0], (1024, 1024, 1024)),
         ("iou conv N=256 K=1536", lambda xs: DF.conv_block(xs, io[0], io[1], True, dt)[0], (512, 512, 512))]
ops.BN_FUSE = False
with torch.no_grad():
    for name, fn, cins in cases:
        xs = [torch.randn(B, L, c, device=dev).to(dt) for L, c in zip(Ls, cins)]
        for cold in (True, False):
            for _ in range(3):
                if cold:
                    big.add_(1.0)
                fn(xs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if cold:
                big.add_(1.0)
            e0.record(); fn(xs); e1.record()
            torch.cuda.synchronize()
            table, t = stamps()
            t0 = t[:, 0].min()
            d = np.diff(t, axis=1)
            q = lambda x: "%.1f/%.1f/%.1f" % (np.percentile(x, 10), np.median(x), np.percentile(x, 90))
            print("%-36s %-4s %-8s %4d wgs | GEMM+BN events %.1f us | GEMM span %.1f | start spread %.1f | 10/50/90 %%: prologue %s  K loop %s  "
                  "tile stores %s  stats+drain %s | last loop end %.1f, last exit %.1f" % (
                      name, "cold" if cold else "hot", table.replace("drn_debug_nt_phases", "tbl"), len(t), e0.elapsed_time(e1) * 1e3,
                      t[:, 4].max() - t0, t[:, 0].max() - t0, q(d[:, 0]), q(d[:, 1]), q(d[:, 2]), q(d[:, 3]), t[:, 2].max() - t0,
                      t[:, 4].max() - t0), flush=True)

# cycle stamps of workgroup 0 / thread 0 inside the LAST w4h launch's epilogue (s_memtime; -DDRN_NT_PHASES)
if hasattr(lib(), "drn_debug_epi_cyc"):
    with torch.no_grad():
        xs = [torch.randn(B, L, 512, device=dev).to(dt) for L in Ls]
        for _ in range(3):
            big.add_(1.0)
            DF.multi_conv_block(xs, lvl, True, dt)
        torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib().drn_debug_epi_cyc(buf, 64)
    c = np.array(buf, dtype=np.int64)
    print("epilogue cycles (level convs, wg 0): bias %d | chunks (acc read, convert+store): %s | inside last chunk: transposes+ds_write->%d lds wait %d "
          "read+global stores %d wait %d | stats: colsum %d, lds+mean %d, colsq %d, tail %d | total %d" % (
              c[1] - c[0] if False else 0, ", ".join("(%d, %d)" % (c[1 + 2 * k] - (c[0] if k == 0 else c[2 * k]), c[2 + 2 * k] - c[1 + 2 * k]) for k in range(4)),
              c[10] - c[7], c[11] - c[10], c[12] - c[11], c[13] - c[12], c[21] - c[20], c[22] - c[21], c[23] - c[22], c[24] - c[23], c[24] - c[0]))
# cycle stamps (s_memtime) of workgroup 0 / thread 0 inside the LAST w4h launch's epilogue
if hasattr(lib(), "drn_debug_epi_cyc"):
    with torch.no_grad():
        xs = [torch.randn(B, L, 512, device=dev).to(dt) for L in Ls]
        for _ in range(3):
            big.add_(1.0)
            DF.multi_conv_block(xs, lvl, True, dt)
        torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    lib().drn_debug_epi_cyc(buf, 64)
    c = np.array(buf, dtype=np.int64)
    print("epilogue counts (level convs, wg 0 wave 0): column sums + means %d | store chunks with the square pass behind them %d | row sums + stat stores %d | drain %d | total %d" % (
        c[1] - c[0], c[3] - c[1], c[4] - c[3], c[5] - c[4], c[5] - c[0]))
