"""The clock the 256 x 256-tile GEMM's K loop runs at (library built with -DDRN_NT_PHASES: bash scripts/experiments/build_phases.sh).
Per workgroup the kernel stamps the 100 MHz wall clock AND s_memtime (shader cycles) at both ends of its loop statement: cycles / time = the
effective clock, cycles / K-step = how far the loop is from the MFMA issue floor (128 MFMAs x 16 cycles = 2048 per K-step and SIMD).
prop_fc's forward shape (8192 x 4096 x 4096), operands: randn / zeros; back to back, and after 1 ms of idle.
usage (GPU box): DRN_LIB_PATH=$PWD/scripts/experiments/libdrn_hip_phases.so python scripts/experiments/gemm_clock.py"""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from drn_amd import ops  # noqa: E402
from drn_amd._lib import lib  # noqa: E402

dev = torch.device("cuda", 0)
M, N, K, T = 8192, 4096, 4096, 256
g = torch.Generator(device="cpu").manual_seed(0)
C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
C2 = torch.empty_like(C)
bias = torch.zeros(N, device=dev)
gate = torch.rand(M // T, N, device=dev)


def stamps(table="drn_debug_nt_phases_w4"):
    nb = 4096
    buf = (ctypes.c_longlong * (nb * 8))()
    getattr(lib(), table)(buf, nb * 8)
    t = np.array(buf, dtype=np.int64).reshape(nb, 8)
    t = t[t[:, 1] > 0]
    t = t[t[:, 1] >= t[:, 1].max() - 50000]          # the last launch (within 500 us)
    wall_ns = (t[:, 2] - t[:, 1]) * 10.0
    cyc = (t[:, 6] - t[:, 5]).astype(np.float64)
    return wall_ns, cyc


def case(name, A, B, before, n=12):
    d = ops.gemm_desc(A, B, C, M, N, K, Lout=T, bias=bias, gate=gate, ldg=N, C2=C2)
    for _ in range(3):
        ops.gemm_nt([d], ops.BF16)
    torch.cuda.synchronize()
    rows = []
    for _ in range(n):
        before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_nt([d], ops.BF16)
        e1.record()
        torch.cuda.synchronize()
        wall, cyc = stamps()
        rows.append((e0.elapsed_time(e1) * 1e3, np.median(wall) / 1e3, np.median(cyc / wall), np.median(cyc) / (K // 64), len(wall)))
    r = np.median(np.array(rows), axis=0)
    print("%-34s launch %.1f us | K loop %.1f us per workgroup (first of two rounds dominate) at %.2f GHz, %.0f cycles per K-step (floor 2048) | %d workgroups stamped"
          % (name, r[0], r[1], r[2], r[3], int(r[4])), flush=True)


def burn():                       # keep the chip busy with the same GEMM right before (the back-to-back state)
    for _ in range(6):
        ops.gemm_nt([ops.gemm_desc(Ar, Br, C, M, N, K, Lout=T, bias=bias, gate=gate, ldg=N, C2=C2)], ops.BF16)


def idle():
    torch.cuda.synchronize()
    time.sleep(1e-3)


Ar = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
Br = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).to(dev)
Az, Bz = torch.zeros_like(Ar), torch.zeros_like(Br)
case("randn operands, back to back", Ar, Br, burn)
case("randn operands, after 1 ms idle", Ar, Br, idle)
case("zero operands, back to back", Az, Bz, burn)
case("zero operands, after 1 ms idle", Az, Bz, idle)


# the convolutions of FPN + heads on the 4-wave kernels (towers: 256 x 256 tiles, k = 3; level convs: 256 x 128 tiles)
import torch.nn as nn  # noqa: E402
from drn_amd import functional as DF  # noqa: E402

ops.BN_FUSE = False


def blk(Cin, Cout, k):
    return nn.Conv1d(Cin, Cout, k, padding=(k - 1) // 2, bias=False).to(dev), nn.BatchNorm1d(Cout).to(dev)


tw = blk(512, 1024, 3)
lvl = [blk(512, 512, 3) for _ in range(3)]
xs = [torch.randn(32, L, 512, device=dev).to(torch.bfloat16) for L in (256, 128, 64)]
big = torch.empty(1 << 28, device=dev)
with torch.no_grad():
    for name, fn, table, floor in (("towers (w4c, 256 x 256, K = 1536)", lambda: DF.conv_block(xs, tw[0], tw[1], True, torch.bfloat16), "drn_debug_nt_phases_w4", 2048),
                                   ("level convs (w4h, 256 x 128, K = 1536)", lambda: DF.multi_conv_block(xs, lvl, True, torch.bfloat16), "drn_debug_nt_phases_w4h", 1024)):
        rows = []
        for _ in range(8):
            big.add_(1.0)                     # (cold operands, a bandwidth-bound kernel right before: the step's situation)
            fn()
            torch.cuda.synchronize()
            wall, cyc = stamps(table)
            rows.append((np.median(wall) / 1e3, np.median(cyc / wall), np.median(cyc) / 24, len(wall)))
        r = np.median(np.array(rows), axis=0)
        print("%-40s K loop %.1f us at %.2f GHz, %.0f cycles per K-step (floor %d) | %d workgroups stamped" % (name, r[0], r[1], r[2], floor, int(r[3])), flush=True)
