"""Where does the fixed part of a small NT GEMM launch go?  Library built with -DDRN_NT_PHASES:
  cd drn_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDRN_NT_PHASES -shared *.hip -o ../../scripts/experiments/libdrn_hip_phases.so
usage (GPU box): DRN_LIB_PATH=scripts/experiments/libdrn_hip_phases.so python scripts/experiments/nt_phases.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib
dev = torch.device("cuda:0")
dt = torch.bfloat16
big = torch.empty(1 << 28, device=dev)
# (name, [(clips, L)], N, Cin, taps, stride)
shapes = [("conv1 fwd  M=4096 N=512 K=768", [(32, 128)], 512, 256, 3, 1), ("laterals-like M=8192 N=512 K=256", [(32, 256)], 512, 256, 1, 1),
          ("level convs g=3 N=512 K=1536", [(32, 256), (32, 128), (32, 64)], 512, 512, 3, 1),
          ("towers g=3 N=1024 K=1536", [(32, 256), (32, 128), (32, 64)], 1024, 512, 3, 1), ("prop_fc M=8192 N=4096 K=4096", [(32, 256)], 4096, 4096, 1, 1)]
for name, levels, N, Cin, taps, stride in shapes:
    W = torch.randn(N, taps * Cin, device=dev).to(dt)
    descs, keep = [], []
    for b, L in levels:
        A = torch.randn(b * L, Cin, device=dev).to(dt)
        C = torch.empty(b * L, N, device=dev, dtype=dt)
        st = torch.empty(((b * L + 127) // 128, 2, N), device=dev, dtype=torch.float32) if taps > 1 or Cin < 4096 else None   # conv blocks: BN statistics in the epilogue
        descs.append(ops.gemm_desc(A, W, C, b * L, N, Cin, taps=taps, pad=(taps - 1) // 2, Lout=L, Lsrc=L, stats=st))
        keep.append((A, C, st))
    for cold in (False, True):
        for _ in range(3):
            if cold:
                big.add_(1.0)
            ops.gemm_nt(descs, ops.BF16)
        torch.cuda.synchronize()
        nb = 4096
        buf = (ctypes.c_longlong * (nb * 8))()
        lib().drn_debug_nt_phases(buf, nb * 8)
        t = np.array(buf, dtype=np.int64).reshape(nb, 8)[:, :5] * 10.0 / 1e3
        t = t[t[:, 0] > 0]
        t = t[t[:, 0] >= t[:, 0].max() - 400.0]              # the last launch only (earlier launches left older stamps)
        t0 = t[:, 0].min()
        d = np.diff(t, axis=1)
        print("%-36s %-5s %4d wgs: span %.1f us | median per workgroup: prologue %.2f, first tile %.2f, K loop %.2f, epilogue %.2f | start spread %.2f" % (
            name, "cold" if cold else "hot", len(t), t[:, 4].max() - t0, np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(d[:, 3]),
            t[:, 0].max() - t0))
