"""Per-workgroup timeline of drn_skinny_group on the gate-projection shape (32 x 1024 -> 4096 + 256 + 512) from a library built with
-DDRN_QD_TRACE:  cd drn_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDRN_QD_TRACE -shared *.hip -o ../../scripts/experiments/libdrn_hip_trace.so
usage (GPU box): DRN_LIB_PATH=scripts/experiments/libdrn_hip_trace.so python scripts/experiments/qd_trace.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib
dev = torch.device("cuda:0")
shapes = {"gates": [(32, 1024, 4096), (32, 1024, 256), (32, 1024, 512)], "wih": [(64, 300, 4096)] * 4, "q0": [(32, 512, 3072)]}
for name, sh in shapes.items():
    probs = [dict(X=torch.randn(M, K, device=dev), W=torch.randn(N, K, device=dev), bias=torch.randn(N, device=dev)) for M, K, N in sh]
    for _ in range(5):
        ops.skinny_group(probs)
    torch.cuda.synchronize()
    nb = min(2048, int(os.environ.get("QD_NB", 0)) or sum((N + 15) // 16 for _, _, N in sh))
    buf = (ctypes.c_longlong * (nb * 4))()
    lib().drn_debug_qd_trace(buf, nb * 4)
    t = np.array(buf, dtype=np.int64).reshape(nb, 4) * 10.0 / 1e3          # us
    t0 = t[:, 0].min()
    print("%s: %d workgroups; launch span %.2f us; first start .. last start %.2f us; last end %.2f us" % (
        name, nb, t[:, 3].max() - t0, t[:, 0].max() - t0, t[:, 3].max() - t0))
    d = t - t[:, :1]
    print("   per workgroup (median / max): stamp0->1 %.2f / %.2f us, 1->2 %.2f / %.2f, 2->3 %.2f / %.2f" % (
        np.median(d[:, 1]), d[:, 1].max(), np.median(d[:, 2] - d[:, 1]), (d[:, 2] - d[:, 1]).max(), np.median(d[:, 3] - d[:, 2]), (d[:, 3] - d[:, 2]).max()))
    order = np.argsort(t[:, 0])
    print("   start times of every 32nd workgroup (us):", np.round(t[order[::32], 0] - t0, 2).tolist())
