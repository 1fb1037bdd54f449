"""Per-wave timeline of the 128x128 NT kernel's K loop (one workgroup per CU regime).
Build:  cd drn_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDRN_NT_TRACE -c gemm_nt.hip -o /tmp/gemm_nt_trace.o \
        && hipcc --offload-arch=gfx950 -shared -fPIC api.o bn.o elementwise.o /tmp/gemm_nt_trace.o gemm_tn.o heads.o lgp.o loss.o \
           lstm.o optim.o postproc.o qenc.o skinny.o -o ../libdrn_hip_trace.so
Run:    DRN_LIB_PATH=drn_amd/libdrn_hip_trace.so python scripts/experiments/nt_trace.py
Stamps per K-step (s_memtime ticks): 0 loop top, 1 after vmcnt wait, 2 after barrier, 3 k-slice-0 fragments arrived,
4 k-slice-0 MFMAs + 2 pieces issued, 5 k-slice-1 fragments arrived, 6 k-slice-1 MFMAs + 2 pieces issued."""
import os
import sys
import ctypes
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib, check

dev = "cuda:0"
dt = torch.bfloat16
big = len(sys.argv) > 1 and sys.argv[1] == "prop_fc"       # the 256x256-tile prop_fc forward instead of a pyramid-level conv
B, L, N, Cin, taps = (32, 256, 4096, 4096, 1) if big else (32, 64, 512, 512, 3)
M = B * L
W = torch.randn(N, taps * Cin, device=dev).to(dt)
A = torch.randn(M, Cin, device=dev).to(dt)
C = torch.empty(M, N, device=dev, dtype=dt)
d = ops.gemm_desc(A, W, C, M, N, Cin, taps=taps, stride=1, pad=(taps - 1) // 2, Lout=L, Lsrc=L)
arr = (type(d) * 1)(d)
for stages in (("2",) if big else ("2", "4")):
    os.environ["DRN_NT_STAGES"] = stages
    trace = torch.zeros(8 * 64 * 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        check(lib().drn_gemm_nt_splitk(arr, 1, ctypes.c_void_p(trace.data_ptr()), ops.BF16, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "trace")
    torch.cuda.synchronize()
    t = trace.view(8, 64, 8).cpu()
    nk = min(64, taps * Cin // 64)
    print("stages=%s: %d K-steps; wave 0 deltas (ticks) per K-step: wait | barrier | frags0 | mfma0 | frags1 | mfma1 | total" % (stages, nk))
    for w in range(2):
        for k in range(4, min(nk, 14)):
            r = t[w, k]
            nxt = t[w, k + 1, 0] if k + 1 < nk else r[6]
            print("  w%d k%2d  %5d | %5d | %5d | %5d | %5d | %5d | %6d" % (w, k, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], nxt - r[0]))
    tot = (t[0, nk - 1, 6] - t[0, 0, 0]).item()
    print("  loop total %d ticks = %.1f per K-step" % (tot, tot / nk))
