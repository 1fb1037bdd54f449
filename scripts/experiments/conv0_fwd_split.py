"""conv0's forward (8192 x 256 x 13056, k = 3, BatchNorm statistics) through ops.gemm_nt: 128x128 tiles with the in-launch 2-way
split (ops.SPLITK256 = False) against gemm_nt_w4c_kernel on 256x256 tiles with 8 fp32 partial planes + the adding launch
(drn_gemm_nt_splitk256, default), cold operands.  usage: python scripts/experiments/conv0_fwd_split.py
(The same split on the 8-wave 256x256 kernel -- scripts/experiments/gemm_nt_splitk256.patch -- measured 94 us against 84-88.)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
dev = torch.device("cuda", 0)
bf = torch.bfloat16
B, L, Cin, N = 32, 256, 4352, 256
M = B * L
X = torch.randn(M, Cin, device=dev).to(bf)
W = (torch.randn(N, 3 * Cin, device=dev) * 0.02).to(bf)
big = torch.empty(1 << 28, device=dev)
def timeit(fn, reps=15):
    for _ in range(3):
        big.add_(1.0); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        big.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
res = {}
for mode in (False, True, False, True):
    ops.SPLITK256 = mode
    C = torch.empty(M, N, device=dev, dtype=bf)
    st = torch.zeros(M // 128, 2, N, device=dev)
    d = ops.gemm_desc(X, W, C, M, N, Cin, taps=3, pad=1, Lout=L, Lsrc=L, stats=st)
    us = timeit(lambda: ops.gemm_nt([d], ops.BF16))
    res[mode] = (C.float().clone(), st.clone())
    print("splitk256=%d  %7.1f us  %6.0f TFLOP/s" % (mode, us, 2.0 * M * N * 3 * Cin / us / 1e6), flush=True)
a, b = res[False], res[True]
print("max |dC| %.3e (scale %.2f)   max rel d(sum) %.3e" % ((a[0] - b[0]).abs().max().item(), a[0].abs().max().item(),
      ((a[1][:, 0] - b[1][:, 0]).abs().max() / a[1][:, 0].abs().max()).item()))
