"""conv0's forward through drn_gemm_nt_splitk256 with different split counts.  6 = 3 taps x 2 channel halves: the splits {0,2,4} and
{1,3,5} of a tile then walk the SAME channel blocks of the same source rows at the same time (one tap apart), on one XCD (grid x =
tile, 32 tiles: XCD = x % 8 for every split) -- the input is read from HBM once instead of three times.
usage: python scripts/experiments/conv0_fwd_split_sweep.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops, _lib
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
C = torch.empty(M, N, device=dev, dtype=bf)
st = torch.zeros(M // 128, 2, N, device=dev)
d = ops.gemm_desc(X, W, C, M, N, Cin, taps=3, pad=1, Lout=L, Lsrc=L, stats=st)
arr = (_lib.GemmDesc * 1)(d)
ws = torch.empty(12 * M * N, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ops.SPLITK256 = False
print("shipped (128x128 tiles, in-launch 2-way split)  %7.1f us" % timeit(lambda: ops.gemm_nt([d], ops.BF16)), flush=True)
ref = C.float().clone()
for rnd in range(2):
    for ks in (3, 4, 6, 8, 12):
        try:
            us = timeit(lambda: _lib.check(_lib.lib().drn_gemm_nt_splitk256(arr, ks, ctypes.c_void_p(ws.data_ptr()), ops.BF16, stream), "splitk256"))
        except Exception as e:
            print("ksplit=%d: %s" % (ks, str(e)[:100]))
            continue
        print("ksplit=%2d  %7.1f us   max|dC| %.3e" % (ks, us, (C.float() - ref).abs().max().item()), flush=True)
