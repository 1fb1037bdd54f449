"""conv0's forward (8192 x 256 x 13056, k = 3, split 4 ways inside gemm_nt_w4h_kernel) timed alone: exchange confirmation off / on,
interleaved-tap walk on / off.  usage: python scripts/experiments/conv0_confirm_bench.py"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from drn_amd import ops, _lib
dev = torch.device("cuda:0")
B, L, Cin, N = 32, 256, 4352, 256
M = B * L
g = torch.Generator().manual_seed(0)
A = torch.randn(M, Cin, generator=g).to(torch.bfloat16).to(dev)
W = (torch.randn(N, 3 * Cin, generator=g) * 0.02).to(torch.bfloat16).to(dev)
C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
d = ops.gemm_desc(A, W, C, M, N, Cin, taps=3, pad=1, Lout=L, Lsrc=L)
print("ksplit", ops._ksplit_w4h([d], ops.BF16))
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timed(n=40):
    ts = []
    for _ in range(n):
        flush.zero_()                                   # cold operands, as inside the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_nt([d], ops.BF16)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for rnd in range(2):
    for tapil in (2048, 0):
        _lib.lib().drn_tune(b"w4h_tapil", tapil)
        for conf in (("0", "1", "2") if os.environ.get("BASE_LIB") != "1" else ("base",)):
            if conf != "base":
                ops.XCHG_CONFIRM = conf                 # (round 6: the mode travels in the call's `ksplit`, drn_amd.ops._ksplit_arg)
            print("round %d tapil %4d confirm %s: %.1f us" % (rnd, tapil, conf, timed()), flush=True)
