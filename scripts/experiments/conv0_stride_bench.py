"""conv0's forward (M = 8192, N = 256, k = 3, Cin = 4352) against the ROW STRIDE of its input: 4352 channels = 8704 bytes = 68 cache
lines, a multiple of 4 -- every row of a 128-byte K-slice starts in the same quarter of the L2 channels?  usage: python conv0_stride_bench.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from drn_amd import ops

dev = torch.device("cuda:0")
B, T, Cin, N = 32, 256, 4352, 256
M = B * T
NBUF = 4
for ks in (0, 1):
    ops.KSPLIT_W4H = bool(ks)
    for lda in (4352, 4352 + 64, 4352 + 128, 4352 + 192, 4352 + 256, 4352 + 32):
        sets = []
        for b in range(NBUF):
            A = torch.randn(M, lda, device=dev).to(torch.bfloat16)
            W = (torch.randn(N, 3 * Cin, device=dev) * 0.02).to(torch.bfloat16)
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            st = torch.empty(M // 128, 2, N, device=dev)
            sets.append(([ops.gemm_desc(A, W, C, M, N, Cin, taps=3, pad=1, Lout=T, Lsrc=T, lda=lda, stats=st)], (A, W, C, st)))
        for descs, _ in sets:
            ops.gemm_nt(descs, ops.BF16)
        torch.cuda.synchronize()
        evs = []
        for rep in range(6):
            for descs, _ in sets:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.gemm_nt(descs, ops.BF16); e1.record()
                evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        fl = 2.0 * M * N * 3 * Cin
        print("w4h split %d  lda %4d (%3d lines)  median %5.1f us  min %5.1f  -> %4.0f TF/s" % (ks, lda, lda * 2 // 128, ts[len(ts) // 2], ts[0], fl / ts[len(ts) // 2] / 1e6), flush=True)
