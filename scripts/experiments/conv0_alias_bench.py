"""Timing-only ablation of conv0's forward (M = 8192, N = 256, k = 3, Cin = 4352): the same launch with the rows of A ALIASED (row
stride 0 / 64 / 512 channels instead of 4352) -- same instruction stream, same L2 -> LDS bytes, but A's footprint is 9 KB / 1 MB / 8 MB
instead of 71 MB.  If the launch is bound by how A comes out of DRAM / the fabric, these run at the loop's own rate."""
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
    for lda in (4352, 0, 64, 512, 1024):
        sets = []
        for b in range(NBUF):
            A = torch.randn(M, 4352, device=dev).to(torch.bfloat16)
            W = (torch.randn(N, 3 * Cin, device=dev) * 0.02).to(torch.bfloat16)
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            st = torch.empty(M // 128, 2, N, device=dev)
            sets.append(([ops.gemm_desc(A, W, C, M, N, Cin, taps=3, pad=1, Lout=T, Lsrc=T, lda=lda, stats=st)], (A, W, C, st)))
        try:
            for descs, _ in sets:
                ops.gemm_nt(descs, ops.BF16)
        except Exception as e:
            print("lda", lda, "refused:", str(e)[:100])
            continue
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
        print("split %d  row stride %4d channels  median %5.1f us  min %5.1f  -> %4.0f TF/s" % (ks, lda, ts[len(ts) // 2], ts[0], fl / ts[len(ts) // 2] / 1e6), flush=True)
