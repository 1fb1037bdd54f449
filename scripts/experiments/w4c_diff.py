"""debug: where do gemm_nt_w4c_kernel<true> and the general kernel disagree?"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (B, L, N, Cin, bias, stats) in [(2, 128, 256, 128, True, True), (2, 128, 256, 128, False, True), (2, 128, 256, 128, True, False), (4, 64, 512, 192, False, True)]:
    M = B * L
    A = torch.randn(M, Cin, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, 3 * Cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias_t = torch.randn(N, generator=g).to(dev) if bias else None
    lib().drn_tune(b"exp0", 1)
    outs = []
    for flag in (0, 1):
        lib().drn_tune(b"nt_w4c", flag)
        C = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
        st = torch.full((M // 128, 2, N), float("nan"), device=dev) if stats else None
        d = ops.gemm_desc(A, W, C, M, N, Cin, taps=3, pad=1, mode=0, Lout=L, Lsrc=L, lda=Cin, bias=bias_t, stats=st)
        ops.gemm_nt([d], ops.BF16)
        torch.cuda.synchronize()
        outs.append((C.float().cpu(), None if st is None else st.cpu()))
    diff = (outs[0][0] - outs[1][0]).abs()
    bad = diff > 0
    print("case", (B, L, N, Cin, bias, stats), "bad elements", int(bad.sum()), "max", float(diff.max()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten().tolist()
        cols = bad.any(0).nonzero().flatten().tolist()
        print("  rows", rows[:40], "...", len(rows))
        print("  cols", cols[:40], "...", len(cols))
        r, c = rows[0], cols[0]
        print("  sample", outs[0][0][r, c:c + 8].tolist(), outs[1][0][r, c:c + 8].tolist())
    if stats:
        print("  stats max diff", float((outs[0][1] - outs[1][1]).abs().max()))
