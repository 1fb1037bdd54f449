import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops
dev = "cuda:0"; dt = torch.bfloat16; code = ops.BF16
def run(M, N, K, stats):
    A = torch.randn(M, K, device=dev).to(dt); W = torch.randn(N, K, device=dev).to(dt); C = torch.empty(M, N, device=dev, dtype=dt)
    st = torch.empty((M + 127) // 128, 2, N, device=dev) if stats else None
    d = ops.gemm_desc(A, W, C, M, N, K, Lout=M, stats=st)
    for _ in range(6): ops.gemm_nt([d], code)
    torch.cuda.synchronize()
for K in (64, 256, 1536):
    run(8192, 512, K, True)
    run(8192, 512, K, False)
    run(2048, 512, K, True)
