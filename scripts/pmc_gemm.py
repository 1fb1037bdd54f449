"""One-shape GEMM launches for PMC collection: python scripts/pmc_gemm.py <shape> (prop_fc | l3 | towers)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops
dev = "cuda:0"; dt = torch.bfloat16; code = ops.BF16
shape = sys.argv[1]
B = 32
def mk(levels, N, Cin, taps=1):
    W = torch.randn(N, taps * Cin, device=dev).to(dt); descs = []; keep = []
    for (b, L) in levels:
        A = torch.randn(b * L, Cin, device=dev).to(dt); C = torch.empty(b * L, N, device=dev, dtype=dt)
        descs.append(ops.gemm_desc(A, W, C, b * L, N, Cin, taps=taps, pad=(taps - 1) // 2, Lout=L, Lsrc=L)); keep.append((A, C))
    return descs, keep, W
if shape == "wgrad":
    dY = torch.randn(B * 256, 4096, device=dev).to(dt); X = torch.randn(B * 256, 4096, device=dev).to(dt)
    dW = torch.empty(4096, 4096, 1, device=dev)
    for _ in range(5):
        ops.gemm_wgrad([ops.wgrad_desc(dY, X, B * 256)], dW, 4096, 4096, dtype=code)
    torch.cuda.synchronize(); sys.exit()
if shape == "wgrad_nt":      # the prop_fc weight gradient as it runs in the step: NT product of the K-major copies, fp32 output
    dZT = torch.randn(4096, B * 256, device=dev).to(dt); xT = torch.randn(4096, B * 256, device=dev).to(dt)
    dW = torch.empty(4096, 4096, device=dev)
    for _ in range(5):
        ops.gemm_nt([ops.gemm_desc(dZT, xT, dW, 4096, 4096, B * 256, out_f32=True)], code)
    torch.cuda.synchronize(); sys.exit()
if shape == "wgrad3":        # conv0 weight gradient through the fused 3-tap kernel
    dY = torch.randn(B * 256, 256, device=dev).to(dt); X = torch.randn(B * 256, 4352, device=dev).to(dt)
    dW = torch.empty(256, 4352, 3, device=dev)
    for _ in range(5):
        ops.gemm_wgrad([ops.wgrad_desc(dY, X, B * 256, Lout=256, Lsrc=256)], dW, 256, 4352, taps=3, stride=1, pad=1, w_layout=1, dtype=code)
    torch.cuda.synchronize(); sys.exit()
if shape == "prop_fc": d, k, W = mk([(B, 256)], 4096, 4096)
elif shape == "l3": d, k, W = mk([(B, 64)], 512, 512, 3)
else: d, k, W = mk([(B, 256), (B, 128), (B, 64)], 1024, 512, 3)
for _ in range(5):
    ops.gemm_nt(d, code)
torch.cuda.synchronize()
