"""conv0's forward (M = 8192, N = 256, K = 3 x 4352: gemm_nt_w4h_kernel<true>, split-K 4 inside the launch, interleaved taps): phases per
workgroup row (blockIdx.y == 0 only is stamped).  Library built with -DDRN_NT_PHASES (scripts/experiments/build_phases.sh)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib
dev = torch.device("cuda:0")
big = torch.empty(1 << 28, device=dev)
g = torch.Generator(device="cuda").manual_seed(0)
B, L, Cin, N = 32, 256, 4352, 256
M = B * L
A = torch.randn(M, Cin, device=dev, generator=g).to(torch.bfloat16)
W = (torch.randn(N, 3 * Cin, device=dev, generator=g) * 0.02).to(torch.bfloat16)
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
st = torch.empty(M // 128, 2, N, device=dev)
d = ops.gemm_desc(A, W, C, M, N, Cin, taps=3, pad=1, Lout=L, Lsrc=L, stats=st)
print("ksplit", ops._ksplit_w4h([d], ops.BF16), flush=True)
for _ in range(3):
    big.add_(1.0)
    ops.gemm_nt([d], ops.BF16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
big.add_(1.0)
e0.record(); ops.gemm_nt([d], ops.BF16); e1.record()
torch.cuda.synchronize()
nb = 4096
buf = (ctypes.c_longlong * (nb * 8))()
lib().drn_debug_nt_phases_w4h(buf, nb * 8)
t = np.array(buf, dtype=np.int64).reshape(nb, 8)[:, :5] * 10.0 / 1e3
t = t[t[:, 0] > 0]
t = t[t[:, 0] >= t[:, 0].max() - 300.0]
t0 = t[:, 0].min()
fin = t[t[:, 4] > t[:, 2]]                      # workgroups that ran the epilogue (the last arrivers among the stamped row)
q = lambda x: "%.1f/%.1f/%.1f" % (np.percentile(x, 10), np.median(x), np.percentile(x, 90))
print("conv0 forward: events %.1f us | %d stamped wgs (split row 0) | prologue %s  K loop %s | last loop end %.1f" % (
    e0.elapsed_time(e1) * 1e3, len(t), q(t[:, 1] - t[:, 0]), q(t[:, 2] - t[:, 1]), t[:, 2].max() - t0))
if len(fin):
    print("  last arrivers among them: %d | exchange (loop end -> epilogue start) %s  epilogue + drain %s | last exit %.1f" % (
        len(fin), q(fin[:, 3] - fin[:, 2]), q(fin[:, 4] - fin[:, 3]), fin[:, 4].max() - t0))
