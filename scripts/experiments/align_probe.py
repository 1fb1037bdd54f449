"""Does the speed of the 67-134 MB GEMMs depend on WHERE their operands sit?  (round 6: two builds of the library whose only difference was a
store instruction in another kernel ran prop_fc's forward at 183 and 197 us -- the allocations had moved by 2 MB.)
One arena, operands carved at chosen absolute alignments, prop_fc's forward shape (8192 x 4096 x 4096, gate + pre-gate copy) timed."""
import itertools
import sys

import torch

sys.path.insert(0, ".")
from drn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
MB = 1 << 20
arena = torch.empty(6 << 30, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
print("arena base %x (mod 1 GiB: %d MiB)" % (base, (base % (1 << 30)) // MB))
top = (base + (1 << 30) - 1) // (1 << 30) * (1 << 30) - base          # first 1 GiB boundary inside the arena


def carve(off, shape, dtype):
    n = 1
    for s in shape:
        n *= s
    nbytes = n * torch.empty(0, dtype=dtype).element_size()
    return arena[off:off + nbytes].view(dtype).view(*shape)


M, N, K, T = 8192, 4096, 4096, 256
g = torch.Generator(device="cpu").manual_seed(0)


def run(offA, offB, offC, offC2, iters=30):
    A = carve(top + offA, (M, K), torch.bfloat16)
    B = carve(top + offB, (N, K), torch.bfloat16)
    C = carve(top + offC, (M, N), torch.bfloat16)
    C2 = carve(top + offC2, (M, N), torch.bfloat16)
    A.copy_(torch.randn(M, K, generator=g).to(torch.bfloat16))
    B.copy_((torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16))
    bias = torch.zeros(N, device=dev)
    gate = torch.rand(M // T, N, device=dev)
    d = ops.gemm_desc(A, B, C, M, N, K, Lout=T, bias=bias, gate=gate, ldg=N, C2=C2)
    for _ in range(3):
        ops.gemm_nt([d], ops.BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm_nt([d], ops.BF16)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# slots 256 MiB apart; each operand shifted inside its slot
S = 256 * MB
print("all on 256 MiB boundaries: %.1f us" % run(0, S, 2 * S, 3 * S))
for sh in (2 * MB, 4 * MB, 8 * MB, 16 * MB, 32 * MB, 64 * MB, 128 * MB):
    print("all shifted by %3d MiB: %.1f us" % (sh // MB, run(sh, S + sh, 2 * S + sh, 3 * S + sh)))
print("-- one operand shifted by 2 MiB, the rest on 256 MiB boundaries")
for i, name in enumerate(("A", "B", "C", "C2")):
    o = [0, S, 2 * S, 3 * S]
    o[i] += 2 * MB
    print("  %-2s +2 MiB: %.1f us" % (name, run(*o)))
print("-- packed back to back (as an allocator would), start shifted")
for sh in (0, 2 * MB, 4 * MB, 6 * MB, 64 * 1024, 1 * MB):
    a = sh
    b = a + 64 * MB
    c = b + 32 * MB
    c2 = c + 64 * MB
    print("  start +%-8d: %.1f us" % (sh, run(a, b, c, c2)))
print("-- small shifts of everything (bytes)")
for sh in (256, 4096, 65536, 262144):
    print("  +%-7d: %.1f us" % (sh, run(sh, S + sh, 2 * S + sh, 3 * S + sh)))
