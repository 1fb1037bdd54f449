"""gemm_nt_w4c_kernel (drn_tune nt_w4c=1) against the general 8-wave 256x256 kernel (nt_w4c=0) on k = 3 / stride 1 convolutions,
forward (mode 0) and data gradient (mode 1), with bias / BatchNorm statistics / gate: bit for bit; then timing, cold operands.
usage (GPU box): python scripts/experiments/check_w4c.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib, check
dev = torch.device("cuda", 0)
bf = torch.bfloat16


def tune(v):
    check(lib().drn_tune(b"nt_w4c", int(v)), "tune")
    check(lib().drn_tune(b"exp0", 1), "tune")       # 256x256 tiles from one big tile on


def case(B, L, N, Cin, mode, bias, stats, gate, lda_pad=0):
    torch.manual_seed(B * 131 + L + N + Cin + mode)
    M = B * L
    A = torch.randn(M, Cin + lda_pad, device=dev).to(bf)
    W = (torch.randn(N, 3 * Cin, device=dev) * 0.05).to(bf)
    bias_t = torch.randn(N, device=dev) if bias else None
    gate_t = torch.rand(B, N, device=dev) if gate else None
    outs = []
    for flag in (0, 1, 1):
        tune(flag)
        C = torch.full((M, N), 7.0, device=dev, dtype=bf)
        st = torch.full((M // 128, 2, N), float("nan"), device=dev) if stats else None
        d = ops.gemm_desc(A, W, C, M, N, Cin, taps=3, pad=1, mode=mode, Lout=L, Lsrc=L, lda=Cin + lda_pad, bias=bias_t, gate=gate_t, ldg=N, stats=st)
        ops.gemm_nt([d], ops.BF16)
        torch.cuda.synchronize()
        outs.append((C, st))
    ok = all(torch.equal(outs[0][0], o[0]) and (not stats or torch.equal(outs[0][1], o[1])) for o in outs[1:])
    fin = torch.isfinite(outs[0][0].float()).all().item()
    d = (outs[0][0].float() - outs[1][0].float()).abs().max().item()
    print("B=%2d L=%4d N=%5d Cin=%5d mode=%d bias=%d stats=%d gate=%d pad=%d  identical=%s finite=%s max|d|=%.3e" % (B, L, N, Cin, mode, bias, stats, gate, lda_pad, ok, fin, d), flush=True)
    return ok


allok = True
allok &= case(1, 256, 256, 64, 0, False, False, False)
allok &= case(2, 128, 256, 128, 0, True, True, False)
allok &= case(4, 64, 512, 192, 0, False, True, False)
allok &= case(2, 256, 256, 256, 1, False, False, False)
allok &= case(8, 32, 256, 128, 1, True, False, True)
allok &= case(2, 128, 256, 128, 0, True, False, True, lda_pad=64)
allok &= case(32, 256, 1024, 512, 0, False, True, False)
allok &= case(32, 256, 4096, 256, 1, False, False, False)
print("ALL OK" if allok else "MISMATCH")

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

for (B, L, N, Cin, mode, stats) in [(32, 256, 1024, 512, 0, True), (32, 256, 4096, 256, 1, False), (56, 256, 1024, 512, 0, True)]:
    M = B * L
    A = torch.randn(M, Cin, device=dev).to(bf)
    W = (torch.randn(N, 3 * Cin, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    st = torch.zeros(M // 128, 2, N, device=dev) if stats else None
    d = ops.gemm_desc(A, W, C, M, N, Cin, taps=3, pad=1, mode=mode, Lout=L, Lsrc=L, stats=st)
    for rnd in range(2):
        for flag in (0, 1):
            tune(flag)
            us = timeit(lambda: ops.gemm_nt([d], ops.BF16))
            print("M=%5d N=%5d K=%5d mode=%d stats=%d  nt_w4c=%d  %7.1f us  %6.0f TFLOP/s" % (M, N, 3 * Cin, mode, stats, flag, us, 2.0 * M * N * 3 * Cin / us / 1e6), flush=True)
