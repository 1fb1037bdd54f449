"""gemm_nt_w4_kernel (drn_tune nt_w4=1) against the general 8-wave 256x256 kernel (nt_w4=0): bit for bit, then timing with the
operands flushed out of the caches before every launch (as inside the step).  usage (GPU box): python scripts/experiments/check_w4.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops
from drn_amd._lib import lib, check
dev = torch.device("cuda", 0)
bf = torch.bfloat16


def tune(v):
    check(lib().drn_tune(b"nt_w4", int(v)), "tune")
    check(lib().drn_tune(b"exp0", 1), "tune")       # 256x256 tiles from one big tile on (small cases too)


def case(B, Lo, N, K, bias, gate, c2, f32out, reps=1):
    torch.manual_seed(B * 131 + Lo + N + K)
    M = B * Lo
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    bias_t = torch.randn(N, device=dev) if bias else None
    gate_t = torch.rand(B, N, device=dev) if gate else None
    outs = []
    for mode in (0, 1):
        tune(mode)
        for _ in range(reps):
            C = torch.full((M, N), 7.0, device=dev, dtype=torch.float32 if f32out else bf)
            C2 = torch.full((M, N), 5.0, device=dev, dtype=bf) if c2 else None
            d = ops.gemm_desc(A, W, C, M, N, K, Lout=Lo, Lsrc=Lo, bias=bias_t, gate=gate_t, ldg=N, C2=C2, out_f32=f32out)
            ops.gemm_nt([d], ops.BF16)
            torch.cuda.synchronize()
            outs.append((C.clone(), None if C2 is None else C2.clone()))
    ref = outs[0]
    ok = True
    for o in outs[1:]:
        ok &= torch.equal(ref[0], o[0]) and (ref[1] is None or torch.equal(ref[1], o[1]))
    fin = torch.isfinite(ref[0].float()).all().item()
    print("M=%5d N=%5d K=%5d bias=%d gate=%d C2=%d f32out=%d  identical=%s finite=%s" % (M, N, K, bias, gate, c2, f32out, ok, fin))
    return ok


allok = True
allok &= case(1, 256, 256, 128, False, False, False, False)
allok &= case(1, 256, 256, 256, True, False, False, False)
allok &= case(2, 256, 512, 512, True, True, True, False)
allok &= case(4, 256, 768, 1152, True, True, False, False)
allok &= case(1, 512, 256, 640, False, False, False, True)
allok &= case(32, 256, 4096, 4096, True, True, True, False, reps=3)
allok &= case(1, 4096, 4096, 8192, False, False, False, True, reps=3)
print("ALL OK" if allok else "MISMATCH")

big = torch.empty(1 << 28, device=dev)
def timeit(fn, reps=20):
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

for (B, Lo, N, K, f32out, full) in [(32, 256, 4096, 4096, False, True), (1, 4096, 4096, 8192, True, False), (1, 4096, 4096, 8192, False, False), (32, 256, 4096, 4096, False, False)]:
    M = B * Lo
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else bf)
    C2 = torch.empty(M, N, device=dev, dtype=bf) if full else None
    bias_t = torch.randn(N, device=dev) if full else None
    gate_t = torch.rand(B, N, device=dev) if full else None
    d = ops.gemm_desc(A, W, C, M, N, K, Lout=Lo, Lsrc=Lo, bias=bias_t, gate=gate_t, ldg=N, C2=C2, out_f32=f32out)
    for rnd in range(2):
        for mode in (0, 1):
            tune(mode)
            us = timeit(lambda: ops.gemm_nt([d], ops.BF16))
            print("M=%5d N=%5d K=%5d f32out=%d epilogue=%s  nt_w4=%d  %7.1f us  %6.0f TFLOP/s" % (M, N, K, f32out, "bias+gate+C2" if full else "plain", mode, us, 2.0 * M * N * K / us / 1e6))
