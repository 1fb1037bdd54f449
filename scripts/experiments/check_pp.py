"""Ping-pong 256x256 NT kernel (DRN_NT_PP=1) vs the shipped 8-wave kernel on the same inputs: expected bit-identical (same
k-order of the fp32 accumulation).  Usage: python scripts/experiments/check_pp.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from drn_amd import ops

dev = "cuda:0"
os.environ["DRN_NT_TILE"] = "256"


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(dt, B, Lo, N, Cin, taps, stride, mode, stats, bias, gate, f32out=False, bench=False):
    torch.manual_seed(B * 1000 + Lo + N + Cin + taps)
    code = ops.dtype_code(torch.empty(1, dtype=dt))
    Ls = Lo * stride if mode == 0 else (Lo // stride)
    W = (torch.randn(N, taps * Cin, device=dev) * 0.05).to(dt)
    A = torch.randn(B * Ls, Cin, device=dev).to(dt)
    bias_t = torch.randn(N, device=dev) if bias else None
    gate_t = torch.rand(B, N, device=dev) if gate else None
    outs, times = {}, {}
    for pp in ("0", "1"):
        os.environ["DRN_NT_PP"] = pp
        C = torch.full((B * Lo, N), 7.0, device=dev, dtype=torch.float32 if f32out else dt)
        st = torch.zeros((B * Lo + 127) // 128, 2, N, device=dev) if stats else None
        d = ops.gemm_desc(A, W, C, B * Lo, N, Cin, taps=taps, stride=stride, pad=(taps - 1) // 2, mode=mode, Lout=Lo, Lsrc=Ls,
                          stats=st, bias=bias_t, gate=gate_t, ldg=N, out_f32=f32out)
        ops.gemm_nt([d], code)
        torch.cuda.synchronize()
        outs[pp] = (C.float().clone(), None if st is None else st.clone())
        if bench:
            times[pp] = timeit(lambda: ops.gemm_nt([d], code))
    same = torch.equal(outs["0"][0], outs["1"][0])
    err = (outs["0"][0] - outs["1"][0]).abs().max().item()
    s_ok = True if not stats else torch.allclose(outs["0"][1], outs["1"][1], rtol=1e-5, atol=1e-4)
    fl = 2.0 * B * Lo * N * taps * Cin
    extra = "" if not bench else "  8-wave %.3f ms (%.0f TF)  ping-pong %.3f ms (%.0f TF)" % (times["0"], fl / times["0"] / 1e9, times["1"], fl / times["1"] / 1e9)
    print("  %s identical=%s max|diff|=%.3e stats_ok=%s%s" % (str(dt).split(".")[-1], same, err, s_ok, extra))
    return same and s_ok


cases = [
    (torch.bfloat16, 32, 256, 4096, 4096, 1, 1, 0, False, True, True),
    (torch.bfloat16, 32, 256, 256, 4352, 3, 1, 0, True, False, False),
    (torch.bfloat16, 32, 256, 4352, 256, 3, 1, 1, False, False, False),
    (torch.bfloat16, 8, 128, 512, 256, 3, 2, 0, True, False, False),
    (torch.bfloat16, 8, 128, 256, 512, 3, 2, 1, False, False, False),
    (torch.bfloat16, 3, 100, 200, 64, 3, 1, 0, True, True, False),
    (torch.bfloat16, 5, 77, 136, 128, 1, 1, 0, False, True, True),
    (torch.float32, 4, 100, 200, 64, 3, 1, 0, True, True, False),
    (torch.float32, 8, 256, 512, 512, 1, 1, 0, False, True, True),
]
allok = True
for c in cases:
    print(c[1:])
    allok &= run(*c, bench=c[1] == 32)
print("f32-output (weight gradient as NT)")
allok &= run(torch.bfloat16, 1, 4096, 4096, 8192, 1, 1, 0, False, False, False, f32out=True, bench=True)
print("ALL OK" if allok else "MISMATCH")
