"""Fused 3-tap weight-gradient kernel vs the per-tap TN kernel: agreement and time on the DRN conv shapes."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops

dev = "cuda:0"
dt = torch.bfloat16
code = ops.dtype_code(torch.empty(1, dtype=dt))


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


def case(name, levels, N, Cin, layout=1, multi=False, bench=True):
    torch.manual_seed(0)
    descs, keep, flops = [], [], 0
    for (B, L) in levels:
        dY = torch.randn(B * L, N, device=dev).to(dt)
        X = torch.randn(B * L, Cin, device=dev).to(dt)
        descs.append(ops.wgrad_desc(dY, X, B * L, Lout=L, Lsrc=L))
        keep.append((dY, X))
        flops += 2.0 * B * L * N * 3 * Cin
    res = {}
    for fused in ("0", "1"):
        os.environ["DRN_TN_FUSED"] = fused
        if multi:
            dWs = [torch.zeros(N, Cin, 3, device=dev) for _ in levels]
            run = lambda: ops.gemm_wgrad_multi(descs, dWs, N, Cin, taps=3, stride=1, pad=1, w_layout=layout, dtype=code)
        else:
            dW = torch.zeros(N, Cin, 3, device=dev)
            dWs = [dW]
            run = lambda: ops.gemm_wgrad(descs, dW, N, Cin, taps=3, stride=1, pad=1, w_layout=layout, dtype=code)
        run()
        torch.cuda.synchronize()
        ms = timeit(run) if bench else 0.0
        res[fused] = ([w.clone() for w in dWs], ms)
    err = max((a - b).abs().max().item() for a, b in zip(res["0"][0], res["1"][0]))
    ref = max(a.abs().max().item() for a in res["0"][0])
    print("%-34s per-tap %7.3f ms  fused %7.3f ms (%6.1f TFLOP/s)  max|diff| %.3e (max|ref| %.2e) identical=%s" % (
        name, res["0"][1], res["1"][1], flops / max(res["1"][1], 1e-9) / 1e9, err, ref,
        all(torch.equal(a, b) for a, b in zip(res["0"][0], res["1"][0]))))


B = 32
os.environ["DRN_TN3_MINROWS"] = "0"
case("tiny 2x20 N=24 Cin=40", [(2, 20)], 24, 40, bench=False)
case("ragged 3x33 N=136 Cin=200", [(3, 33)], 136, 200, layout=0, bench=False)
case("grouped 4x(64,32,16)", [(4, 64), (4, 32), (4, 16)], 128, 128, bench=False)
case("multi 4x(64,32,16)", [(4, 64), (4, 32), (4, 16)], 128, 128, multi=True, bench=False)
case("conv0 wgrad 256x13056 r8192", [(B, 256)], 256, 4352)
case("towers wgrad 1024x1536 r14336", [(B, 256), (B, 128), (B, 64)], 1024, 512)
case("fpn layers (multi) 512x1536", [(B, 256), (B, 128), (B, 64)], 512, 512, multi=True)
case("layer L3 wgrad 512x1536 r2048", [(B, 64)], 512, 512)
