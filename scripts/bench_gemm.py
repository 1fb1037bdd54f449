"""Micro-benchmark of the MFMA GEMM kernels on the DRN shapes (B=32, T=256, D=4096): TFLOP/s per launch."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import ops

dev = "cuda:0"
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
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


def nt(name, levels, N, Cin, taps=1, stride=1, mode=0, stats=False):
    """levels: list of (B, Lout); mode 0: A rows are inputs (B, Lout*stride, Cin)."""
    descs, keep, flops = [], [], 0
    W = torch.randn(N, taps * Cin, device=dev).to(dt)
    for (B, Lo) in levels:
        Ls = Lo * stride if mode == 0 else (Lo // stride)
        A = torch.randn(B * Ls, Cin, device=dev).to(dt)
        C = torch.empty(B * Lo, N, device=dev, dtype=dt)
        st = torch.empty((B * Lo + 127) // 128, 2, N, device=dev) if stats else None
        descs.append(ops.gemm_desc(A, W, C, B * Lo, N, Cin, taps=taps, stride=stride, pad=(taps - 1) // 2, mode=mode,
                                   Lout=Lo, Lsrc=Ls, stats=st))
        keep.append((A, C, st))
        flops += 2.0 * B * Lo * N * taps * Cin
    ms = timeit(lambda: ops.gemm_nt(descs, code))
    print("%-34s %8.3f ms %8.1f TFLOP/s" % (name, ms, flops / ms / 1e9))


def wg(name, levels, N, Cin, taps=1, stride=1):
    descs, keep, flops = [], [], 0
    dW = torch.empty(N, Cin, taps, device=dev)
    for (B, Lo) in levels:
        Ls = Lo * stride
        dY = torch.randn(B * Lo, N, device=dev).to(dt)
        X = torch.randn(B * Ls, Cin, device=dev).to(dt)
        descs.append(ops.wgrad_desc(dY, X, B * Lo, Lout=Lo, Lsrc=Ls))
        keep.append((dY, X))
        flops += 2.0 * B * Lo * N * taps * Cin
    ms = timeit(lambda: ops.gemm_wgrad(descs, dW, N, Cin, taps=taps, stride=stride, pad=(taps - 1) // 2, w_layout=1, dtype=code))
    print("%-34s %8.3f ms %8.1f TFLOP/s" % (name, ms, flops / ms / 1e9))


B = 32
print("DRN_NT_STAGES=%s dtype=%s" % (os.environ.get("DRN_NT_STAGES", "auto"), dt))
nt("prop_fc fwd 8192x4096x4096", [(B, 256)], 4096, 4096)
nt("wgrad-as-NT 4096x4096x8192", [(1, 4096)], 4096, 8192)
nt("conv0 fwd 8192x256x13056", [(B, 256)], 256, 4352, taps=3, stats=True)
nt("conv0 dgrad 8192x4352x768", [(B, 256)], 4352, 256, taps=3, mode=1)
nt("conv1 fwd s2 4096x512x768", [(B, 128)], 512, 256, taps=3, stride=2, stats=True)
nt("conv2 fwd s2 2048x1024x1536", [(B, 64)], 1024, 512, taps=3, stride=2, stats=True)
nt("fpn layer L1 8192x512x1536", [(B, 256)], 512, 512, taps=3, stats=True)
nt("fpn layer L3 2048x512x1536", [(B, 64)], 512, 512, taps=3, stats=True)
nt("fpn inner1 8192x512x256", [(B, 256)], 512, 256, stats=True)
nt("towers fwd 14336x1024x1536", [(B, 256), (B, 128), (B, 64)], 1024, 512, taps=3, stats=True)
nt("towers dgrad 14336x512x3072", [(B, 256), (B, 128), (B, 64)], 512, 1024, taps=3, mode=1)
nt("conv2 dgrad 4096x512x3072", [(B, 128)], 512, 1024, taps=3, stride=2, mode=1)
if "wgrad" in sys.argv:
    wg("prop_fc wgrad 4096x4096 r8192", [(B, 256)], 4096, 4096)
    wg("conv0 wgrad 256x13056 r8192", [(B, 256)], 256, 4352, taps=3)
    wg("towers wgrad 1024x1536 r14336", [(B, 256), (B, 128), (B, 64)], 1024, 512, taps=3)
    wg("layer L3 wgrad 512x1536 r2048", [(B, 64)], 512, 512, taps=3)
