"""Isolated timing of the under-filled pyramid-level NT GEMMs (128 output tiles on 256 CUs) for split-K factors 1..8
(drn_gemm_nt_splitk: partial tiles summed by the last-arriving split, one launch)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd import _lib, ops

dev = torch.device("cuda", 0)
dt = torch.bfloat16
L = _lib.lib()
B = 32
# name, Lsrc, Lout, Cin, Cout, taps, stride, mode
shapes = [("conv0 fwd", 256, 256, 4352, 256, 3, 1, 0), ("conv1 fwd", 256, 128, 256, 512, 3, 2, 0),
          ("conv2 fwd", 128, 64, 512, 1024, 3, 2, 0), ("conv2 dgrad", 64, 128, 1024, 512, 3, 2, 1),
          ("conv1 dgrad", 128, 256, 512, 256, 3, 2, 1)]
counters = torch.zeros(2048, dtype=torch.int32, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, Lsrc, Lout, Cin, Cout, taps, stride, mode in shapes:
    A = torch.randn(B, Lsrc, Cin, device=dev).to(dt)
    W = torch.randn(Cout, taps * Cin, device=dev).to(dt)
    M = B * Lout
    C = torch.empty(M, Cout, device=dev, dtype=dt)
    stats = torch.empty((M + 127) // 128, 2, Cout, device=dev) if mode == 0 else None
    d = ops.gemm_desc(A, W, C, M, Cout, Cin, taps=taps, stride=stride, pad=1, mode=mode, Lout=Lout, Lsrc=Lsrc, stats=stats)
    arr = (_lib.GemmDesc * 1)(d)
    res = []
    for ks in (1, 2, 3, 4, 6, 8):
        ws = torch.empty(int(L.drn_gemm_nt_splitk_ws_elems(M, Cout, ks)), dtype=torch.float32, device=dev)
        fn = lambda: _lib.check(L.drn_gemm_nt_splitk(arr, ks, ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(counters.data_ptr()), 1, stream), "x")
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append("ks%d %.1f" % (ks, e0.elapsed_time(e1) / 50 * 1e3))
    print("%-12s M=%5d N=%4d K=%5d tiles=%3d ksteps=%3d : %s us" % (name, M, Cout, taps * Cin, ((M + 127) // 128) * ((Cout + 127) // 128),
                                                                 taps * Cin // 64, "  ".join(res)))
