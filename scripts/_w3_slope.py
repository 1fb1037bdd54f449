import os, sys, torch
sys.path.insert(0, "/root/repo")
from drn_amd import ops
dev = "cuda:0"; dt = torch.bfloat16; code = ops.BF16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N, Cin = 256, 4352
os.environ["DRN_TN3_TARGET"] = "68"
for dbg in (0, 1, 2, 4, 8, 3, 5, 6, 7, 15):
    os.environ["DRN_TN3_DBG"] = str(dbg)
    r = []
    for B in (32, 64):
        dY = torch.randn(B * 256, N, device=dev).to(dt); X = torch.randn(B * 256, Cin, device=dev).to(dt)
        dW = torch.zeros(N, 3, Cin, device=dev)
        d = [ops.wgrad_desc(dY, X, B * 256, Lout=256, Lsrc=256)]
        r.append(timeit(lambda: ops.gemm_wgrad(d, dW, N, Cin, taps=3, stride=1, pad=1, w_layout=0, dtype=code)))
    print("dbg=%2d  (1 noMFMA 2 noLDSread 4 noStage 8 noBarrier)  %7.3f %7.3f ms -> %6.3f us/block" % (dbg, r[0], r[1], (r[1] - r[0]) * 1e3 / 128))
