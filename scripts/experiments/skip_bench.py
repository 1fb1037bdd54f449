"""Times the norm pass of a step's real gradient bucket: plain / skip with host classes / skip walking the ranges, cache-warm
(back to back) and cache-cold (a 1 GB fill in between)."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench as B
from drn_amd import dist as ddist, functional as DF, optim, _lib
from drn_amd.model import mainModel
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
dev = torch.device("cuda:0")
cfg = default_cfg("C3D", 4096, 1)
batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]
m = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
params = B.stage_params(m, 1)
m.train()
red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=m.grad_stack_groups(), bucket_bytes=1 << 30)
opt = optim.FusedAdam(red, lr=1e-3, max_norm=0.5)
red.zero()
_, ls = m(*batch)
DF.backward(DF.loss_total(ls))
red.finish()
notes = list(red.sumsq_notes)
flat = red.buckets[0].flat
base, n = flat.data_ptr(), flat.numel()
rng = sorted(((p - base) // 4, (p - base) // 4 + ne) for nt in notes for p, ne in nt[0])
merged = []
for lo, hi in rng:
    if merged and merged[-1][1] >= lo:
        merged[-1][1] = max(merged[-1][1], hi)
    else:
        merged.append([lo, hi])
print("n", n, "merged", merged, "skipped", sum(h - l for l, h in merged))
L = _lib.lib()
nb = L.drn_opt_nblocks(ctypes.c_int64(n))
lo_a = (ctypes.c_int64 * len(merged))(*[x[0] for x in merged])
hi_a = (ctypes.c_int64 * len(merged))(*[x[1] for x in merged])
host = (ctypes.c_ubyte * nb)()
L.drn_sumsq_block_classes(ctypes.c_int64(n), lo_a, hi_a, len(merged), host)
cls = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(dev)
print("blocks", nb, "classes", torch.bincount(cls.long()).tolist())
part = torch.zeros(nb, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
junk = torch.empty(1 << 28, device=dev)
def plain(): L.drn_sumsq_partials(P(flat), ctypes.c_int64(n), P(part), None, st)
def skipc(): L.drn_sumsq_partials_skip(P(flat), ctypes.c_int64(n), P(part), None, lo_a, hi_a, len(merged), P(cls), st)
def skipw(): L.drn_sumsq_partials_skip(P(flat), ctypes.c_int64(n), P(part), None, lo_a, hi_a, len(merged), None, st)
for name, fn in (("plain", plain), ("skip+classes", skipc), ("skip walking", skipw)):
    for cold in (False, True):
        ts = []
        for _ in range(12):
            if cold:
                junk.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print("%-14s %s  median %.1f us  min %.1f" % (name, "cold" if cold else "warm", ts[len(ts) // 2], ts[0]))
from drn_amd import ops
print("deferred reduce: %.1f MB per launch" % (ops.last_reduce_bytes / 1e6))
