"""Amplifier for the rare mismatch of the two-branch hipGraph step (tests/test_graph_gpu.py::test_forked_graph_step_is_bit_identical,
1 fresh process in ~400 on its own): with lr = 0 every replay must reproduce the same losses and the same gradient bucket bit for bit,
so any replay that differs from the first one IS the event -- and the gradient slices that moved say which launch it hit.
usage: python scripts/experiments/forked_race_hunt.py [replays] [bf16|f32] [forked|linear]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from drn_amd.dist import GradReducer
from drn_amd.graph import ForkedStep, GraphedStep
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
import drn_amd.functional as DF

from drn_amd import _lib
for _k in ("exp0", "exp1", "exp2", "exp3", "exp4"):
    if os.environ.get("HUNT_" + _k.upper()):
        assert _lib.lib().drn_tune(_k.encode(), int(os.environ["HUNT_" + _k.upper()])) == 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dtype = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.bfloat16
mode = sys.argv[3] if len(sys.argv) > 3 else "forked"
dev = "cuda:0"
shape = os.environ.get("HUNT_SHAPE", "4,32,64")
B, T, D = [int(x) for x in shape.split(",")]
m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY" if D != 4096 else "C3D", D, 3)), compute_dtype=dtype)
m.load_state_dict(seeded_state_dict(m, 0))
m = m.to(dev).train()
params = [p for p in m.parameters() if p.requires_grad]
if os.environ.get("HUNT_ROTATE") == "1":          # optimizer-first order of the two-branch step (query side in buckets of its own)
    qset = set(id(p) for p in m.query_parameters())
    red = GradReducer(params, world_size=1, bucket_bytes=1 << 30, adjacent=m.grad_stack_groups(),
                      groups=[[p for p in params if id(p) in qset], [p for p in params if id(p) not in qset]])
else:
    red = GradReducer(params, world_size=1, bucket_bytes=1 << 30, adjacent=m.grad_stack_groups())
opt = FusedAdam(red, lr=0.0, max_norm=0.5)
batch = [b.to(dev) for b in synthetic_batch(B, T, D, seed=1)]
names = {id(p): n for n, p in m.named_parameters()}
if mode == "forked":
    fs = ForkedStep(m, batch, DF.loss_total, red, opt)
    for _ in range(3):
        fs()
    fs.capture()
else:
    def step():
        red.zero()
        _, ls = m(*batch)
        DF.backward(DF.loss_total(ls))
        red.finish()
        opt.step()
        return ls
    fs = GraphedStep(step, warmup=3).capture()
ls = fs()
torch.cuda.synchronize()
flat0 = [b.flat.clone() for b in red.buckets]
loss0 = torch.stack([ls[k].reshape(-1)[0].float() for k in ("loss_cls", "loss_reg", "loss_iou")]).clone()
sd0 = {k: v.clone() for k, v in m.state_dict().items() if "running" not in k and "num_batches" not in k}
events = 0
t0 = time.time()
for it in range(N):
    ls = fs()
    cur = torch.stack([ls[k].reshape(-1)[0].float() for k in ("loss_cls", "loss_reg", "loss_iou")])
    same = bool(torch.equal(cur, loss0)) and all(bool(torch.equal(b.flat, f0)) for b, f0 in zip(red.buckets, flat0))
    if not same:
        events += 1
        moved = []
        for b, f0 in zip(red.buckets, flat0):
            for p, v in zip(b.params, b.views):
                off = v.data_ptr() - b.flat.data_ptr()
                ref = f0.view(torch.uint8)[off:off + v.numel() * 4].view(torch.float32)
                ne = (v != ref)
                if bool(ne.any()):
                    moved.append("%s %d/%d maxdiff %.3g" % (names.get(id(p), "?"), int(ne.sum()), v.numel(), float((v - ref).abs().max())))
        print("EVENT replay %d: losses %s; %d gradient tensors moved: %s" % (it, "same" if bool(torch.equal(cur, loss0)) else "MOVED", len(moved), "; ".join(moved[:int(os.environ.get("HUNT_SHOW", "3"))])), flush=True)
drift = [k for k, v in m.state_dict().items() if k in sd0 and not torch.equal(v, sd0[k])]
print("%s %s %s: %d replays, %d events, %.1f s; parameters that moved with lr = 0: %d" % (mode, dtype, shape, it + 1, events, time.time() - t0, len(drift)), flush=True)
