"""How long does the HOST spend in one hipGraph replay of the bench step, and does alternating two captured copies help?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
from drn_amd import dist as ddist
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.graph import GraphedStep

dev = torch.device("cuda", 0)
cfg = default_cfg("C3D", 4096, 1)
m = mainModel(VOCAB_SIZE, as_namespace(cfg), compute_dtype=torch.bfloat16)
m.load_state_dict(seeded_state_dict(m, 0)); m = m.to(dev).train()
for n, p in m.named_parameters():
    if "iou_scores" in n or "mix_fc" in n: p.requires_grad_(False)
params = [p for p in m.parameters() if p.requires_grad]
red = ddist.GradReducer(params, world_size=1)
opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]
def step():
    red.zero(); _, losses = m(*batch); sum(losses.values()).backward(); red.finish(); opt.step(); return losses
g1 = GraphedStep(step, warmup=3).capture()
g2 = GraphedStep.__new__(GraphedStep); g2.step_fn, g2.graph, g2.out, g2.stream = step, None, None, g1.stream
g2.capture()
def run(fn, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
for name, fn in [("one graph", lambda i: g1()), ("two graphs alternating", lambda i: (g1 if i % 2 == 0 else g2)())]:
    run(fn, 5)
    h, t = run(fn)
    print("%-26s host %.3f ms/replay   total %.3f ms/step" % (name, h, t))
# replay with a sync after each: pure GPU latency of one replay
def synced(i):
    g1(); torch.cuda.synchronize()
h, t = run(synced)
print("%-26s total %.3f ms/step" % ("one graph, sync each", t))
