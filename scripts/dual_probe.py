"""A/B in one process: the single-graph step (GraphedStep) vs the two-stream step (DualStreamStep) on bench.py's workload,
interleaved rounds; host time per call and total time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from drn_amd import dist as ddist, functional as DF
from drn_amd.graph import DualStreamStep, GraphedStep
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
prio = int(sys.argv[1]) if len(sys.argv) > 1 else -1
wfirst = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mfirst = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
cfg = default_cfg("C3D", 4096, 1)
batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]

def make():
    model = Bn.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
    params = Bn.stage_params(model, 1)
    model.train()
    red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=model.grad_stack_groups(), bucket_bytes=1 << 30)
    return model, red, FusedAdam(red, lr=1e-3, max_norm=0.5)

m1, r1, o1 = make()
def step():
    r1.zero()
    _, ls = m1(*batch)
    DF.backward(DF.loss_total(ls))
    r1.finish()
    o1.step()
    return ls
single = GraphedStep(step, warmup=3).capture()
m2, r2, o2 = make()
dual = DualStreamStep(m2, batch[:5], DF.loss_total, r2, o2, wgrads_first=bool(wfirst), side_priority=prio, main_first=bool(mfirst)).warm(3).capture()

def timeit(fn, N=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3
for rnd in range(3):
    a = timeit(single)
    b = timeit(dual)
    print("round %d  single: host %.3f total %.3f ms | dual(prio %d, wgrads_first %d, main_first %d): host %.3f total %.3f ms"
          % (rnd, a[0], a[1], prio, wfirst, mfirst, b[0], b[1]))
