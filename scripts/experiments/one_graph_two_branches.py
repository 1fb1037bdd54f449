"""ONE hipGraph with two branches (the query side beside input prep / weight gradients) against the shipped single linear graph.
Round 2 recorded that such a graph "leaves the fast submission path (host 0.05 -> 1.9 ms per replay)"; this measures what the GPU
does with it: wall clock per step over back-to-back replays (host-bound or not, that is the step time a user sees).
usage (GPU box): python scripts/experiments/one_graph_two_branches.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench as B
from drn_amd import dist as ddist, functional as DF
from drn_amd.graph import DualStreamStep, GraphedStep
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import default_cfg, synthetic_batch

dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def setup():
    cfg = default_cfg("C3D", 4096, 1)
    m = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
    params = B.stage_params(m, 1)
    m.train()
    red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=m.grad_stack_groups(), bucket_bytes=1 << 30)
    opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
    batch = [b.to(dev) for b in synthetic_batch(32, T, 4096, seed=1)]
    return m, red, opt, batch


def wall(run, n=100):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def host_only(run, n=20):
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2] * 1e3


loss_of = lambda losses: DF.loss_total(losses)
# (a) the shipped linear graph
m, red, opt, batch = setup()


def step():
    red.zero()
    _, losses = m(*batch)
    DF.backward(loss_of(losses))
    red.finish()
    opt.step()
    return losses


g = GraphedStep(step, warmup=3).capture()
print("linear graph            : %.3f ms/step (host side of one replay %.3f ms)" % (wall(g), host_only(g)))
del g, m, red, opt

# (b) one graph, two branches: schedule variants, each on the best of four candidate side streams (ForkedStep.capture)
from drn_amd.graph import ForkedStep
for name, kw, wf in (("prop_fc weight gradient first (shipped)", {}, False), ("small weight gradients first", {}, True),
                     ("split gate: query encoder beside the prop_fc GEMM", {"split_gate": True}, False)):
    m, red, opt, batch = setup()
    fs = ForkedStep(m, batch[:5], loss_of, red, opt, **kw)
    fs.wgrads_first = wf
    fs.warm(3).capture(tries=4)
    print("forked, %-52s: %.3f ms/step (host side of one replay %.3f ms) candidates %s" % (name, wall(fs), host_only(fs), fs.probe_log))
    del fs, m, red, opt
