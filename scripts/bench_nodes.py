"""Every launch of the benchmarked step ON ITS OWN: records the C-ABI calls of one eager step (entry point + arguments, which stay
alive), then replays each call 50 x as a linear hipGraph and reports us per node -- the kernel hot and alone, with the per-node
dispatch cost of a graph (an empty kernel: 1.6 us) but none of the step's cold operands.  Next to the in-step durations
(profiles/*_step_kernel_sequence.txt) this separates "the kernel is slow" from "its operands are cold": round 3 found the query
attention kernels at 12 us here (a 3000-instruction select chain), i.e. slow by themselves.
usage (GPU box): python scripts/bench_nodes.py [--T 256] [--min-us 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from drn_amd import _lib as L
from drn_amd import dist as ddist
from drn_amd import functional as DF
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import default_cfg, synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=256)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--min-us", type=float, default=0.0)
ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="drn_tune(KEY, VALUE) before the replays (experiments)")
ap.add_argument("--only", default=None, help="only entry points whose name contains this")
ap.add_argument("--lstm-f32", action="store_true", help="keep the BiLSTM's recurrent products on the fp32 kernels in the bf16 model")
args = ap.parse_args()
if args.lstm_f32:
    DF._lstm_lowp = lambda lowp, H: False
dev = torch.device("cuda:0")
cfg = default_cfg("C3D", 4096, 1)
model = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
params = B.stage_params(model, 1)
model.train()
red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=model.grad_stack_groups(), bucket_bytes=1 << 30)
opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
batch = [b.to(dev) for b in synthetic_batch(32, args.T, 4096, seed=1)]


def step():
    red.zero()
    _, losses = model(*batch)
    DF.backward(DF.loss_total(losses))
    red.finish()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()

real = L.lib()
calls = []
NO_LAUNCH = ("drn_last_error", "drn_abi_version", "drn_tune", "drn_opt_nblocks")


class Recorder(object):
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith("drn_") or name in NO_LAUNCH or name.endswith("_ws_elems") or name.endswith("_ws_rows"):
            return fn

        def wrapped(*a):
            calls.append((name, fn, a))
            return fn(*a)
        return wrapped


L._lib = Recorder()
step()
torch.cuda.synchronize()
L._lib = real
for kv in args.tune:
    k, v = kv.split("=")
    L.check(real.drn_tune(k.encode(), int(v)), "drn_tune")
keep = list(calls)                      # (the argument tuples keep host descriptor arrays alive; device tensors live in the autograd graph
print("%d C-ABI calls in one step" % len(keep))                                           # of the last step, which we never free)
s = torch.cuda.Stream()
rows = []
for i, (name, fn, a) in enumerate(keep):
    if args.only and args.only not in name:
        continue
    a = list(a)
    if os.environ.get("BENCH_NODES_DEBUG"):
        print("call %d %s%r" % (i, name, tuple(x.value if hasattr(x, "value") else x for x in a)), flush=True)
    # the stream argument is the last one: replay on the capture stream
    import ctypes
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(s):
            a[-1] = ctypes.c_void_p(s.cuda_stream)
            for _ in range(2):
                fn(*a)
        torch.cuda.synchronize()
        with torch.cuda.stream(s):          # (not `with torch.cuda.graph`: it empties the allocator cache first, which unmaps the
            g.capture_begin()               # step's freed temporaries the recorded pointers still name)
            for _ in range(args.reps):
                fn(*a)
            g.capture_end()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (4 * args.reps) * 1e3
    except Exception as e:                      # noqa: BLE001
        us = float("nan")
        print("  (%s failed: %s)" % (name, str(e).split("\n")[0]))
    rows.append((i, name, us))
    print("%3d %-34s %8.2f us / call (all its launches)" % (i, name, us), flush=True)
tot = 0.0
for i, name, us in rows:
    tot += us if us == us else 0.0
print("sum over the step's calls: %.1f us (hot, isolated, graph nodes)" % tot)
