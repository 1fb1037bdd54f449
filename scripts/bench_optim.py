"""The optimizer part of the benchmarked step on its own: builds bench.py's model + reducer + FusedAdam, runs a few real steps so
that every weight-copy cache entry exists, lists the tensors drn_adam_tiled walks (shape, copies, bucket-offset alignment, tiles)
and times clip + Adam (drn_sumsq_partials + drn_sumsq_finalize + drn_adam_bucket + drn_adam_tiled) as one replayed hipGraph.
usage (GPU box): python scripts/bench_optim.py [--items]   (run it under `rocprofv3 --kernel-trace --stats` for per-kernel times)"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from drn_amd import dist as ddist
from drn_amd import functional as DF
from drn_amd._lib import AdamTiledItem
from drn_amd.graph import GraphedStep
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import default_cfg, synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--items", action="store_true")
ap.add_argument("--T", type=int, default=256)
ap.add_argument("--reps", type=int, default=50)
args = ap.parse_args()
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
st = opt.state[0]
n = red.buckets[0].flat.numel()
if st.get("tiled") is not None:
    raw, bi, bt, nb = st["tiled"]
    items = (AdamTiledItem * (raw.numel() // ctypes.sizeof(AdamTiledItem))).from_buffer_copy(raw.cpu().numpy().tobytes())
    tot = sum(it.R * it.C * it.k for it in items)
    print("bucket %.2fM params; tiled: %d items, %d tiles, %.2fM params; linear kernel: %.2fM" % (n / 1e6, len(items), nb, tot / 1e6, (n - tot) / 1e6))
    if args.items:
        name_of = dict((p.data_ptr(), k) for k, p in model.named_parameters())
        for it in items:
            print("  %-44s R=%5d C=%5d k=%d copies %s/%s off%%32=%2d tiles=%4d" % (
                name_of.get(it.p, "?"), it.R, it.C, it.k, ("-", "f32", "bf16")[(it.code1 + 1) if it.m1 else 0],
                ("-", "f32", "bf16")[(it.code2 + 1) if it.m2 else 0], it.off % 32, ((it.R + 63) // 64) * it.tiles_c))

g = GraphedStep(lambda: opt.step(), warmup=2).capture()
g()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    g()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / args.reps * 1e3
print("clip + Adam as a graph: %.1f us per step, %.0f GB/s on 28 B/param (+ copies)" % (us, n * 28 / us / 1e3))
