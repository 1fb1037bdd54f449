"""Trainer.train_epoch (device-resident batches, T = 256) with the linear and the two-branch captured step, next to bench.py's
back-to-back replay: where does the gap between what train.py runs and the headline come from?
usage (GPU box): python scripts/experiments/trainer_gap.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench as B_
from drn_amd import trainer as TR
from drn_amd.model import mainModel
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
dev = torch.device("cuda:0")
B, T, D, stage = 32, int(os.environ.get("T", "256")), 4096, 1
cfg = default_cfg("C3D", D, stage)
batches = [B_.collate_like([t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(B, T, D, seed=100 + i)], ["v%d" % i] * B) for i in range(8)]
for rnd in range(1 if os.environ.get("ONLY_FORKED") == "1" else 2):
  for pre_dev in ((int(os.environ["PRE_DEV"]),) if "PRE_DEV" in os.environ else (1 << 40, 64 << 20)):
    for forked in ((True,) if os.environ.get("ONLY_FORKED") == "1" else (False, True)):
        m = B_.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
        tr = TR.Trainer(m, stage, lr=1e-3, clip_gradient=0.5, graph=True, forked=forked)
        tr.prefetch_device_bytes = pre_dev
        for _ in range(3):
            tr.train_epoch(batches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_epoch(batches * 25)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200 * 1e3
        print("round %d T=%d forked=%s device-prefetch=%s: %.3f ms/step" % (rnd, T, forked, pre_dev < (1 << 40), dt), flush=True)
        tr.reducer.remove()
        del m, tr
