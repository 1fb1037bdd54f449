"""bench.py's trainer section alone (Trainer.train_epoch: device-resident, pinned bf16 / fp32 host batches, evaluation)."""
import faulthandler, json, os, sys
faulthandler.enable()
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench as B_
from drn_amd.utils.synthetic import default_cfg
dev = torch.device("cuda:0")
Ts = [int(t) for t in os.environ.get("TS", "32,256").split(",")]
out = B_.trainer_lines(default_cfg("C3D", 4096, 1), dev, torch.bfloat16, 1, 32, 4096, Ts, 64, graph_modes=(True,))
for k, v in out.items():
    print(k, v, flush=True)
