"""rocprofv3 --kernel-trace target: ForkedStep only (scripts/experiments/one_graph_two_branches.py measures it against the linear graph)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench as B
from drn_amd import dist as ddist, functional as DF
from drn_amd.graph import ForkedStep
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
dev = torch.device("cuda:0")
cfg = default_cfg("C3D", 4096, 1)
m = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
params = B.stage_params(m, 1)
m.train()
red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=m.grad_stack_groups(), bucket_bytes=1 << 30)
opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]
fs = ForkedStep(m, batch[:5], lambda l: DF.loss_total(l), red, opt).warm(3).capture()
for _ in range(20):
    fs()
torch.cuda.synchronize()
