"""rocprofv3 --kernel-trace target: Trainer.train_epoch at T = 256 fed from pinned bf16 host batches (what train.py does)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench as B
from drn_amd import trainer as TR
from drn_amd.model import mainModel
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
dev = torch.device("cuda:0")
cfg = default_cfg("C3D", 4096, 1)
host = len(sys.argv) > 1 and sys.argv[1] == "host"
bs = [B.collate_like(list(synthetic_batch(32, 256, 4096, seed=100 + i)), ["v%d" % i] * 32) for i in range(8)]
if host:
    bs = [tuple((t.bfloat16() if i == 2 else t).pin_memory() if torch.is_tensor(t) else t for i, t in enumerate(b)) for b in bs]
else:
    bs = [tuple(t.to(dev) if torch.is_tensor(t) else t for t in b) for b in bs]
m = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
tr = TR.Trainer(m, 1, lr=1e-3, clip_gradient=0.5, graph=True)
for _ in range(6):
    tr.train_epoch(bs)
torch.cuda.synchronize()
