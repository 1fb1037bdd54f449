"""cProfile of Trainer.evaluate on synthetic device-resident batches (B=32, T=256): where the eval loop's host time goes."""
import cProfile, pstats, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench as BN
from drn_amd import trainer as TR
from drn_amd.model import mainModel
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
dev = torch.device("cuda:0")
B, T, D = 32, 256, 4096
cfg = default_cfg("C3D", D, 1)
m = BN.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
tr = TR.Trainer(m, 1, lr=1e-3, clip_gradient=0.5, graph=False)
batches = [BN.collate_like([t.to(dev) if torch.is_tensor(t) else t for t in synthetic_batch(B, T, D, seed=100 + i)], ["v%d" % i] * B) for i in range(4)]
tr.evaluate(batches)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
tr.evaluate(batches)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
