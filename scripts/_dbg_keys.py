import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from drn_amd import dist as ddist, functional as DF
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import default_cfg, synthetic_batch
dev = torch.device("cuda", 0)
cfg = default_cfg("C3D", 4096, 1)
model = Bn.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
params = Bn.stage_params(model, 1)
model.train()
red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=model.grad_stack_groups(), bucket_bytes=1 << 30)
opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]
for _ in range(2):
    red.zero(); _, ls = model(*batch); DF.backward(DF.loss_total(ls)); red.finish(); opt.step()
names = {p.data_ptr(): n for n, p in model.named_parameters()}
skip = opt._mirror_keys
for key, (ver, buf, ref) in DF._pack_cache.items():
    w = ref()
    if key not in skip and ("pack", key) not in skip:
        print("pack NOT skipped:", names.get(w.data_ptr()), tuple(w.shape), key[2], key[3], "trainable", w.requires_grad)
for key, (ver, buf, refs) in DF._pstack_cache.items():
    if ("pstack", key) not in skip:
        print("pstack NOT skipped:", [names.get(r().data_ptr()) for r in refs], key[1], key[2])
for key, (ver, buf, refs) in DF._stack_cache.items():
    if ("stack", key) not in skip:
        print("stack NOT skipped:", [names.get(r().data_ptr()) for r in refs], "T" if key[0] == "t" else "")
