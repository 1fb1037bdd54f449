import os, sys, torch
sys.path.insert(0, os.getcwd())
from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
from drn_amd import dist as ddist, functional as DF
from drn_amd.model import mainModel
dev = torch.device("cuda", 0)
for stage in (1, 3):
    cfg = default_cfg("C3D", 4096, stage)
    m = mainModel(VOCAB_SIZE, as_namespace(cfg), compute_dtype=torch.bfloat16); m.load_state_dict(seeded_state_dict(m, 0)); m = m.to(dev).train()
    if stage == 1:
        for n, p in m.named_parameters():
            if "iou_scores" in n or "mix_fc" in n: p.requires_grad_(False)
    params = [p for p in m.parameters() if p.requires_grad]
    red = ddist.GradReducer(params, world_size=1, adjacent=m.grad_stack_groups(), bucket_bytes=1 << 30)
    batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]
    red.zero(); _, losses = m(*batch); DF.loss_total(losses).backward()
    names = {id(p): n for n, p in m.named_parameters()}
    for b in red.buckets:
        for p, v in zip(b.params, b.views):
            if p.grad is None: print(stage, "NONE   ", names[id(p)])
            elif p.grad.data_ptr() != v.data_ptr(): print(stage, "FOREIGN", names[id(p)], tuple(p.shape))
