import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drn_amd.dist import GradReducer
from drn_amd.graph import GraphedStep
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
B, T, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
def make():
    m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("C3D" if D == 4096 else "X", D, 1)), compute_dtype=torch.bfloat16)
    m.load_state_dict(seeded_state_dict(m, 0)); m = m.cuda().train()
    for n, p in m.named_parameters():
        if "iou_scores" in n or "mix_fc" in n: p.requires_grad_(False)
    params = [p for p in m.parameters() if p.requires_grad]
    red = GradReducer(params, world_size=1); opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
    batch = [b.cuda() for b in synthetic_batch(B, T, D, seed=1)]
    def step():
        red.zero(); _, losses = m(*batch); sum(losses.values()).backward(); red.finish(); opt.step(); return losses
    return m, step, opt
m1, s1, o1 = make()
e = [float(s1()["loss_cls"].detach()) for _ in range(10)]
m2, s2, o2 = make()
g = GraphedStep(s2, warmup=2).capture()
gl = [float(g()["loss_cls"].detach()) for _ in range(8)]
print("eager", ["%.4f" % x for x in e])
print("graph", ["%.4f" % x for x in gl], "(starts at step 2)")
print("norms", float(o1.total_norm()), float(o2.total_norm()))
d = max(float((a - b).abs().max()) for a, b in zip(m1.state_dict().values(), m2.state_dict().values()) if a.is_floating_point())
print("max param diff", d)
