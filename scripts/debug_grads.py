"""Debug helper: per-parameter gradient-norm comparison of the HIP model against a golden case."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import build_model, case_inputs, load_golden
from drn_amd.model import mainModel

name = sys.argv[1]
g = load_golden(name)
cfg, batch = case_inputs(g)
m = build_model(mainModel, cfg, device="cuda:0")
batch = [b.cuda() if i != 1 else b for i, b in enumerate(batch)]
m.train()
_, losses = m(*batch)
print({k: v.detach().cpu().numpy() for k, v in losses.items()}, {k: g[k] for k in ("loss_cls", "loss_reg", "loss_iou")})
print("counts", m.fcos.loss_evaluator.last_counts)
stage = int(g["stage"])
loss = losses["loss_iou"] if stage == 2 else sum(l for l in losses.values())
loss.backward()
for k, p in m.named_parameters():
    if "gn/" + k not in g:
        continue
    ref = float(g["gn/" + k][0])
    got = float(p.grad.norm()) if p.grad is not None else float("nan")
    flag = "" if abs(got - ref) <= 1e-4 * ref + 1e-7 else "  <<<<<"
    print("%-50s ref %.6e got %.6e rel %.2e%s" % (k, ref, got, abs(got - ref) / (ref + 1e-12), flag))
