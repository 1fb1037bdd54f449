"""Debug helper: HIP model (GPU) vs CPU oracle on the same seeded case: activations and gradients, elementwise."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import build_model, case_inputs, load_golden
from drn_amd.model import mainModel
from oracle import drn_oracle as O

name = sys.argv[1]
dtype = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
g = load_golden(name)
cfg, batch = case_inputs(g)
mo = build_model(O.mainModel, cfg)
mh = build_model(mainModel, cfg, device="cuda:0")
mh.set_compute_dtype(dtype)
taps_o = {}
mods = dict(mo.named_modules())
for t in ["backbone_net.forward_conv0", "backbone_net.forward_conv1", "backbone_net.forward_conv2", "fpn.fpn_layer1", "fpn.fpn_layer2", "fpn.fpn_layer3"]:
    mods[t].register_forward_hook(lambda mod, i, o, n=t: taps_o.__setitem__(n, o))
mo.fcos.head.register_forward_hook(lambda mod, i, o: taps_o.__setitem__("head", o))
mo.train(); mh.train()
bo = batch
_, lo = mo(*bo)
mh.taps = {}
_, lh = mh(*[b.cuda() if i != 1 else b for i, b in enumerate(batch)])
for k, v in taps_o.items():
    if k == "head":
        for j, nm in ((0, "logits"), (1, "reg"), (3, "iou")):
            for l in range(3):
                a, b = v[j][l], mh.taps["head"][j][l].detach().cpu().double()
                print("%-34s max|err| %.3e  scale %.3g" % ("%s%d" % (nm, l), (a - b).abs().max(), a.abs().max()))
    else:
        a, b = v, mh.taps[k].detach().cpu().double()
        print("%-34s max|err| %.3e  scale %.3g" % (k, (a - b).abs().max(), a.abs().max()))
for k in lo:
    print(k, float(lo[k].reshape(-1)[0]), float(lh[k].reshape(-1)[0]))
stage = int(g["stage"])
(lo["loss_iou"] if stage == 2 else sum(lo.values())).backward()
(lh["loss_iou"] if stage == 2 else sum(lh.values())).backward()
po = dict(mo.named_parameters())
for k, p in mh.named_parameters():
    if po[k].grad is None or p.grad is None:
        continue
    a, b = po[k].grad.double(), p.grad.detach().cpu().double()
    print("%-50s rel-L2 %.2e  |g| %.3e" % (k, float((a - b).norm() / (a.norm() + 1e-30)), float(a.norm())))
