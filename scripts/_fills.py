import os, sys, torch, traceback
sys.path.insert(0, os.getcwd())
from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
from drn_amd import dist as ddist, functional as DF
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
cfg = default_cfg("C3D", 4096, 1)
m = mainModel(VOCAB_SIZE, as_namespace(cfg), compute_dtype=torch.bfloat16); m.load_state_dict(seeded_state_dict(m, 0)); m = m.to(dev).train()
for n, p in m.named_parameters():
    if "iou_scores" in n or "mix_fc" in n: p.requires_grad_(False)
params = [p for p in m.parameters() if p.requires_grad]
red = ddist.GradReducer(params, world_size=1, adjacent=m.grad_stack_groups(), bucket_bytes=1 << 30)
opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]
def step():
    red.zero(); _, losses = m(*batch); DF.backward(DF.loss_total(losses)); red.finish(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
orig = torch.Tensor.zero_
def zz(self):
    print("ZERO_", tuple(self.shape)); traceback.print_stack(limit=6); return orig(self)
torch.Tensor.zero_ = zz
step(); torch.cuda.synchronize()
torch.Tensor.zero_ = orig
sys.exit()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::ones_like", "aten::copy_", "aten::sum", "aten::_foreach_copy_", "aten::to", "aten::_to_copy", "aten::add", "aten::mul"):
        st = [s for s in (e.stack or []) if "drn_amd" in s or "bench" in s or "_fills" in s][:3]
        print(e.name, e.input_shapes if hasattr(e, "input_shapes") else "", st)
