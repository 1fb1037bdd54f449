"""Long-run bit-identity of the two-branch step: N replays of ForkedStep against N eager steps (same seeds), bf16 and f32, TINY model;
a transient race between the branches shows up as a divergence somewhere in the run.  Control: the linear GraphedStep.
usage: python scripts/experiments/forked_stress.py [N=1500]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from drn_amd.dist import GradReducer
from drn_amd.graph import ForkedStep, GraphedStep
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
import drn_amd.functional as DF

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = "cuda:0"


def build(dtype):
    m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=dtype)
    m.load_state_dict(seeded_state_dict(m, 0))
    m = m.to(dev).train()
    red = GradReducer([p for p in m.parameters() if p.requires_grad], world_size=1, bucket_bytes=1 << 30, adjacent=m.grad_stack_groups())
    return m, red, FusedAdam(red, lr=1e-5, max_norm=0.5)


def losses(ls):
    return torch.cat([ls[k].detach().reshape(-1)[:1] for k in ("loss_cls", "loss_reg", "loss_iou")])


batch = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]
for dtype in (torch.bfloat16, torch.float32):
    m1, r1, o1 = build(dtype)

    def step1():
        r1.zero()
        _, ls = m1(*batch)
        DF.backward(DF.loss_total(ls))
        r1.finish()
        o1.step()
        return ls
    ref = torch.empty(N, 3, device=dev)
    for i in range(N):
        ref[i] = losses(step1())
    r1.remove()
    for kind in ("forked", "linear"):
        m2, r2, o2 = build(dtype)
        if kind == "forked":
            st = ForkedStep(m2, batch, DF.loss_total, r2, o2)
            st.warm(3)
            st.capture(tries=1, probe=0)
            done = 3
        else:
            def step2():
                r2.zero()
                _, ls = m2(*batch)
                DF.backward(DF.loss_total(ls))
                r2.finish()
                o2.step()
                return ls
            st = GraphedStep(step2, warmup=3).capture()
            done = 3
        got = torch.full((N, 3), float("nan"), device=dev)
        for i in range(done, N):
            got[i] = losses(st())
        torch.cuda.synchronize()
        diff = (got[done:] != ref[done:]).any(dim=1).nonzero().reshape(-1)
        sd1, sd2 = m1.state_dict(), m2.state_dict()
        nbad = sum(not torch.equal(sd1[k], sd2[k]) for k in sd1)
        print(dtype, kind, "steps", N, "first differing step:", (int(diff[0]) + done) if diff.numel() else None, "params differing:", nbad, flush=True)
        r2.remove()
