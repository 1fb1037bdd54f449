"""Hunt for the order-dependent mismatch of tests/test_graph_gpu.py::test_forked_graph_step_is_bit_identical (seen ~1 run in 8 when
test_functional_gpu.py / test_gemm_gpu.py ran earlier in the same process): after those tests, compare (a) eager vs eager, (b) forked
vs eager, several rounds each, and say which side moved.  usage: python scripts/experiments/flake_probe.py [rounds]"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
import torch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
if os.environ.get("PROBE_PRE", "1") == "1":
    pytest.main(["-q", "-x", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_functional_gpu.py"), os.path.join(ROOT, "tests", "test_gemm_gpu.py")])

from drn_amd.dist import GradReducer
from drn_amd.graph import ForkedStep
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
import drn_amd.functional as DF
dev = "cuda:0"


def build(dtype):
    m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=dtype)
    m.load_state_dict(seeded_state_dict(m, 0))
    m = m.to(dev).train()
    red = GradReducer([p for p in m.parameters() if p.requires_grad], world_size=1, bucket_bytes=1 << 30, adjacent=m.grad_stack_groups())
    return m, red, FusedAdam(red, lr=1e-4, max_norm=0.5)


def eager(dtype, n, batch):
    m, r, o = build(dtype)
    out = []
    for _ in range(n):
        r.zero()
        _, ls = m(*batch)
        DF.backward(DF.loss_total(ls))
        r.finish()
        o.step()
        out.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    r.remove()
    return out, {k: v.clone() for k, v in m.state_dict().items()}


def forked(dtype, n, batch):
    m, r, o = build(dtype)
    fs = ForkedStep(m, batch, DF.loss_total, r, o)
    out = []
    for _ in range(3):
        ls = fs()
        out.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    fs.capture()
    out += [None] * fs.tuning_steps
    for _ in range(n - 3 - fs.tuning_steps):
        ls = fs()
        out.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    torch.cuda.synchronize()
    r.remove()
    return out, {k: v.clone() for k, v in m.state_dict().items()}


def first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x is not None and y is not None and x != y:
            return i
    return None


batch = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]
for dtype in ((torch.bfloat16,) if os.environ.get("PROBE_DTYPES") == "bf16" else (torch.bfloat16, torch.float32)):
    base, sd0 = eager(dtype, 64, batch)
    for rnd in range(rounds):
        e, sde = eager(dtype, 64, batch)
        f, sdf = forked(dtype, 64, batch)
        print(dtype, rnd, "eager-vs-eager first diff", first_diff(base, e), "forked-vs-eager first diff", first_diff(base, f),
              "params differ e/f:", sum(not torch.equal(sd0[k], sde[k]) for k in sd0), sum(not torch.equal(sd0[k], sdf[k]) for k in sd0), flush=True)
