"""Is the step bit-identical whichever candidate ForkedStep.capture() keeps?  Runs tests/test_graph_gpu.py's forked test body with
the choice forced to candidate 0 / candidate 1 (normally the faster of the two: timing-dependent, so a defect in one of the two
paths shows up as a flaky test).  usage: python scripts/experiments/forked_candidate_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from drn_amd.dist import GradReducer
from drn_amd.graph import ForkedStep
from drn_amd.model import mainModel
from drn_amd.optim import FusedAdam
from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
import drn_amd.functional as DF

dev = "cuda:0"


def run(dtype, force):
    def build():
        m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=dtype)
        m.load_state_dict(seeded_state_dict(m, 0))
        m = m.to(dev).train()
        red = GradReducer([p for p in m.parameters() if p.requires_grad], world_size=1, bucket_bytes=1 << 30, adjacent=m.grad_stack_groups())
        return m, red, FusedAdam(red, lr=1e-4, max_norm=0.5)
    batch = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]
    n = 64
    m1, r1, o1 = build()
    ref = []
    for _ in range(n):
        r1.zero()
        _, ls = m1(*batch)
        DF.backward(DF.loss_total(ls))
        r1.finish()
        o1.step()
        ref.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    r1.remove()
    m2, r2, o2 = build()
    fs = ForkedStep(m2, batch, DF.loss_total, r2, o2)
    got = []
    for _ in range(3):
        ls = fs()
        got.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    # force the choice: make candidate `force` look fastest by patching the clock the probe uses
    real = time.perf_counter
    state = {"calls": 0}

    def fake():
        # capture() reads the clock twice per candidate: (t0, t1) -> candidate i's time = 1 s if it is the forced one, else 2 s
        k = state["calls"]
        state["calls"] += 1
        cand, second = divmod(k, 2)
        return float(cand * 10) + (0.0 if not second else (1.0 if cand == force else 2.0))
    time.perf_counter = fake
    try:
        fs.capture()
    finally:
        time.perf_counter = real
    skipped = fs.tuning_steps
    got += [None] * skipped
    for _ in range(n - 3 - skipped):
        ls = fs()
        got.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    torch.cuda.synchronize()
    bad = [(i, a, b) for i, (a, b) in enumerate(zip(got, ref)) if a is not None and a != b]
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    badk = [k for k in sd1 if not torch.equal(sd1[k], sd2[k])]
    r2.remove()
    return len(bad), (bad[0][0] if bad else None), len(badk), skipped, fs.probe_log


for rep in range(3):
    for dtype in (torch.bfloat16, torch.float32):
        for force in (0, 1):
            print(rep, dtype, "force", force, "->", run(dtype, force), flush=True)
