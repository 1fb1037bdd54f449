"""A hipGraph-replayed training step must produce the same parameters as the eager step (same kernels, same order)."""
import copy

import pytest
import torch

from helpers import isolated

pytestmark = pytest.mark.gpu


def make(seed_model=0):
    from drn_amd.dist import GradReducer
    from drn_amd.model import mainModel
    from drn_amd.optim import FusedAdam
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    dev = "cuda:0"
    m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=torch.bfloat16)
    m.load_state_dict(seeded_state_dict(m, seed_model))
    m = m.to(dev).train()
    params = [p for p in m.parameters() if p.requires_grad]
    red = GradReducer(params, world_size=1)
    opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
    batch = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]

    def step():
        red.zero()
        _, losses = m(*batch)
        sum(losses.values()).backward()
        red.finish()
        opt.step()
        return losses

    return m, step


def test_graph_replay_matches_eager():
    from drn_amd.graph import GraphedStep
    n = 6
    m1, step1 = make()
    eager_losses = []
    for _ in range(n):
        eager_losses.append(float(step1()["loss_cls"]))
    m2, step2 = make()
    g = GraphedStep(step2, warmup=2).capture()          # 2 eager warm-up steps ran; capture itself executes nothing
    graph_losses = []
    for _ in range(n - 2):
        graph_losses.append(float(g()["loss_cls"]))
    torch.cuda.synchronize()
    assert eager_losses[0] != eager_losses[-1], "training made no progress: %s" % eager_losses
    # replay k is training step 2+k
    for k, v in enumerate(graph_losses):
        assert abs(v - eager_losses[2 + k]) <= 2e-3 * max(1.0, abs(v)), (k, graph_losses, eager_losses)
    for (k1, p1), (k2, p2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        if p1.is_floating_point():
            assert torch.allclose(p1, p2, atol=2e-3, rtol=2e-3), (k1, float((p1 - p2).abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_dual_stream_step_is_bit_identical(dtype):
    """drn_amd.graph.DualStreamStep (seven linear hipGraphs on two streams, weight gradients deferred beside the query
    side's backward) only reorders independent launches: losses and EVERY parameter / buffer after n steps must equal the
    plain single-stream eager step bit for bit -- eagerly on the two streams (warm-up) and on replay."""
    from drn_amd.dist import GradReducer
    from drn_amd.graph import DualStreamStep
    from drn_amd.model import mainModel
    from drn_amd.optim import FusedAdam
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    dev = "cuda:0"

    def build():
        m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=dtype)
        m.load_state_dict(seeded_state_dict(m, 0))
        m = m.to(dev).train()
        red = GradReducer([p for p in m.parameters() if p.requires_grad], world_size=1, bucket_bytes=1 << 30,
                          adjacent=m.grad_stack_groups())
        return m, red, FusedAdam(red, lr=1e-3, max_norm=0.5)

    import drn_amd.functional as DF
    batch = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]
    n = 7
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
    ds = DualStreamStep(m2, batch, DF.loss_total, r2, o2)
    got = []
    for _ in range(3):                                       # eager, two streams
        ls = ds()
        got.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    ds.capture()
    for _ in range(n - 3):                                   # replays
        ls = ds()
        got.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    torch.cuda.synchronize()
    assert ref[0] != ref[-1], "training made no progress"
    assert got == ref, (got, ref)
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), (k, float((sd1[k].float() - sd2[k].float()).abs().max()))
    r2.remove()


def test_deferred_mode_graph_carries_foreign_gradients():
    """Multi-GPU graph mode (bench.py N>1): forward+backward+collect() replay as a hipGraph, the collectives and the
    optimizer run after it.  Gradients that reach the flat buckets through a copy (Scale parameters, the stacked tower
    parameters) must be this step's on EVERY replay, not the capture step's."""
    from drn_amd.dist import GradReducer
    from drn_amd.graph import GraphedStep
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    dev = "cuda:0"

    def build():
        m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=torch.float32)
        m.load_state_dict(seeded_state_dict(m, 0))
        return m.to(dev).train()

    bA = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]
    other = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=2)]
    bB = list(bA)                                            # same queries (token shapes are seed dependent) ...
    bB[2], bB[4] = other[2], other[4]                        # ... different clip features and ground truth
    assert all(a.shape == b.shape for a, b in zip(bA, bB))
    # reference: plain eager gradients on batch B
    mr = build()
    _, losses = mr(*bB)
    sum(losses.values()).backward()
    want = {k: p.grad.clone() for k, p in mr.named_parameters() if p.grad is not None}
    # deferred mode: capture on batch A, then replay on batch B (inputs refreshed in place)
    m = build()
    red = GradReducer([p for p in m.parameters() if p.requires_grad], world_size=1, overlap=False)
    static = [b.clone() for b in bA]

    def fwd_bwd():
        red.zero()
        _, ls = m(*static)
        sum(ls.values()).backward()
        red.collect()
        return ls

    g = GraphedStep(fwd_bwd, warmup=2).capture()
    g(); red.rearm(); red.finish()
    for s, b in zip(static, bB):
        s.copy_(b)
    g(); red.rearm(); red.finish()
    torch.cuda.synchronize()
    checked = 0
    for k, p in m.named_parameters():
        if k in want:
            ref = want[k]
            tol = 1e-4 * max(float(ref.abs().max()), 1e-6) + 1e-7
            assert float((p.grad - ref).abs().max()) <= tol, (k, float((p.grad - ref).abs().max()), tol)
            checked += 1
    assert checked > 40
    for k in ("fcos.head.scales.0.scale", "fcos.head.cls_tower.0.weight", "fcos.head.bbox_tower.1.weight"):
        assert k in want and float(want[k].abs().max()) > 0


@pytest.mark.parametrize("nphases", [3, 4])
@pytest.mark.parametrize("stage", [1, 3])
def test_two_phase_step_equals_single_backward(stage, nphases):
    """drn_amd.graph.TwoPhaseStep (trunk backward, input-stage backward, query-side backward: three hipGraphs sharing a
    pool, bucket groups [trunk, input, query]) must leave exactly the gradients of one plain backward in the flat buckets
    -- eagerly and on replay with refreshed inputs."""
    from drn_amd.dist import GradReducer
    from drn_amd.graph import TwoPhaseStep
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    dev = "cuda:0"

    def build():
        m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, stage)), compute_dtype=torch.float32)
        m.load_state_dict(seeded_state_dict(m, 0))
        m = m.to(dev).train()
        if stage == 1:
            for n, p in m.named_parameters():
                if "iou_scores" in n or "mix_fc" in n:
                    p.requires_grad_(False)
        return m

    loss_of = lambda ls: sum(ls.values())
    bA = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]
    other = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=2)]
    bB = list(bA)
    bB[2], bB[4] = other[2], other[4]
    mr = build()
    _, ls = mr(*bB)
    loss_of(ls).backward()
    want = {k: p.grad.clone() for k, p in mr.named_parameters() if p.grad is not None}

    m = build()
    assert set(map(id, m.trunk_parameters())) | set(map(id, m.front_parameters())) == set(map(id, m.parameters()))
    params = [p for p in m.parameters() if p.requires_grad]
    assert set(map(id, m.input_parameters())) | set(map(id, m.query_parameters())) == set(map(id, m.front_parameters()))
    if nphases == 4:      # round 5: the gate projections' gradients in a bucket of their own, exchanged during the encoder's backward
        live = lambda ps: [p for p in ps if p.requires_grad]
        assert set(map(id, m.gate_parameters())) | set(map(id, m.encoder_parameters())) == set(map(id, m.query_parameters()))
        red = GradReducer(params, world_size=1, overlap=False, bucket_bytes=1 << 30,
                          groups=[m.trunk_parameters(), m.input_parameters(), m.gate_parameters(), m.encoder_parameters()])
    else:
        red = GradReducer(params, world_size=1, overlap=False, bucket_bytes=1 << 30,
                          groups=[m.trunk_parameters(), m.input_parameters(), m.query_parameters()])
    assert len(red.group_buckets) == nphases and len(red.buckets) == nphases
    static = [b.clone() for b in bA[:5]]
    calls = []
    two = TwoPhaseStep(m, static, loss_of, red, between=lambda k: (calls.append(k), red.reduce(red.group_buckets[k])))

    def run():
        red.rearm()
        out = two()
        red.finish()
        return out

    def check(tag):
        torch.cuda.synchronize()
        n, bad = 0, []
        for k, p in m.named_parameters():
            if k in want:
                ref = want[k]
                tol = 1e-4 * max(float(ref.abs().max()), 1e-6) + 1e-7
                if not float((p.grad - ref).abs().max()) <= tol:
                    bad.append((k, float((p.grad - ref).abs().max()), tol))
                n += 1
        assert not bad, (tag, bad)
        assert n == len(want)

    for s, b in zip(static, bB[:5]):
        s.copy_(b)
    run()                                     # eager two-phase on batch B
    check("eager")
    for s, b in zip(static, bA[:5]):
        s.copy_(b)
    run()
    two.capture()                             # captured on batch A ...
    run()
    for s, b in zip(static, bB[:5]):
        s.copy_(b)
    l = run()                                 # ... replayed on batch B
    check("replay")
    assert two.NPHASES == nphases
    assert calls[-(nphases - 1):] == list(range(nphases - 1)) and len(calls) >= 4 * (nphases - 1) and torch.isfinite(l["loss_cls"]).all()


def test_adjacent_stack_groups_write_gradients_in_place():
    """GradReducer(adjacent=mainModel.grad_stack_groups()): the sources of the stacked tower conv lie back to back in the
    flat bucket, so the stacked weight gradient is written there directly (functional.grad_buffer) instead of being copied
    in afterwards.  Every parameter gradient must equal the default layout's, bit for bit (same kernels, same inputs)."""
    from drn_amd.dist import GradReducer
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    dev = "cuda:0"
    grads = []
    for adjacent in (False, True):
        m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=torch.bfloat16)
        m.load_state_dict(seeded_state_dict(m, 0))
        m = m.to(dev).train()
        params = [p for p in m.parameters() if p.requires_grad]
        groups = m.grad_stack_groups()
        assert groups and all(len(g) in (2, 3) for g in groups)      # tower pairs + the three Scale parameters
        red = GradReducer(params, world_size=1, adjacent=groups if adjacent else None)
        batch = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]
        red.zero()
        _, losses = m(*batch)
        sum(losses.values()).backward()
        if adjacent:       # the conv weights of the two towers: gradient views already inside the bucket, back to back
            a, b = groups[0]
            assert a.grad.data_ptr() + a.grad.numel() * 4 == b.grad.data_ptr()
            sink = red._of[a][0].views[red._of[a][1]]
            assert a.grad.data_ptr() == sink.data_ptr()
        red.finish()
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        red.remove()
    assert grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@isolated
def test_forked_graph_step_is_bit_identical(dtype):
    """drn_amd.graph.ForkedStep -- the whole step as ONE hipGraph with two branches (query side beside input preparation / weight
    gradients; bench.py's launch mode at N = 1 when it measures faster) -- against the plain single-stream eager step: losses
    and every parameter / buffer after n steps, bit for bit, through warm-up and 30 replays."""
    from drn_amd.dist import GradReducer
    from drn_amd.graph import ForkedStep
    from drn_amd.model import mainModel
    from drn_amd.optim import FusedAdam
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    import drn_amd.functional as DF
    dev = "cuda:0"

    def build():
        m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=dtype)
        m.load_state_dict(seeded_state_dict(m, 0))
        m = m.to(dev).train()
        red = GradReducer([p for p in m.parameters() if p.requires_grad], world_size=1, bucket_bytes=1 << 30,
                          adjacent=m.grad_stack_groups())
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
    fs.capture()                                             # times candidate side streams: spends `tuning_steps` real steps
    skipped = fs.tuning_steps
    assert 0 < skipped < n - 8
    got += [None] * skipped
    for _ in range(n - 3 - skipped):
        ls = fs()
        got.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    torch.cuda.synchronize()
    assert ref[0] != ref[-1], "training made no progress"
    bad = [(i, a, b) for i, (a, b) in enumerate(zip(got, ref)) if a is not None and a != b]
    if bad:                                                  # rare (profiles/HISTORY.md): say which side moved -- a second eager run decides
        r2.remove()
        m3, r3, o3 = build()
        again = []
        for _ in range(n):
            r3.zero()
            _, ls = m3(*batch)
            DF.backward(DF.loss_total(ls))
            r3.finish()
            o3.step()
            again.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
        e_bad = [i for i, (a, b) in enumerate(zip(again, ref)) if a != b]
        sd1, sd3 = m1.state_dict(), m3.state_dict()
        assert False, ("forked vs eager: %d steps differ, first %r; eager vs eager: %d steps differ (first %r), %d tensors differ; skipped %d"
                       % (len(bad), bad[:2], len(e_bad), e_bad[:1], sum(not torch.equal(sd1[k], sd3[k]) for k in sd1), skipped))
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
    r2.remove()


def test_forked_graph_replays_reproduce_themselves():
    """With lr = 0 every replay of the two-branch step computes the same losses and the same gradient bucket, so a replay that differs
    from the first one is a data race between the branches.  This is the amplifier that found the one behind the rare mismatch of the
    test above (scripts/experiments/forked_race_hunt.py; profiles/HISTORY.md, round 5): the K-split exchange of skinny_group_kernel
    let its ticket overtake a write-through partial when a bandwidth-bound kernel of the other branch ran beside it -- 12 events in
    180 k replays, none in 390 k since the stores are confirmed (qdense.hip).  8000 replays here (~8 s)."""
    from drn_amd.dist import GradReducer
    from drn_amd.graph import ForkedStep
    from drn_amd.model import mainModel
    from drn_amd.optim import FusedAdam
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    import drn_amd.functional as DF
    dev = "cuda:0"
    m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=torch.bfloat16)
    m.load_state_dict(seeded_state_dict(m, 0))
    m = m.to(dev).train()
    red = GradReducer([p for p in m.parameters() if p.requires_grad], world_size=1, bucket_bytes=1 << 30, adjacent=m.grad_stack_groups())
    opt = FusedAdam(red, lr=0.0, max_norm=0.5)
    batch = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]
    fs = ForkedStep(m, batch, DF.loss_total, red, opt)
    for _ in range(3):
        fs()
    fs.capture()
    keys = ("loss_cls", "loss_reg", "loss_iou")
    ls = fs()
    torch.cuda.synchronize()
    flat0 = [b.flat.clone() for b in red.buckets]
    loss0 = torch.stack([ls[k].reshape(-1)[0].float() for k in keys]).clone()
    moved = torch.zeros((), dtype=torch.int64, device=dev)
    for it in range(8000):
        ls = fs()
        cur = torch.stack([ls[k].reshape(-1)[0].float() for k in keys])
        moved += (cur != loss0).any().long()
        for b, f0 in zip(red.buckets, flat0):
            moved += (b.flat != f0).any().long()
    assert int(moved) == 0, "%d of 8000 replays differed from the first" % int(moved)
    red.remove()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_split_gate_input_stage(dtype):
    """mainModel.split_gate: the prop_fc GEMM un-gated + drn_gate_fwd as its own pass (model/backbone.py:28-30 for level 0).  In
    f32 the result is the fused epilogue's bit for bit (one multiplication either way); in bf16 the gated value is rounded from the
    rounded pre-gate value -- within one bf16 ulp of the fused one -- and the gradients follow."""
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    dev = "cuda:0"
    batch = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=2)]
    outs = []
    for split in (False, True):
        m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 1)), compute_dtype=dtype)
        m.load_state_dict(seeded_state_dict(m, 0))
        m = m.to(dev).train()
        m.split_gate = split
        g0, gates = m.forward_front(*batch[:4])
        g0.float().square().sum().backward()
        outs.append((g0.detach().float(), m.prop_fc.weight.grad.clone(), m.qInput0.weight.grad.clone()))
    (a, wa, qa), (b, wb, qb) = outs
    if dtype == torch.float32:
        assert torch.equal(a, b) and torch.equal(wa, wb) and torch.equal(qa, qb)
    else:
        assert float((a - b).abs().max()) <= 2.0 ** -7 * float(a.abs().max())
        assert float((wa - wb).norm()) <= 2e-2 * float(wa.norm()) and float((qa - qb).norm()) <= 2e-2 * float(qa.norm())


def test_forked_graph_at_the_benchmarked_shape_equals_the_linear_graph():
    """bench.py's two launch modes at N = 1 on the benchmarked workload (B = 32, T = 256, D = 4096, stage 1, bf16): the step
    replayed as one linear hipGraph and as one hipGraph with two branches (ForkedStep) leave the same losses and the same
    parameters, bit for bit, after the same number of steps."""
    import bench as B
    from drn_amd import dist as ddist, functional as DF
    from drn_amd.graph import ForkedStep, GraphedStep
    from drn_amd.model import mainModel
    from drn_amd.optim import FusedAdam
    from drn_amd.utils.synthetic import default_cfg, synthetic_batch
    dev = torch.device("cuda:0")
    cfg = default_cfg("C3D", 4096, 1)
    batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]

    def setup():
        m = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
        params = B.stage_params(m, 1)
        m.train()
        red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=m.grad_stack_groups(), bucket_bytes=1 << 30)
        return m, red, FusedAdam(red, lr=1e-3, max_norm=0.5)

    m1, r1, o1 = setup()

    def step():
        r1.zero()
        _, ls = m1(*batch)
        DF.backward(DF.loss_total(ls))
        r1.finish()
        o1.step()
        return ls
    lin = GraphedStep(step, warmup=2).capture()
    m2, r2, o2 = setup()
    fk = ForkedStep(m2, batch[:5], DF.loss_total, r2, o2).warm(2).capture()
    n_lin, n_fk = 2, 2 + fk.tuning_steps
    total = n_fk + 5
    for _ in range(total - n_lin):
        l1 = lin()
    for _ in range(total - n_fk):
        l2 = fk()
    torch.cuda.synchronize()
    for k in ("loss_cls", "loss_reg"):
        assert torch.equal(l1[k], l2[k]), (k, float(l1[k].reshape(-1)[0]), float(l2[k].reshape(-1)[0]))
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
    r1.remove()
    r2.remove()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@isolated
def test_forked_graph_optimizer_first_order_is_bit_identical(dtype):
    """ForkedStep with the query side's gradients in buckets of their own runs OPTIMIZER-FIRST (a call = pending update, then
    forward + backward; the query encoder's forward beside the Adam kernels): over a run -- prime, eager steps, capture, replays,
    flush -- the losses and every parameter / buffer equal the plain loop's (same reducer layout), bit for bit."""
    from drn_amd.dist import GradReducer
    from drn_amd.graph import ForkedStep
    from drn_amd.model import mainModel
    from drn_amd.optim import FusedAdam
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    import drn_amd.functional as DF
    dev = "cuda:0"

    def build():
        m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=dtype)
        m.load_state_dict(seeded_state_dict(m, 0))
        m = m.to(dev).train()
        params = [p for p in m.learned_parameters()]
        qset = set(id(p) for p in m.query_parameters())
        red = GradReducer(params, world_size=1, bucket_bytes=1 << 30, adjacent=m.grad_stack_groups(),
                          groups=[[p for p in params if id(p) in qset], [p for p in params if id(p) not in qset]])
        assert len(red.buckets) == 2
        return m, red, FusedAdam(red, lr=1e-4, max_norm=0.5)

    batch = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]
    n = 48
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
    fs = ForkedStep(m2, batch, DF.loss_total, r2, o2, rotate=True)          # (explicit opt-in since round 6)
    assert fs.rotate
    got = []
    for _ in range(3):                                        # call 1 = prime (gradients only), calls 2, 3 = update + gradients
        ls = fs()
        got.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    # parameters lag one update behind the plain loop until a flush
    fs.flush()
    torch.cuda.synchronize()
    p3 = {k: v.clone() for k, v in m2.state_dict().items()}
    fs.capture()                                              # re-primes (one more forward + backward), then times candidates
    skipped = fs.tuning_steps + 1
    got += [None] * skipped
    for _ in range(n - 3 - skipped):
        ls = fs()
        got.append([float(ls[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    fs.flush()
    torch.cuda.synchronize()
    assert ref[0] != ref[-1], "training made no progress"
    bad = [(i, a, b) for i, (a, b) in enumerate(zip(got, ref)) if a is not None and a != b]
    assert not bad, bad[:3]
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
    assert any(not torch.equal(p3[k], sd2[k]) for k in p3)
    r2.remove()


def test_forked_optimizer_first_at_the_benchmarked_shape_equals_the_plain_loop():
    """The launch mode bench.py reports at N = 1 (two branches, optimizer-first) on the benchmarked workload against the plain
    eager loop with the same two-bucket reducer: same losses, same parameters after the flush, bit for bit."""
    import bench as B
    from drn_amd import dist as ddist, functional as DF
    from drn_amd.graph import ForkedStep
    from drn_amd.model import mainModel
    from drn_amd.optim import FusedAdam
    from drn_amd.utils.synthetic import default_cfg, synthetic_batch
    dev = torch.device("cuda:0")
    cfg = default_cfg("C3D", 4096, 1)
    batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]

    def setup():
        m = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
        params = B.stage_params(m, 1)
        m.train()
        qset = set(id(p) for p in m.query_parameters())
        red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=m.grad_stack_groups(), bucket_bytes=1 << 30,
                                groups=[[p for p in params if id(p) in qset], [p for p in params if id(p) not in qset]])
        return m, red, FusedAdam(red, lr=1e-3, max_norm=0.5)

    m2, r2, o2 = setup()
    fk = ForkedStep(m2, batch[:5], DF.loss_total, r2, o2, rotate=True).warm(2).capture()
    assert fk.rotate
    for _ in range(5):
        l2 = fk()
    l2 = {k: v.clone() for k, v in l2.items()}
    fk.flush()
    n = 2 + fk.tuning_steps + 5
    m1, r1, o1 = setup()
    for _ in range(n):
        r1.zero()
        _, l1 = m1(*batch)
        DF.backward(DF.loss_total(l1))
        r1.finish()
        o1.step()
    torch.cuda.synchronize()
    for k in ("loss_cls", "loss_reg"):
        assert torch.equal(l1[k], l2[k]), (k, float(l1[k].reshape(-1)[0]), float(l2[k].reshape(-1)[0]))
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
    r1.remove()
    r2.remove()
