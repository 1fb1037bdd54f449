"""world_size-2 gloo tests of the data-parallel exchange step (drn_amd.dist.GradReducer) on CPU tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from drn_amd.dist import GradReducer, init_from_env
    init_from_env(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    unused = torch.nn.Linear(4, 4)                                    # never gets a gradient (like textualAttention)
    params = list(net.parameters()) + list(unused.parameters())
    red = GradReducer(params, world_size=world, bucket_bytes=256)     # several small buckets
    assert len(red.buckets) >= 3
    results = []
    for it in range(2):
        g = torch.Generator().manual_seed(100 * it + rank)
        x = torch.randn(5, 16, generator=g)
        red.zero()
        net(x).pow(2).sum().backward()
        red.finish()
        results.append([p.grad.clone() for p in params])
    # reference: average of both ranks' local gradients, computed independently on every rank
    for it in range(2):
        acc = None
        for r in range(world):
            g = torch.Generator().manual_seed(100 * it + r)
            x = torch.randn(5, 16, generator=g)
            net.zero_grad(set_to_none=True)
            for p in params:
                p.grad = None
            net(x).pow(2).sum().backward()
            gs = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in params]
            acc = gs if acc is None else [a + b for a, b in zip(acc, gs)]
        for a, b in zip(acc, results[it]):
            assert torch.allclose(a / world, b, atol=1e-6), (it, float((a / world - b).abs().max()))
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]


def test_grad_reducer_single_process_is_identity():
    from drn_amd.dist import GradReducer
    net = torch.nn.Linear(4, 3)
    red = GradReducer(net.parameters(), world_size=1)
    red.zero()
    net(torch.ones(2, 4)).sum().backward()
    red.finish()
    assert torch.allclose(net.weight.grad, torch.full((3, 4), 2.0))
    assert net.weight.grad.data_ptr() == red.buckets[0].flat.data_ptr() or net.bias.grad.data_ptr() == red.buckets[0].flat.data_ptr()


def _worker_two_phase(rank, world, port, q):
    """Deferred mode as bench.py N>1 uses it: bucket groups [trunk, front], backward in two parts with the trunk's
    all-reduce launched (reduce(group)) while the front's backward still runs, then finish()."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from drn_amd.dist import GradReducer, init_from_env
    init_from_env(backend="gloo")
    torch.manual_seed(0)
    front = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU())
    trunk = torch.nn.Sequential(torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    params = list(front.parameters()) + list(trunk.parameters())
    red = GradReducer(params, world_size=world, overlap=False, bucket_bytes=1 << 30,
                      groups=[list(trunk.parameters()), list(front.parameters())])
    assert [len(g) for g in red.group_buckets] == [1, 1]
    for it in range(2):
        g = torch.Generator().manual_seed(100 * it + rank)
        x = torch.randn(5, 16, generator=g)
        red.zero()
        red.rearm()
        h = front(x)
        hd = h.detach().requires_grad_()
        trunk(hd).pow(2).sum().backward()
        red.collect(red.group_buckets[0])
        red.reduce(red.group_buckets[0])                  # in flight while the front's backward runs
        assert red.group_buckets[0][0].launched and not red.group_buckets[1][0].launched
        torch.autograd.backward([h], [hd.grad])
        red.finish()
        got = [p.grad.clone() for p in params]
        acc = None
        for r in range(world):
            g = torch.Generator().manual_seed(100 * it + r)
            x = torch.randn(5, 16, generator=g)
            for p in params:
                p.grad = None
            trunk(front(x)).pow(2).sum().backward()
            gs = [p.grad.clone() for p in params]
            acc = gs if acc is None else [a + b for a, b in zip(acc, gs)]
        for a, b in zip(acc, got):
            assert torch.allclose(a / world, b, atol=1e-6), (it, float((a / world - b).abs().max()))
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_grad_reducer_groups_two_phase_world2_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_two_phase, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]


def _worker_four_groups(rank, world, port, q):
    """bench.py's N > 1 layout in miniature: four bucket groups reduced one after the other while the next part's backward runs."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from drn_amd.dist import GradReducer, init_from_env
    init_from_env(backend="gloo")
    torch.manual_seed(0)
    parts = [torch.nn.Linear(16, 24), torch.nn.Linear(24, 20), torch.nn.Linear(20, 12), torch.nn.Linear(12, 4)]     # encoder .. trunk
    params = [p for m in parts for p in m.parameters()]
    red = GradReducer(params, world_size=world, overlap=False, bucket_bytes=1 << 30, groups=[list(m.parameters()) for m in parts[::-1]])
    assert [len(g) for g in red.group_buckets] == [1, 1, 1, 1]
    for it in range(2):
        x = torch.randn(5, 16, generator=torch.Generator().manual_seed(100 * it + rank))
        red.zero()
        red.rearm()
        acts, cuts = [], []
        h = x
        for m in parts:
            h = m(h)
            acts.append(h)
            h = h.detach().requires_grad_()
            cuts.append(h)
        acts[-1].pow(2).sum().backward()                      # the last part (group 0)
        for k in range(4):
            red.collect(red.group_buckets[k])
            if k < 3:
                red.reduce(red.group_buckets[k])              # in flight while the next part's backward runs
                assert red.group_buckets[k][0].launched and not red.group_buckets[k + 1][0].launched
                torch.autograd.backward([acts[2 - k]], [cuts[2 - k].grad])
        tm = []
        red.finish(timings=tm)
        assert tm == []                                       # (CPU tensors: nothing to time with device events)
        got = [p.grad.clone() for p in params]
        acc = None
        for r in range(world):
            x = torch.randn(5, 16, generator=torch.Generator().manual_seed(100 * it + r))
            for p in params:
                p.grad = None
            torch.nn.Sequential(*parts)(x).pow(2).sum().backward()
            gs = [p.grad.clone() for p in params]
            acc = gs if acc is None else [a + b for a, b in zip(acc, gs)]
        for a, b in zip(acc, got):
            assert torch.allclose(a / world, b, atol=1e-6), (it, float((a / world - b).abs().max()))
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_grad_reducer_four_groups_world2_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_four_groups, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, "ok"), (1, "ok")]


def test_adjacent_groups_are_contiguous_in_the_bucket():
    """GradReducer(adjacent=[[a, b], ...]): the members of a group occupy consecutive slices of one flat bucket in the given
    order (no alignment padding between them), whatever their position in the parameter list; everything is still
    covered exactly once."""
    from drn_amd.dist import GradReducer
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(*shape)) for shape in [(8, 4), (3,), (16, 4, 3), (5,), (16, 4, 3), (12,), (12,)]]
    groups = [[ps[4], ps[2]], [ps[6], ps[5]]]
    red = GradReducer(ps, world_size=1, bucket_bytes=64, adjacent=groups)
    for a, b in groups:
        (ba, ia), (bb, ib) = red._of[a], red._of[b]
        assert ba is bb and ib == ia + 1
        assert ba.offsets[ia] + a.numel() == ba.offsets[ib]
    seen = [id(p) for b in red.buckets for p in b.params]
    assert sorted(seen) == sorted(id(p) for p in ps) and len(seen) == len(set(seen))
    red.remove()


# ---------------------------------------------------------------------------------------------------------------------
# Trainer under one process per GPU, torch-optimizer path (stage 2 / fused=False): gradients of ALL parameters are averaged
# before the all-parameter clip (drn_amd/trainer.py:_average_grads); every rank runs the same epochs; the sampler is told
# the epoch.
def _worker_trainer_stage2(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from drn_amd.dist import init_from_env
    from drn_amd import trainer as T
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    from oracle import drn_oracle as O
    init_from_env(backend="gloo")
    torch.set_num_threads(2)
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traj_s2.npz"))

    def model():
        m = O.mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", int(g["D"]), 2)))
        m.load_state_dict(seeded_state_dict(m, 0))
        return m

    # the reference trajectory's two batches (GT matched to a prediction, so tIoU > 0.9 positives exist and loss_iou has a
    # gradient); rank r takes batch (it + r) % 2
    two = []
    for i, seed in enumerate((1, 2)):
        b = list(synthetic_batch(int(g["B"]), int(g["T"]), int(g["D"]), seed=seed))
        b[4] = torch.from_numpy(g["gt%d" % i])
        two.append(b)
    batches = [[two[(it + r) % 2] for r in range(world)] for it in range(3)]
    m = model()
    tr = T.Trainer(m, 2, lr=1e-3, clip_gradient=0.5, world_size=world, fused=False)
    for it in range(3):
        tr.train_step(batches[it][rank])
    # single-process emulation: per step, average the two ranks' local gradients by hand (the frozen trunk's accumulate)
    e = model()
    params, lr, _, which = T.stage_plan(e, 2, 1e-3)
    opt = torch.optim.Adam(params, lr)
    opt.zero_grad()
    replicas = [model() for _ in range(world)]
    for it in range(3):
        for rep in replicas:
            rep.load_state_dict(e.state_dict())
            rep.train()
        local = []
        for r, rep in enumerate(replicas):
            rep.zero_grad(set_to_none=True)
            _, ld = rep(*batches[it][r])
            T.select_loss(ld, which).backward()
            local.append([p.grad for p in rep.parameters()])
        for i, p in enumerate(e.parameters()):
            gs = [l[i] for l in local if l[i] is not None]
            if gs:
                avg = sum(gs) / world
                p.grad = avg if p.grad is None else p.grad + avg
        e.load_state_dict(replicas[0].state_dict())            # rank 0's BN buffers (no SyncBN)
        torch.nn.utils.clip_grad_norm_(e.parameters(), 0.5)
        opt.step()
        opt.zero_grad()
    worst = 0.0
    for (n, a), b in zip(m.named_parameters(), e.parameters()):
        worst = max(worst, float((a - b).abs().max()))
    mine = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    other = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    same = all(torch.equal(o, mine) for o in other)
    q.put((rank, "ok" if (worst <= 2e-6 and same) else "mismatch worst=%g same=%s" % (worst, same)))
    dist.destroy_process_group()


def test_trainer_stage2_world2_averages_all_gradients():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_trainer_stage2, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert got == [(0, "ok"), (1, "ok")], got


def test_fit_train_only_runs_the_same_epochs_and_sets_sampler_epoch():
    from drn_amd import trainer as T

    class Sampler(object):
        def __init__(self):
            self.epochs = []

        def set_epoch(self, e):
            self.epochs.append(e)

    class Loader(list):
        pass

    tr = T.Trainer.__new__(T.Trainer)
    tr.default_epochs = None
    tr.graph = False
    seen = []
    tr.train_step = lambda args: seen.append(1)
    loader = Loader()
    loader.sampler = Sampler()
    hist = tr.fit_train_only(loader, 7, start_epoch=4)
    assert [h["epoch"] for h in hist] == [4, 5, 6] and loader.sampler.epochs == [4, 5, 6]
    tr.default_epochs = 10                                    # stage 1: fixed 10 epochs (main.py:126), resumed at 8
    assert [h["epoch"] for h in tr.fit_train_only(loader, 50, start_epoch=8)] == [8, 9]


# ---------------------------------------------------------------------------------------------------------------------
# Sharded evaluation (main.py:275-366 on one process): every rank scores its ShardSampler share of the test
# queries, the ranks merge with all_gather_object, all ranks return the single-process numbers.
def _mini_eval_setup(perturb=False):
    from torch.utils.data import DataLoader
    from drn_amd import trainer as T
    from drn_amd.data import CharadesSTA, ShardSampler, collate_data
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict
    from oracle import drn_oracle as O
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "charades_mini")
    cfg = default_cfg("TINY", 12, 3)
    cfg["feature_type"] = "C3D"
    cfg["C3D"] = {"feature_root": "./features", "feature_dim": 12, "ft_window_size": 16, "ft_overlap": 0.5}
    cfg["props_file_path"] = "./data/dataset/Charades/mini_props.txt"
    ds = torch.utils.data.ConcatDataset([CharadesSTA(cfg, "train", root, lambda s: s.split()), CharadesSTA(cfg, "test", root, lambda s: s.split())])
    ds = torch.utils.data.Subset(ds, list(range(len(ds) - 1 + len(ds) % 2)))      # an odd number of queries: the shards differ in size
    m = O.mainModel(VOCAB_SIZE, as_namespace(cfg))
    m.load_state_dict(seeded_state_dict(m, 0))
    with torch.no_grad():                                     # pass enough locations for NMS / top-k to matter
        m.fcos.head.cls_logits.bias.fill_(0.5)
    world_now = dist.get_world_size() if dist.is_initialized() else 1
    if perturb:
        # this rank built / resumed its model differently: other weights AND other BatchNorm running statistics.  The Trainer
        # must bring every rank to rank 0's replica (drn_amd.dist.sync_model_state; nn.DataParallel's broadcast, main.py:99)
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.05 * torch.randn(p.shape, generator=torch.Generator().manual_seed(99)))
            for b in m.buffers():
                if b.dtype.is_floating_point:
                    b.mul_(1.5).add_(0.1)
    tr = T.Trainer(m, 3, lr=1e-3, fused=False, world_size=world_now)
    if perturb:
        # ... and its BatchNorm running statistics drifted apart again during training (per-rank batches): evaluate() scores
        # every shard with rank 0's buffers, the replica whose state_dict fit() saves under the merged metric
        with torch.no_grad():
            for b in m.buffers():
                if b.dtype.is_floating_point:
                    b.mul_(0.7).sub_(0.05)
    loader = lambda world, rank: DataLoader(ds, batch_size=3, shuffle=False, collate_fn=collate_data,
                                            sampler=ShardSampler(ds, world, rank) if world > 1 else None)
    return tr, loader, len(ds)


def _worker_sharded_eval(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from drn_amd.dist import init_from_env
    init_from_env(backend="gloo")
    torch.set_num_threads(2)
    tr, loader, n = _mini_eval_setup(perturb=rank == 1)
    loss, topks, accs, results = tr.evaluate(loader(world, rank), iou_topk={"iou": [0.3, 0.5], "topk": [1, 5]})
    # after the pass every rank holds rank 0's buffers and parameters
    mine = torch.cat([t.detach().double().reshape(-1) for t in list(tr.model.parameters()) + list(tr.model.buffers())])
    other = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    assert all(torch.equal(o, mine) for o in other), "replicas differ after a sharded evaluation"
    q.put((rank, loss, topks, accs, sum(len(v) for v in results.values())))
    dist.destroy_process_group()


def test_sharded_evaluation_world2_equals_single_process():
    tr, loader, n = _mini_eval_setup()
    loss1, topks1, accs1, res1 = tr.evaluate(loader(1, 0), iou_topk={"iou": [0.3, 0.5], "topk": [1, 5]})
    assert sum(len(v) for v in res1.values()) == n and n % 2 == 1, "an odd number of queries: the shards differ in size"
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sharded_eval, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, loss, topks, accs, nrec in got:
        assert topks == topks1 and accs == accs1, (rank, accs, accs1)
        assert nrec == n
        # (the validation loss is normalised per BATCH -- n_pos + B, model/loss.py -- so it depends on how the queries fall into
        # batches and is not expected to be shard-invariant; both ranks must agree on the merged value)
        assert np.isfinite(loss) and loss == got[0][1]


def _worker_sync_state(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from drn_amd.dist import init_from_env, sync_model_state
    init_from_env(backend="gloo")
    torch.manual_seed(rank)                                   # every rank its own weights, statistics and step counter
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.BatchNorm1d(8), torch.nn.Linear(8, 3))
    net.train()
    for _ in range(rank + 1):
        net(torch.randn(5, 6))
    before = [p.detach().clone() for p in net.parameters()]
    sync_model_state(net, src=0, buffers_only=True)           # buffers follow rank 0, parameters stay
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
    assert int(net[1].num_batches_tracked) == 1
    bufs = torch.cat([b.detach().double().reshape(-1) for b in net.buffers()])
    got = [torch.empty_like(bufs) for _ in range(world)]
    dist.all_gather(got, bufs)
    assert all(torch.equal(g, got[0]) for g in got)
    sync_model_state(net, src=0)                              # the whole replica
    flat = torch.cat([t.detach().double().reshape(-1) for t in list(net.parameters()) + list(net.buffers())])
    got = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(got, flat)
    assert all(torch.equal(g, got[0]) for g in got)
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.BatchNorm1d(8), torch.nn.Linear(8, 3))
    assert all(torch.equal(a, b) for a, b in zip(ref.parameters(), net.parameters())), "not rank 0's parameters"
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_sync_model_state_world2_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sync_state, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, "ok"), (1, "ok")]


def test_adjacent_group_with_alignment_padding_takes_the_copy_path():
    """A stacked group whose members cannot lie back to back under _offsets' 32-element alignment is dropped from the adjacency
    plan (with a warning) instead of silently ending up apart."""
    import warnings
    from drn_amd.dist import GradReducer
    a, b = torch.nn.Parameter(torch.randn(1030)), torch.nn.Parameter(torch.randn(2048))      # 1030 % 32 != 0: padding before b
    c, d = torch.nn.Parameter(torch.randn(64, 32)), torch.nn.Parameter(torch.randn(64, 32))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        red = GradReducer([a, b, c, d], world_size=1, adjacent=[[a, b], [c, d]])
    assert any("adjacent group" in str(x.message) for x in w)
    (bc, ic), (bd, idd) = red._of[c], red._of[d]
    assert bc is bd and idd == ic + 1 and bc.offsets[ic] + c.numel() == bc.offsets[idd]
    red.remove()


def test_shard_sampler_covers_every_sample_once():
    from drn_amd.data import ShardSampler
    for n in (0, 1, 7, 8, 23):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in ShardSampler(range(n), world, r))
            assert seen == list(range(n))
            assert sum(len(ShardSampler(range(n), world, r)) for r in range(world)) == n
