"""world_size-2 gloo tests of the data-parallel exchange step (drn_amd.dist.GradReducer) on CPU tensors."""
import os
import socket

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
