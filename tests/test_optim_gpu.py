"""Fused clip+Adam HIP kernels vs torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (main.py:140,238-243)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_norm", [0.5, 1e9])
def test_fused_adam_matches_torch(max_norm):
    from drn_amd.dist import GradReducer
    from drn_amd.optim import FusedAdam
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    shapes = [(300, 64, 3), (64,), (1,), (4096, 33), (7,), (50000,), (2, 3, 5)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    red = GradReducer(pa, world_size=1, bucket_bytes=200 << 10)      # several buckets
    assert len(red.buckets) >= 2
    fused = FusedAdam(red, lr=1e-2, max_norm=max_norm)
    ref = torch.optim.Adam(pb, lr=1e-2)
    for it in range(4):
        grads = [torch.randn(s, generator=g).to(dev) * (0.1 + it) for s in shapes]
        red.zero()
        for p, q, gr in zip(pa, pb, grads):
            p.grad = gr.clone()          # finish() moves hand-set gradients into the flat bucket
            q.grad = gr.clone()
        red.finish()
        fused.step()
        tn = torch.nn.utils.clip_grad_norm_(pb, max_norm)
        ref.step()
        assert abs(float(fused.total_norm()) - float(tn)) <= 1e-4 * float(tn)
        for p, q in zip(pa, pb):
            assert torch.allclose(p, q, atol=2e-6, rtol=1e-5), float((p - q).abs().max())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_repack_all_refreshes_cached_weight_layouts_in_place(dt):
    """FusedAdam.step() ends with one drn_pack_weights launch over every cached GEMM-layout copy: same addresses,
    contents equal to a fresh permute+cast of the updated parameters (conv and linear weights, > 24 items)."""
    from drn_amd import functional as DF
    from drn_amd import ops
    from drn_amd.dist import GradReducer
    from drn_amd.optim import FusedAdam
    dev = "cuda:0"
    code = ops.BF16 if dt == torch.bfloat16 else ops.F32
    g = torch.Generator().manual_seed(1)
    shapes = [(24 + i, 16 + 8 * (i % 3), 3 if i % 2 else 1) for i in range(28)] + [(40, 72), (8, 1000)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    perms = [(0, 2, 1), (1, 2, 0)]
    copies = [(p, perm, DF.packed(p, perm, code)) for p in params for perm in (perms if p.dim() == 3 else perms[:1])]
    ptrs = [c.data_ptr() for _, _, c in copies]
    # fp32 concatenations (rewritten by the Adam kernel itself: same element order) and a transposed stack (repack_all)
    stack, stack_t = DF.stacked([params[0], params[6]]), DF.stacked_t([params[28]])
    sptr, tptr = stack.data_ptr(), stack_t.data_ptr()
    red = GradReducer(params, world_size=1)
    opt = FusedAdam(red, lr=1e-1, max_norm=1e9)
    twins = [torch.nn.Parameter(p.detach().clone()) for p in params]
    ref = torch.optim.Adam(twins, lr=1e-1)
    before = [p.detach().clone() for p in params]
    for it in range(3):
        red.zero()
        for p, q in zip(params, twins):
            p.grad = torch.randn(p.shape, generator=g).to(dev)
            q.grad = p.grad.clone()
        red.finish()
        opt.step()
        ref.step()
    # parameters with re-laid copies are updated tile by tile (drn_adam_tiled), the others linearly: same Adam either way
    assert any(st.get("tiled") is not None for st in opt.state)
    for p, q in zip(params, twins):
        assert torch.allclose(p, q, atol=5e-6, rtol=1e-5), float((p - q).abs().max())
    assert all(not torch.equal(a, p) for a, p in zip(before, params))
    for (p, perm, c), ptr in zip(copies, ptrs):
        again = DF.packed(p, perm, code)
        assert again.data_ptr() == ptr                       # cache hit, refreshed in place
        w3 = p.detach().unsqueeze(-1) if p.dim() == 2 else p.detach()
        want = w3.permute(*perm).contiguous().to(dt)
        want = want.view(want.shape[0], -1) if p.dim() == 2 else want
        assert torch.equal(again, want)
    again = DF.stacked([params[0], params[6]])
    assert again.data_ptr() == sptr and torch.equal(again, torch.cat([params[0].detach(), params[6].detach()]))
    again = DF.stacked_t([params[28]])
    assert again.data_ptr() == tptr and torch.equal(again, params[28].detach().t().contiguous())


def test_weight_copies_belong_to_their_model():
    """drn_amd.functional.WeightCopies: every mainModel owns the re-laid GEMM copies of its weights; an optimizer of model A
    refreshes A's copies and never looks at B's (whose copies go stale only when B's own parameters change)."""
    from drn_amd import functional as DF
    from drn_amd import ops
    from drn_amd.dist import GradReducer
    from drn_amd.model import mainModel
    from drn_amd.optim import FusedAdam
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict
    dev = "cuda:0"
    cfg = default_cfg("TINY", 64, 1)
    ma, mb = (mainModel(VOCAB_SIZE, as_namespace(cfg), compute_dtype=torch.bfloat16) for _ in range(2))
    for m in (ma, mb):
        m.load_state_dict(seeded_state_dict(m, 0))
        m.to(dev)
    assert ma.weight_copies is not mb.weight_copies and DF.store_of(ma.prop_fc.weight) is ma.weight_copies
    wa, wb = ma.backbone_net.forward_conv1[0].weight, mb.backbone_net.forward_conv1[0].weight
    ca, cb = DF.packed(wa, (0, 2, 1), ops.BF16), DF.packed(wb, (0, 2, 1), ops.BF16)
    assert len(ma.weight_copies.pack) == 1 and len(mb.weight_copies.pack) == 1
    gen_b = mb.weight_copies.gen
    red = GradReducer(ma.learned_parameters(), world_size=1)
    opt = FusedAdam(red, lr=1e-2, max_norm=0.0)
    assert opt.stores == [ma.weight_copies]
    red.zero()
    for p in red.params:
        p.grad = torch.randn_like(p)
    red.finish()
    opt.step()
    torch.cuda.synchronize()
    assert DF.packed(wa, (0, 2, 1), ops.BF16).data_ptr() == ca.data_ptr()            # refreshed in place
    assert torch.equal(ca, wa.detach().permute(0, 2, 1).contiguous().to(torch.bfloat16))
    assert mb.weight_copies.gen == gen_b and mb.weight_copies.epoch == 0               # B's store untouched
    assert DF.packed(wb, (0, 2, 1), ops.BF16).data_ptr() == cb.data_ptr()
    assert torch.equal(cb, wb.detach().permute(0, 2, 1).contiguous().to(torch.bfloat16))
    red.remove()
    del ma, opt, red
    import gc
    gc.collect()
    assert all(st is not None for st in DF.all_stores())


def test_sumsq_skip_ranges_and_external_partials():
    """drn_sumsq_partials_skip + drn_sumsq_finalize2: ranges left out of the pass (whole blocks, ragged edges, a range inside one
    block) and their squared sums handed in as external partial arrays give the norm of the whole buffer."""
    import ctypes
    from drn_amd import _lib
    dev = torch.device("cuda:0")
    L = _lib.lib()
    n = 4096 * 37 + 1234
    g = torch.randn(n, device=dev)
    ranges = [(100, 900), (4096 * 3, 4096 * 9), (4096 * 10 + 7, 4096 * 20 + 501), (n - 600, n)]
    nb = int(L.drn_opt_nblocks(ctypes.c_int64(n)))
    part = torch.full((nb,), float("nan"), device=dev)
    lo = (ctypes.c_int64 * len(ranges))(*[r[0] for r in ranges])
    hi = (ctypes.c_int64 * len(ranges))(*[r[1] for r in ranges])
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(L.drn_sumsq_partials_skip(P(g), ctypes.c_int64(n), P(part), None, lo, hi, len(ranges), None, st), "skip")
    host = (ctypes.c_ubyte * nb)()
    _lib.check(L.drn_sumsq_block_classes(ctypes.c_int64(n), lo, hi, len(ranges), host), "classes")
    cls = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(dev)
    assert sorted(set(cls.tolist())) == [0, 1, 2]
    part2 = torch.full((nb,), float("nan"), device=dev)
    _lib.check(L.drn_sumsq_partials_skip(P(g), ctypes.c_int64(n), P(part2), None, lo, hi, len(ranges), P(cls), st), "skip+classes")
    torch.cuda.synchronize()
    assert torch.equal(part, part2)
    ext1 = torch.stack([(g[a:b].double() ** 2).sum() for a, b in ranges[:2]]).float()
    ext2 = torch.stack([(g[a:b].double() ** 2).sum() for a, b in ranges[2:]]).float()
    tot = torch.zeros(1, device=dev)
    ext = (ctypes.c_void_p * 2)(ext1.data_ptr(), ext2.data_ptr())
    ext_n = (ctypes.c_int32 * 2)(2, 2)
    _lib.check(L.drn_sumsq_finalize2(P(part), nb, ext, ext_n, 2, P(tot), ctypes.c_float(0.5), st), "fin2")
    torch.cuda.synchronize()
    keep = torch.ones(n, dtype=torch.bool, device=dev)
    for a, b in ranges:
        keep[a:b] = False
    want_pass = float((g[keep].double() ** 2).sum())
    assert abs(float(part.double().sum()) - want_pass) <= 1e-5 * want_pass
    want = 0.25 * float((g.double() ** 2).sum())
    assert abs(float(tot) - want) <= 1e-5 * want


@pytest.mark.parametrize("dtype", [torch.bfloat16])
def test_norm_pass_with_producer_partials_equals_the_plain_pass(dtype, monkeypatch):
    """One step of the benchmarked model: with the squared sums of the prop_fc weight gradient (GEMM epilogue) and of the conv weight
    gradients (one-launch reduce) handed to the optimizer, the squared global norm equals the plain pass's within fp32 re-association,
    ranges were really left out, and the updated parameters agree."""
    import bench as B
    from drn_amd import dist as ddist, functional as DF, optim
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import default_cfg, synthetic_batch
    dev = torch.device("cuda:0")
    cfg = default_cfg("C3D", 4096, 1)
    batch = [b.to(dev) for b in synthetic_batch(32, 256, 4096, seed=1)]
    res = {}
    for ext in (False, True):
        monkeypatch.setattr(optim, "EXT_SUMSQ", ext)
        m = B.build(mainModel, cfg, dev, compute_dtype=dtype)
        params = B.stage_params(m, 1)
        m.train()
        red = ddist.GradReducer(params, world_size=1, overlap=True, adjacent=m.grad_stack_groups(), bucket_bytes=1 << 30)
        opt = optim.FusedAdam(red, lr=1e-3, max_norm=0.5)
        assert red.ext_sumsq == ext
        for _ in range(1):                       # (ONE step: the two variants' norms differ in the last bits, and training amplifies that)
            red.zero()
            _, ls = m(*batch)
            DF.backward(DF.loss_total(ls))
            red.finish()
            notes = list(red.sumsq_notes) if ext else []
            opt.step()
        torch.cuda.synchronize()
        res[ext] = (float(opt.total_sumsq), {k: v.detach().clone() for k, v in m.state_dict().items()}, notes)
        red.remove()
    notes = res[True][2]
    covered = sum(ne for nt in notes for _, ne in nt[0])
    assert len(notes) == 2 and covered > 25e6, (len(notes), covered)          # prop_fc.weight 16.8 M + the conv weights ~10 M elements
    a, b = res[False][0], res[True][0]
    assert abs(a - b) <= 1e-5 * a, (a, b)
    for k, v in res[False][1].items():
        w = res[True][1][k]
        if v.dtype.is_floating_point:
            assert float((v - w).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), k
