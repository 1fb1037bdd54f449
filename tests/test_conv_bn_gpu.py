"""conv -> BatchNorm(train) -> ReLU as ONE launch (drn_conv_bn_train) against the two-launch path it
replaces (drn_gemm_nt + drn_bn_train_apply): every output bit-identical -- raw conv output, normalised output, gated output,
scale/shift, (mean, invstd), running statistics -- for single blocks, level groups that share one BatchNorm module
(model/fcos.py:93-102), independent blocks (FPN output convs, model/FPN.py:56,69) and the FPN top-down chain
(model/FPN.py:63-68), at toy sizes (ragged tiles, the generic staging path) and at the benchmarked pyramid shapes.  The
reference semantics themselves (model/basic_blocks.py:9-31) are pinned by tests/test_functional_gpu.py, which now reaches
the one-launch kernel through DF.conv_block / DF.multi_conv_block wherever it applies."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _launches(fn):
    """Run fn() with the MFMA launch recorder on; -> (result, [tags])."""
    from drn_amd import ops
    ops.kernel_timer = []
    try:
        out = fn()
        torch.cuda.synchronize()
        tags = [t[0] for t in ops.kernel_timer]
    finally:
        ops.kernel_timer = None
    return out, tags


def _both(fn, may_decline=False):
    """fn() through the one-launch path and through the two-launch path; asserts the first really fused (may_decline: the
    launcher is allowed to decline -- the exact-f32 128x128 variant spills a few registers and is only trusted with one
    workgroup per CU, so a 448-workgroup f32 launch runs as two launches)."""
    from drn_amd import ops
    from drn_amd._lib import lib
    saved = ops.BN_FUSE
    # the two-launch side on the kernels whose slab statistics are summed in the fused kernel's order (the 128 x 128 / 256 x 256
    # general tiles): the 4-wave kernels (gemm_nt_w4h / w4c) sum theirs in another fixed order since round 6 -- equal to fp32 rounding,
    # which can move a normalised bf16 output by an ulp (tests/test_gemm_gpu.py pins those kernels)
    lib().drn_tune(b"nt_w4h", 0)
    lib().drn_tune(b"nt_w4c", 0)
    try:
        ops.BN_FUSE = True                 # (off by default: slower inside the step, see drn_amd/ops.py)
        fused, tags = _launches(fn)
        assert may_decline or any(t.endswith("+bn") for t in tags), "the one-launch kernel did not run: %s" % tags
        ops.BN_FUSE = False
        plain, tags2 = _launches(fn)
    finally:
        ops.BN_FUSE = saved
        lib().drn_tune(b"nt_w4h", 160)
        lib().drn_tune(b"nt_w4c", 1)
    assert not any(t.endswith("+bn") for t in tags2)
    assert ops.conv_bn_train_timeouts() == 0
    return fused, plain


def _same(a, b, what):
    assert a.shape == b.shape and a.dtype == b.dtype, what
    assert torch.equal(a, b), "%s differs: max |d| = %.3e" % (what, float((a.float() - b.float()).abs().max()))


def _mk_block(Cin, Cout, k, stride, seed, bias=False):
    conv = nn.Conv1d(Cin, Cout, k, stride=stride, padding=(k - 1) // 2, bias=bias).to(DEV)
    bn = nn.BatchNorm1d(Cout).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(rnd(Cout, Cin, k, seed=seed) / np.sqrt(Cin * k))
        if bias:
            conv.bias.copy_(rnd(Cout, seed=seed + 1) * 0.2)
        bn.weight.copy_(rnd(Cout, seed=seed + 2).abs() + 0.5)
        bn.bias.copy_(rnd(Cout, seed=seed + 3) * 0.3)
    return conv, bn


def _saved_of(t):
    """raw / scale-shift / save tensors of the autograd node behind an output (what backward will read)."""
    return [s for s in t.grad_fn.saved_tensors]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,L,Cin,Cout,k,stride,gate", [
    (3, 50, 64, 128, 3, 1, False),        # ragged last row tile (150 rows), one tile column
    (3, 50, 48, 256, 3, 2, True),         # generic (non-FAST) staging path, stride 2, gated second output
    (4, 64, 128, 384, 1, 1, True),        # three tile columns
    (32, 128, 256, 512, 3, 1, True),      # conv1 of the backbone at the benchmarked shape (gated)
])
def test_single_block_matches_two_launch_path(dt, B, L, Cin, Cout, k, stride, gate):
    from drn_amd import functional as DF
    x = rnd(B, L, Cin, seed=1).to(DEV, dt)
    g = (rnd(B, Cout, seed=2).abs() + 0.1).to(DEV) if gate else None

    def run():
        conv, bn = _mk_block(Cin, Cout, k, stride, seed=10, bias=True)
        xs = x.clone().requires_grad_()
        outs, gated = DF.conv_block([xs], conv, bn, True, dt, gate=g)
        return outs[0], gated, bn.running_mean.clone(), bn.running_var.clone(), _saved_of(outs[0])

    (o1, g1, rm1, rv1, sv1), (o2, g2, rm2, rv2, sv2) = _both(run)
    _same(o1, o2, "out")
    if gate:
        _same(g1, g2, "gated")
    _same(rm1, rm2, "running_mean")
    _same(rv1, rv2, "running_var")
    assert len(sv1) == len(sv2)
    for i, (a, b) in enumerate(zip(sv1, sv2)):
        _same(a, b, "saved tensor %d (raw / scale-shift / save)" % i)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Ls,C,N", [(2, (64, 32, 16), 64, 128), (32, (256, 128, 64), 512, 1024)])
def test_levels_sharing_one_module(dt, B, Ls, C, N):
    """The head towers: one conv + one BatchNorm applied per pyramid level -- statistics per level call, running statistics
    updated three times in level order (the 256x256 tile at the benchmarked shape)."""
    from drn_amd import functional as DF
    xs = [rnd(B, L, C, seed=3 + i).to(DEV, dt) for i, L in enumerate(Ls)]

    def run():
        conv, bn = _mk_block(C, N, 3, 1, seed=20, bias=True)
        outs, _ = DF.conv_block([x.clone().requires_grad_() for x in xs], conv, bn, True, dt)
        return outs, bn.running_mean.clone(), bn.running_var.clone(), _saved_of(outs[0])

    (o1, rm1, rv1, sv1), (o2, rm2, rv2, sv2) = _both(run)
    for l, (a, b) in enumerate(zip(o1, o2)):
        _same(a, b, "out level %d" % l)
    _same(rm1, rm2, "running_mean")
    _same(rv1, rv2, "running_var")
    for i, (a, b) in enumerate(zip(sv1, sv2)):
        _same(a, b, "saved tensor %d" % i)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("chain", [False, True])
@pytest.mark.parametrize("B,Ls,Cins,N,k", [(2, (64, 32, 16), (64, 128, 256), 128, 1), (3, (40, 20, 10), (64, 64, 64), 256, 3),
                                           (32, (256, 128, 64), (256, 512, 1024), 512, 1)])
def test_independent_blocks_and_the_top_down_chain(dt, chain, B, Ls, Cins, N, k):
    """FPN laterals (chain: out_l = relu(bn_l(conv_l(x_l))) + nearest_x2(out_{l+1})) and output convs (no chain)."""
    from drn_amd import functional as DF
    xs = [rnd(B, L, Ci, seed=5 + i).to(DEV, dt) for i, (L, Ci) in enumerate(zip(Ls, Cins))]

    def run():
        blocks = [_mk_block(Ci, N, k, 1, seed=30 + 7 * i) for i, Ci in enumerate(Cins)]
        outs = DF.multi_conv_block([x.clone().requires_grad_() for x in xs], blocks, True, dt, chain_up=chain)
        stats = [t.clone() for _, bn in blocks for t in (bn.running_mean, bn.running_var)]
        return outs, stats, _saved_of(outs[0])

    (o1, st1, sv1), (o2, st2, sv2) = _both(run, may_decline=(dt == torch.float32 and B == 32))
    for l, (a, b) in enumerate(zip(o1, o2)):
        _same(a, b, "out level %d" % l)
    for i, (a, b) in enumerate(zip(st1, st2)):
        _same(a, b, "running statistic %d" % i)
    for i, (a, b) in enumerate(zip(sv1, sv2)):
        _same(a, b, "saved tensor %d" % i)


def test_one_launch_is_deterministic_and_advances_its_generation(monkeypatch):
    """Ten launches in a row on the same inputs: the same bits every time, and the generation word that tags the statistics
    pairs moved on by exactly one per launch (a stale pair of an earlier launch can never pass for a fresh one)."""
    from drn_amd import functional as DF, ops
    dt = torch.bfloat16
    xs = [rnd(32, L, 512, seed=9 + i).to(DEV, dt) for i, L in enumerate((256, 128, 64))]
    conv, bn = _mk_block(512, 512, 3, 1, seed=40)
    first = None
    monkeypatch.setattr(ops, "BN_FUSE", True)
    DF.conv_block(xs, conv, bn, True, dt)
    ws, gen = ops._bn_tagged_ws(torch.device(DEV), 0)
    g0 = int(gen)
    for _ in range(10):
        outs, _ = DF.conv_block(xs, conv, bn, True, dt)
        if first is None:
            first = [o.clone() for o in outs]
        else:
            for a, b in zip(outs, first):
                assert torch.equal(a, b)
    torch.cuda.synchronize()
    assert ops.conv_bn_train_timeouts() == 0
    assert int(gen) == g0 + 10


def test_unsupported_launches_fall_back_to_two_launches(monkeypatch):
    """Different N per group / N not a multiple of 128: drn_conv_bn_train declines (nothing launched) and the caller runs the
    GEMM and the BatchNorm pass separately."""
    from drn_amd import functional as DF, ops
    monkeypatch.setattr(ops, "BN_FUSE", True)
    dt = torch.float32
    x = rnd(2, 32, 64, seed=1).to(DEV, dt)
    conv, bn = _mk_block(64, 64, 3, 1, seed=50)
    (outs, _), tags = _launches(lambda: DF.conv_block([x], conv, bn, True, dt))
    assert not any(t.endswith("+bn") for t in tags) and len(tags) == 1
    ref = torch.relu(torch.nn.functional.batch_norm(torch.nn.functional.conv1d(x.permute(0, 2, 1), conv.weight, padding=1), None, None,
                                                    bn.weight, bn.bias, True))
    assert float((outs[0].permute(0, 2, 1) - ref).abs().max()) < 1e-4


def test_waiting_launches_under_load_never_time_out(monkeypatch):
    """300 one-launch conv->BN blocks back to back at the benchmarked pyramid shapes (448 workgroups, two per CU; 224 of the
    256x256 tile) with and without the top-down chain: every launch fuses and none of their waits runs into the watchdog.
    (A version whose 512 threads all polled the statistics they merge flooded the fabric with L2-bypassing loads and starved the
    workgroups still computing: launches of more than one workgroup per CU timed out now and then.)"""
    from drn_amd import functional as DF, ops
    dt = torch.bfloat16
    B = 32
    xs = [rnd(B, L, Ci, seed=5 + i).to(DEV, dt) for i, (L, Ci) in enumerate(zip((256, 128, 64), (256, 512, 1024)))]
    lat = [_mk_block(Ci, 512, 1, 1, seed=60 + i) for i, Ci in enumerate((256, 512, 1024))]
    lvl = [_mk_block(512, 512, 3, 1, seed=70 + i) for i in range(3)]
    tower = _mk_block(512, 1024, 3, 1, seed=80)
    monkeypatch.setattr(ops, "BN_FUSE", True)
    ops.conv_bn_train_timeouts()
    with torch.no_grad():
        def step():
            inner = DF.multi_conv_block(xs, lat, True, dt, chain_up=True)
            outs = DF.multi_conv_block(inner, lvl, True, dt)
            return DF.conv_block(outs, tower[0], tower[1], True, dt)[0]
        _, tags = _launches(step)
        assert [t.endswith("+bn") for t in tags] == [True, True, True], tags
        first = [o.clone() for o in step()]
        for _ in range(100):
            last = step()
        torch.cuda.synchronize()
    assert ops.conv_bn_train_timeouts() == 0
    for a, b in zip(first, last):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Ls,C", [(2, (64, 32, 16), 128), (3, (40, 20), 64), (32, (256, 128, 64), 512)])
def test_top_down_chain_in_one_batchnorm_launch(dt, B, Ls, C):
    """drn_bn_train_apply with the whole FPN top-down chain in ONE launch (a level whose `up` is another level's output of the
    same launch recomputes the rows it adds from that level's raw rows and statistics; model/FPN.py:63-68) against one launch per
    level, coarse to fine: the same bits."""
    from drn_amd import ops
    code = ops.BF16 if dt == torch.bfloat16 else ops.F32
    n = len(Ls)
    raws = [(rnd(B, L, C, seed=90 + i) * (1.0 + i)).to(DEV, dt) for i, L in enumerate(Ls)]
    gam = [(rnd(C, seed=95 + i).abs() + 0.5).to(DEV) for i in range(n)]
    bet = [(rnd(C, seed=98 + i) * 0.3).to(DEV) for i in range(n)]

    def stats_of(raw):
        """per-128-row-slab (sum, M2) as the GEMM epilogue leaves them (from the stored values: any consistent numbers do)"""
        x = raw.float().reshape(-1, C)
        out = []
        for r0 in range(0, x.shape[0], 128):
            blk = x[r0:r0 + 128]
            out.append(torch.stack([blk.sum(0), ((blk - blk.mean(0)) ** 2).sum(0)]))
        return torch.stack(out).contiguous()
    sts = [stats_of(r) for r in raws]

    def run(one_launch):
        outs = [torch.empty_like(r) for r in raws]
        lvs = []
        for l in range(n):
            M = B * Ls[l]
            up = outs[l + 1] if l + 1 < n else None
            lvs.append(dict(raw=raws[l], ld_raw=C, out=outs[l], ld_out=C, M=M, L=Ls[l], up=up, ld_up=C if up is not None else 0,
                            stats=sts[l], tiles=sts[l].shape[0], ss=torch.empty(2, C, device=DEV), save=torch.empty(2, C, device=DEV),
                            gamma=gam[l], beta=bet[l], momentum=0.1, eps=1e-5,
                            running_mean=torch.zeros(C, device=DEV), running_var=torch.ones(C, device=DEV)))
        if one_launch:
            ops.bn_train_apply(lvs, C, code)
        else:
            for l in range(n - 1, -1, -1):
                ops.bn_train_apply([lvs[l]], C, code)
        torch.cuda.synchronize()
        return outs, [lv[k] for lv in lvs for k in ("ss", "save", "running_mean", "running_var")]
    (o1, s1), (o2, s2) = run(True), run(False)
    for l in range(n):
        _same(o1[l], o2[l], "out level %d" % l)
    for i, (a, b) in enumerate(zip(s1, s2)):
        _same(a, b, "statistic tensor %d" % i)
    # and against the definition, fp64
    ref_up = None
    for l in range(n - 1, -1, -1):
        x = raws[l].double().reshape(-1, C)
        y = torch.relu((x - x.mean(0)) / torch.sqrt(x.var(0, unbiased=False) + 1e-5) * gam[l].double() + bet[l].double()).reshape(B, Ls[l], C)
        if ref_up is not None:
            y = y + ref_up.repeat_interleave(2, dim=1)
        ref_up = y
        tol = 3e-2 if dt == torch.bfloat16 else 1e-4
        assert float((o1[l].double() - y).abs().max()) <= tol * max(1.0, float(y.abs().max())), l
