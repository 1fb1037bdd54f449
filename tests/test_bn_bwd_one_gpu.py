"""BatchNorm(train)+ReLU backward in ONE launch (drn_bn_bwd_one) against (a) the same formulas in fp64 on the host -- what
nn.BatchNorm1d + ReLU's autograd computes (model/basic_blocks.py:23-31) -- and (b) the two-launch path it replaces
(drn_bn_bwd_multi): same coefficients, partial sums over different row blocks, so agreement to rounding.  Covers one level and
level groups that share a module (dgamma / dbeta accumulated in level order, model/fcos.py:93-102), ragged row counts, strided
buffers, in-place (draw aliasing dout), both dtypes, the benchmarked pyramid shapes, run-to-run bit identity with the generation
word advancing, and the fall-back for grids the chip does not hold at once."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _levels(Ms, C, dt, seed, ld_extra=0, shared=True, inplace=False):
    """Device tensors of a launch: per level dout / raw [M, ld] (column slice of a wider buffer when ld_extra), scale_shift, save."""
    lv = []
    gamma = (rnd(C, seed=seed + 90) * 0.3 + 1.0).to(DEV)
    beta = (rnd(C, seed=seed + 91) * 0.2).to(DEV)
    for i, M in enumerate(Ms):
        ld = C + ld_extra
        raw = (rnd(M, ld, seed=seed + i) * 1.5 + 0.3).to(DEV).to(dt)
        dout = rnd(M, ld, seed=seed + 40 + i).to(DEV).to(dt)
        x = raw[:, :C].double()
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        g = gamma if shared or i == 0 else (rnd(C, seed=seed + 95 + i) * 0.3 + 1.0).to(DEV)
        sc = (g.double() * invstd).float()
        sh = (beta.double() - mean * g.double() * invstd).float()
        lv.append(dict(raw=raw, dout=dout, ss=torch.cat([sc, sh]).contiguous(), save=torch.cat([mean.float(), invstd.float()]).contiguous(),
                       gamma=g, M=M, ld=ld))
    return lv


def _expected(lv, C, relu, shared):
    """fp64: dgamma / dbeta per module and draw per level, from the values the kernels read (rounded inputs, fp32 scale/shift)."""
    draws, dg_all, db_all = [], [], []
    for v in lv:
        x = v["raw"][:, :C].double()
        g = v["dout"][:, :C].double()
        sc, sh = v["ss"][:C].double(), v["ss"][C:].double()
        mean, istd = v["save"][:C].double(), v["save"][C:].double()
        if relu:
            g = torch.where(torch.addcmul(sh, x, sc) > 0, g, torch.zeros_like(g))
        xhat = (x - mean) * istd
        db, dg = g.sum(0), (g * xhat).sum(0)
        M = x.shape[0]
        draws.append(v["gamma"].double() * istd * (g - db / M - xhat * dg / M))
        dg_all.append(dg)
        db_all.append(db)
    if shared:
        return draws, [sum(dg_all)], [sum(db_all)]
    return draws, dg_all, db_all


def _run(lv, C, dt, relu, shared, one, inplace=False):
    from drn_amd import ops
    code = ops.BF16 if dt == torch.bfloat16 else ops.F32
    n = len(lv)
    dgs = [torch.full((C,), float("nan"), device=DEV) for _ in range(1 if shared else n)]
    dbs = [torch.full((C,), float("nan"), device=DEV) for _ in range(1 if shared else n)]
    levels, draws = [], []
    for i, v in enumerate(lv):
        dout = v["dout"].clone()
        draw = dout if inplace else torch.zeros_like(dout)
        draws.append(draw)
        levels.append(dict(dout=dout[:, :C], ld_dout=v["ld"], raw=v["raw"][:, :C], ld_raw=v["ld"], ss=v["ss"], save=v["save"], gamma=v["gamma"],
                           draw=draw[:, :C], ld_draw=v["ld"], dgamma=dgs[0 if shared else i], dbeta=dbs[0 if shared else i],
                           accumulate=shared and i > 0, M=v["M"]))
    saved = ops.BN_BWD_ONE
    try:
        ops.BN_BWD_ONE = one
        ops.bn_bwd_multi(levels, C, code, relu=relu)
    finally:
        ops.BN_BWD_ONE = saved
    torch.cuda.synchronize()
    return [d[:, :C].clone() for d in draws], dgs, dbs, draws


def _plan_fits(lv, C, dt):
    from drn_amd import _lib, ops
    arr = (_lib.BnBwdDesc * len(lv))()
    for d, v in zip(arr, lv):
        d.M = v["M"]
    return int(_lib.lib().drn_bn_bwd_one_ws_bytes(arr, len(lv), C, ops.BF16 if dt == torch.bfloat16 else ops.F32)) > 0


CASES = [((200,), 64, 0, True), ((64, 32, 16), 128, 0, True), ((130, 67), 192, 64, False), ((8192, 4096, 2048), 512, 0, False),
         ((8192, 4096, 2048), 1024, 0, True), ((8192,), 256, 256, True), ((1024, 512, 256), 1024, 0, True), ((3,), 64, 0, True)]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("Ms,C,ld_extra,shared", CASES)
def test_one_launch_backward_matches_fp64_and_the_two_launch_path(dt, relu, Ms, C, ld_extra, shared):
    from drn_amd import ops
    lv = _levels(Ms, C, dt, seed=7, ld_extra=ld_extra, shared=shared)
    # (every bf16 case runs the one-launch kernel; in f32 -- 16 rows per pass -- the 14336 x 1024 case does not fit and falls back)
    assert _plan_fits(lv, C, dt) or (dt == torch.float32 and C == 1024 and sum(Ms) > 10000)
    want, dg_w, db_w = _expected(lv, C, relu, shared)
    got1, dg1, db1, full1 = _run(lv, C, dt, relu, shared, one=True)
    got2, dg2, db2, _ = _run(lv, C, dt, relu, shared, one=False)
    assert ops.bn_bwd_one_timeouts() == 0
    rows = sum(Ms)
    for a, b, w in zip(got1, got2, want):
        tol = 2e-2 if dt == torch.bfloat16 else 1e-4
        assert float((a.double() - w).abs().max()) <= tol * max(1.0, float(w.abs().max())), "draw vs fp64"
        # against the two-launch path: the coefficients agree to fp32 rounding; a bf16 output may flip its last bit
        d = (a.float() - b.float()).abs()
        assert float(d.max()) <= (float(w.abs().max()) * 2 ** -7 if dt == torch.bfloat16 else 1e-5 * max(1.0, float(w.abs().max())))
    for a, b, w in zip(dg1 + db1, dg2 + db2, dg_w + db_w):
        scale = max(1.0, float(w.abs().max()))
        assert float((a.double() - w).abs().max()) <= 3e-5 * scale * max(1.0, rows ** 0.5 / 16), "dgamma / dbeta vs fp64"
        assert float((a - b).abs().max()) <= 3e-5 * scale * max(1.0, rows ** 0.5 / 16)
    if ld_extra:                                      # columns past C belong to somebody else
        for f in full1:
            assert float(f[:, C:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_in_place_and_bit_identical_from_run_to_run(dt):
    from drn_amd import ops
    Ms, C = (8192, 4096, 2048), 512
    lv = _levels(Ms, C, dt, seed=3)
    a, dga, dba, _ = _run(lv, C, dt, True, True, one=True)
    tws = [v for k, v in ops._persistent.items() if isinstance(k[0], tuple) and k[0][0] == "bn_bwd_one"]      # (keyed per stream)
    assert tws, "the one-launch path did not run"
    gens = [int(t[0].item()) for t in tws]
    b, dgb, dbb, _ = _run(lv, C, dt, True, True, one=True, inplace=True)
    assert sum(int(t[0].item()) for t in tws) == sum(gens) + 1, "one launch = one generation"
    for x, y in zip(a + dga + dba, b + dgb + dbb):
        assert torch.equal(x, y)
    assert ops.bn_bwd_one_timeouts() == 0


def test_grids_the_chip_does_not_hold_at_once_take_two_launches():
    from drn_amd import _lib, ops
    L = _lib.lib()
    C, dt = 1024, torch.bfloat16
    lv = _levels((40000,), C, dt, seed=5)             # 79 row blocks of 512 rows > 64, 1264 workgroups > 512
    assert not _plan_fits(lv, C, dt)
    arr = (_lib.BnBwdDesc * 1)()
    arr[0].M = 40000
    tws = torch.zeros(1 << 20, dtype=torch.int64, device=DEV)
    rc = L.drn_bn_bwd_one(arr, 1, C, 1, ctypes.c_void_p(tws.data_ptr()), ctypes.c_int64(tws.numel() * 8), ops.BF16, None)
    assert rc == ops.DRN_ERR_UNSUPPORTED and b"drn_bn_bwd_multi" in L.drn_last_error()
    want, dg_w, db_w = _expected(lv, C, True, True)
    got, dg, db, _ = _run(lv, C, dt, True, True, one=True)        # falls back inside ops.bn_bwd_multi
    assert float((got[0].double() - want[0]).abs().max()) <= 2e-2 * max(1.0, float(want[0].abs().max()))
    # a smaller workgroup budget moves a launch that fits onto bigger row blocks -- or out
    lv2 = _levels((8192, 4096, 2048), 512, dt, seed=5)
    assert _plan_fits(lv2, 512, dt)
    try:
        L.drn_tune(b"bn1_maxwg", 100)
        assert not _plan_fits(lv2, 512, dt)
    finally:
        L.drn_tune(b"bn1_maxwg", 512)


def test_training_step_gradients_agree_with_the_two_launch_path():
    """The whole backward pass with and without the one-launch kernel: every gradient to bf16-path rounding."""
    import bench as B
    from drn_amd import functional as DF, ops
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import default_cfg, synthetic_batch
    dev = torch.device(DEV)
    cfg = default_cfg("C3D", 512, 1)
    batch = [b.to(dev) for b in synthetic_batch(4, 64, 512, seed=1)]
    grads = []
    for one in (True, False):
        torch.manual_seed(0)
        m = B.build(mainModel, cfg, dev, compute_dtype=torch.bfloat16)
        m.train()
        saved = ops.BN_BWD_ONE
        try:
            ops.BN_BWD_ONE = one
            _, ls = m(*batch)
            DF.backward(DF.loss_total(ls))
        finally:
            ops.BN_BWD_ONE = saved
        torch.cuda.synchronize()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert ops.bn_bwd_one_timeouts() == 0
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 20
    for n in grads[0]:
        a, b = grads[0][n].double(), grads[1][n].double()
        assert float((a - b).norm()) <= 2e-2 * max(float(b.norm()), 1e-6), n


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,L,C,with_add", [(32, 128, 512, True), (32, 64, 1024, True), (32, 256, 256, True), (8, 32, 128, False),
                                            (4, 48, 64, True), (32, 128, 512, False)])
def test_query_gate_backward_inside_the_batchnorm_launch(dt, B, L, C, with_add):
    """DrnBnBwdDesc::gb_*: drn_gate_bwd's work (model/backbone.py:28-30 backward: dout = add + dG * gate[clip], dgate = sum_t dG * out)
    done by drn_bn_bwd_one on the rows it loads -- against the two launches: the BatchNorm input gradient bit for bit when both use
    the same row blocks, everything else to rounding; clips of one / two / half a row block, with and without the un-gated gradient,
    and a clip length the kernel cannot take (48 rows: ops.bn_bwd_multi then launches drn_gate_bwd itself)."""
    from drn_amd import ops
    code = ops.BF16 if dt == torch.bfloat16 else ops.F32
    M = B * L
    lv = _levels((M,), C, dt, seed=11)[0]
    raw, ss, save, gamma = lv["raw"], lv["ss"], lv["save"], lv["gamma"]
    dG = rnd(M, C, seed=21).to(DEV).to(dt)
    add = rnd(M, C, seed=22).to(DEV).to(dt) if with_add else None
    gate = (rnd(B, C, seed=23) * 0.5 + 1.0).to(DEV)

    def run(fused):
        dgate = torch.full((B, C), float("nan"), device=DEV)
        dgamma, dbeta = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        draw = torch.empty((M, C), device=DEV, dtype=dt)
        if fused:
            level = dict(dout=add, ld_dout=C, raw=raw, ld_raw=C, ss=ss, save=save, gamma=gamma, draw=draw, ld_draw=C, dgamma=dgamma,
                         dbeta=dbeta, accumulate=False, M=M,
                         gb=dict(dg=dG, ld_dg=C, gate=gate, ldg=C, dgate=dgate, L=L, act=act_lib, ld_act=C))
        else:
            d = torch.empty((M, C), device=DEV, dtype=dt)
            ops.gate_bwd(dG, C, act_lib, C, gate, d, C, add, C, dgate, B, L, C, code)
            level = dict(dout=d, ld_dout=C, raw=raw, ld_raw=C, ss=ss, save=save, gamma=gamma, draw=draw, ld_draw=C, dgamma=dgamma,
                         dbeta=dbeta, accumulate=False, M=M)
        ops.bn_bwd_multi([level], C, code, relu=True)
        torch.cuda.synchronize()
        return draw, dgate, dgamma, dbeta
    # the stored activation, produced the way the forward does: relu(fma(raw, scale, shift)) rounded to dt -- through drn_bn_apply
    act_lib = torch.empty((M, C), device=DEV, dtype=dt)
    ops.bn_apply(raw, C, ss, act_lib, C, M, C, L, code, relu=True)
    a = run(True)
    b = run(False)
    assert ops.bn_bwd_one_timeouts() == 0
    scale = max(1.0, float(b[0].float().abs().max()))
    tol = 2.0 ** -7 if dt == torch.bfloat16 else 2e-5
    assert float((a[0].float() - b[0].float()).abs().max()) <= tol * scale, "draw"
    ref = (dG.double() * act_lib.double()).view(B, L, C).sum(1)
    assert float((a[1].double() - ref).abs().max()) <= 3e-5 * max(1.0, float(ref.abs().max())) * max(1.0, L ** 0.5 / 4), "dgate vs fp64"
    assert float((a[1] - b[1]).abs().max()) <= 3e-5 * max(1.0, float(ref.abs().max())) * max(1.0, L ** 0.5 / 4)
    for x, y in zip(a[2:], b[2:]):
        assert float((x - y).abs().max()) <= 3e-5 * max(1.0, float(y.abs().max())) * max(1.0, M ** 0.5 / 16)
