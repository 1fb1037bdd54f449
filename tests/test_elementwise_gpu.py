"""HBM-bound helper kernels of the path (drn_amd/csrc/elementwise.hip) against plain torch fp32/fp64 references."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DTS = [torch.float32, torch.bfloat16]


def _code(dt):
    from drn_amd import ops
    return ops.BF16 if dt == torch.bfloat16 else ops.F32


def _tol(dt):
    return dict(rtol=1e-5, atol=1e-5) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("with_add", [False, True])
@pytest.mark.parametrize("shape", [(3, 10, 24), (32, 256, 512), (2, 7, 1032)])
def test_gate_bwd(dt, with_add, shape):
    """backward of x = q[:, :, None] * x (model/backbone.py:28-30) + the per-clip column sums used for prop_fc's bias."""
    from drn_amd import ops
    B, L, C = shape
    g = torch.Generator().manual_seed(0)
    dev = "cuda:0"
    ld = C + 8                                            # dG is a column slice of a wider buffer
    dGw = torch.randn(B, L, ld, generator=g).to(dev).to(dt)
    act = torch.randn(B, L, C, generator=g).to(dev).to(dt)
    gate = torch.randn(B, C, generator=g).to(dev)
    add = torch.randn(B, L, C, generator=g).to(dev).to(dt) if with_add else None
    dC = torch.empty(B, L, C, device=dev, dtype=dt)
    dgate = torch.empty(B, C, device=dev)
    dsum = torch.empty(B, C, device=dev)
    ops.gate_bwd(dGw, ld, act, C, gate, dC, C, add, C, dgate, B, L, C, _code(dt), dsum=dsum)
    dG = dGw[:, :, :C].double()
    want_dC = dG * gate.double()[:, None, :] + (add.double() if with_add else 0)
    assert torch.allclose(dC.double(), want_dC, **_tol(dt))
    assert torch.allclose(dgate.double(), (dG * act.double()).sum(1), rtol=1e-4, atol=1e-4 * L ** 0.5)
    assert torch.allclose(dsum.double(), (dG * gate.double()[:, None, :]).sum(1), rtol=1e-4, atol=1e-4 * L ** 0.5)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("accumulate", [False, True])
def test_pairsum_add(dt, accumulate):
    """backward of F.interpolate(nearest, x2) + add (model/FPN.py:63-68)."""
    from drn_amd import ops
    B, L, C = 5, 12, 40
    g = torch.Generator().manual_seed(1)
    dev = "cuda:0"
    src = torch.randn(B, 2 * L, C, generator=g).to(dev).to(dt)
    dst0 = torch.randn(B, L, C, generator=g).to(dev).to(dt)
    dst = dst0.clone()
    ops.pairsum_add(dst, C, src, C, B * L, C, _code(dt), accumulate=accumulate)
    want = src.double().view(B, L, 2, C).sum(2) + (dst0.double() if accumulate else 0)
    assert torch.allclose(dst.double(), want, **_tol(dt))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,L2,C", [(3, 5, 40), (32, 64, 512)])
def test_pairsum_chain_in_one_launch_gives_the_bits_of_two(dt, B, L2, C):
    """both top-down backward steps of a three-level pyramid (model/FPN.py:63-68) in one launch (drn_pairsum_chain3): level-2
    threads recompute the level-1 rows they need, rounded where the two-launch order stores them."""
    from drn_amd import ops
    g = torch.Generator().manual_seed(2)
    dev = "cuda:0"
    d0 = torch.randn(B, 4 * L2, C, generator=g).to(dev).to(dt)
    own1 = torch.randn(B, 2 * L2, C, generator=g).to(dev).to(dt)
    own2 = torch.randn(B, L2, C, generator=g).to(dev).to(dt)
    a1, a2 = torch.empty_like(own1), torch.empty_like(own2)
    ops.pairsum_add_to(a1, C, own1, C, d0, C, B * 2 * L2, C, _code(dt))
    ops.pairsum_add_to(a2, C, own2, C, a1, C, B * L2, C, _code(dt))
    b1, b2 = torch.full_like(own1, float("nan")), torch.full_like(own2, float("nan"))
    ops.pairsum_chain3(d0, own1, b1, own2, b2, B * 2 * L2, C, _code(dt))
    torch.cuda.synchronize()
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    want1 = own1.double() + d0.double().view(B, 2 * L2, 2, C).sum(2)
    assert torch.allclose(b1.double(), want1, **_tol(dt))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,C", [(5, 8), (100, 256), (8192, 256), (9001, 264)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_pos_embed_bwd(dt, M, C, accumulate):
    """gradients of position_transform = nn.Linear(3, C) (model/main_model.py:34,51-55) from a column slice."""
    from drn_amd import ops
    g = torch.Generator().manual_seed(2)
    dev = "cuda:0"
    ld = C + 16
    dwide = torch.randn(M, ld, generator=g).to(dev).to(dt)
    feat = torch.rand(M, 3, generator=g).to(dev)
    dW0, db0 = torch.randn(C, 3, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    dW, db = dW0.clone(), db0.clone()
    ops.pos_embed_bwd(dwide[:, 16:], ld, feat, M, C, dW, db, _code(dt), accumulate=accumulate)
    d = dwide[:, 16:].double()
    want_W = d.t() @ feat.double() + (dW0.double() if accumulate else 0)
    want_b = d.sum(0) + (db0.double() if accumulate else 0)
    assert torch.allclose(dW.double(), want_W, rtol=1e-4, atol=1e-4 * M ** 0.5)
    assert torch.allclose(db.double(), want_b, rtol=1e-4, atol=1e-4 * M ** 0.5)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,L,Cout,P,k,stride", [(2, 8, 16, 8, 3, 1), (32, 256, 256, 256, 3, 1), (5, 50, 264, 40, 3, 2), (3, 33, 72, 12, 1, 1),
                                                 (40, 256, 64, 256, 3, 2)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_conv_tail_bwd(dt, B, L, Cout, P, k, stride, accumulate):
    """Position-embedding gradients through the conv that reads the embedding == the conv's input gradient on those channels
    (conv_transpose of dY with the weight slice) reduced against the row features (model/backbone.py:31-32, main_model.py:51-55)."""
    import torch.nn.functional as F
    from drn_amd import ops
    g = torch.Generator().manual_seed(5)
    dev = "cuda:0"
    pad = (k - 1) // 2
    Lo = (L + 2 * pad - k) // stride + 1
    dY = torch.randn(B, Lo, Cout, generator=g).to(dt)
    Cin = P + 24
    wd = (torch.randn(Cin, k, Cout, generator=g) / (k * Cout) ** 0.5).to(dt)          # the (Cin, k, Cout) copy
    feat = torch.rand(B * L, 3, generator=g)
    dW0, db0 = torch.randn(P, 3, generator=g), torch.randn(P, generator=g)
    dW, db = dW0.to(dev), db0.to(dev)
    dYd, wdd = dY.to(dev), wd.to(dev)
    ops.conv_tail_bwd(dYd, Cout, B, Lo, Cout, wdd[Cin - P:], k * Cout, k, stride, pad, feat.to(dev), L, P, dW, db, _code(dt),
                      accumulate=accumulate)
    w_oik = wd[Cin - P:].double().permute(2, 0, 1)                                        # (Cout, P, k)
    dx = F.conv_transpose1d(dY.double().permute(0, 2, 1), w_oik, stride=stride, padding=pad,
                            output_padding=L - ((Lo - 1) * stride - 2 * pad + k))         # (B, P, L)
    d = dx.permute(0, 2, 1).reshape(B * L, P)
    want_W = d.t() @ feat.double() + (dW0.double() if accumulate else 0)
    want_b = d.sum(0) + (db0.double() if accumulate else 0)
    tol = 1e-4                                             # (bf16 operands are exact inputs here; products and sums are fp32)
    scale = float(want_W.abs().max())
    assert float((dW.double().cpu() - want_W).abs().max()) <= tol * max(scale, 1.0)
    assert float((db.double().cpu() - want_b).abs().max()) <= tol * max(float(want_b.abs().max()), 1.0)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,C", [(1, 8), (32, 4096), (127, 24), (128, 24), (5000, 520)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_colsum(dt, M, C, accumulate):
    from drn_amd import ops
    g = torch.Generator().manual_seed(3)
    dev = "cuda:0"
    X = torch.randn(M, C, generator=g).to(dev).to(dt)
    out0 = torch.randn(C, generator=g).to(dev)
    out = out0.clone()
    ops.colsum(X, C, M, C, out, _code(dt), accumulate=accumulate)
    want = X.double().sum(0) + (out0.double() if accumulate else 0)
    assert torch.allclose(out.double(), want, rtol=1e-4, atol=1e-4 * M ** 0.5)


@pytest.mark.parametrize("dt", DTS)
def test_cast_and_pack(dt):
    from drn_amd import ops
    g = torch.Generator().manual_seed(4)
    dev = "cuda:0"
    x = torch.randn(1000003, generator=g).to(dev)
    assert torch.equal(ops.cast(x, _code(dt)), x.to(dt))
    for shape in [(20, 12, 3), (64, 64, 3), (260, 132, 3), (7, 5, 1), (128, 256, 1)]:
        w = torch.randn(shape, generator=g).to(dev)
        for perm in [(0, 2, 1), (1, 2, 0)]:
            want = w.permute(*perm).contiguous().to(dt)
            assert torch.equal(ops.pack_weight(w, perm, _code(dt)), want)
            out = torch.empty_like(want)
            ops.pack_weights_into([(w, perm, out)], _code(dt))
            assert torch.equal(out, want)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,K", [(8, 8), (64, 64), (72, 200), (8192, 4096), (1000, 24)])
def test_transpose2d(dt, M, K):
    from drn_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, K, generator=g).to("cuda:0").to(dt)
    assert torch.equal(ops.transpose2d(x, _code(dt)), x.t().contiguous())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,K", [(64, 64), (200, 72), (8192, 4096)])
def test_cast_transpose(dt, M, K):
    from drn_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(M, K, generator=g).to("cuda:0")
    a, b = ops.cast_transpose(x, _code(dt))
    assert torch.equal(a, x.to(dt)) and torch.equal(b, x.to(dt).t().contiguous())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(3, 32, 24), (32, 256, 512), (2, 64, 1032)])
def test_gate_bwd_transposed_output(dt, shape):
    """drn_gate_bwd_t: same reductions as drn_gate_bwd, the gated gradient written only as its transpose."""
    from drn_amd import ops
    B, L, C = shape
    g = torch.Generator().manual_seed(8)
    dev = "cuda:0"
    ld = C + 8
    dGw = torch.randn(B, L, ld, generator=g).to(dev).to(dt)
    act = torch.randn(B, L, C, generator=g).to(dev).to(dt)
    gate = torch.randn(B, C, generator=g).to(dev)
    dC = torch.empty(B, L, C, device=dev, dtype=dt)
    dCT = torch.empty(C, B * L, device=dev, dtype=dt)
    dgate, dsum, dgate2, dsum2 = (torch.empty(B, C, device=dev) for _ in range(4))
    ops.gate_bwd(dGw, ld, act, C, gate, dC, C, None, 0, dgate, B, L, C, _code(dt), dsum=dsum)
    ops.gate_bwd_t(dGw, ld, act, C, gate, dCT, dgate2, B, L, C, _code(dt), dsum=dsum2)
    assert torch.equal(dCT, dC.view(B * L, C).t().contiguous())
    assert torch.allclose(dgate2, dgate, rtol=1e-5, atol=1e-5) and torch.allclose(dsum2, dsum, rtol=1e-5, atol=1e-5)


def test_copy_multi_refills_step_inputs_in_one_launch():
    """drn_copy_multi: flat ranges of mixed dtypes / sizes (16-byte aligned and not), a zero-fill, and a token matrix padded out to a
    wider row -- against plain tensor copies."""
    from drn_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(1, 1000, (32, 6), generator=g).to(dev)                     # (B, L) -> (B, 8): two zero columns
    qlen = torch.randint(1, 7, (32,), generator=g).to(dev)
    feats = torch.randn(32, 32, 4096, generator=g).to(dev)
    pse = torch.rand(32, 32, 2, generator=g, dtype=torch.float64).to(dev)
    gt = torch.rand(32, 2, generator=g, dtype=torch.float64).to(dev)
    odd = torch.randn(1001, generator=g).to(dev)[1:]                                # 4000 bytes at a 4-byte offset: the byte path
    dst = [torch.full((32, 8), 7, dtype=torch.int64, device=dev), torch.full_like(qlen, 7), torch.full_like(feats, 7.0),
           torch.full_like(pse, 7.0), torch.full_like(gt, 7.0), torch.full((1000,), 7.0, device=dev), torch.full((513,), 7.0, device=dev)]
    ops.copy_multi([(dst[0], tok), (dst[1], qlen), (dst[2], feats), (dst[3], pse), (dst[4], gt), (dst[5], odd), (dst[6], None)])
    torch.cuda.synchronize()
    assert torch.equal(dst[0][:, :6], tok) and int(dst[0][:, 6:].abs().sum()) == 0
    for d, s_ in zip(dst[1:6], (qlen, feats, pse, gt, odd)):
        assert torch.equal(d, s_)
    assert float(dst[6].abs().sum()) == 0.0
    full = torch.randint(1, 1000, (32, 8), generator=g).to(dev)                    # no padding: a flat range
    ops.copy_multi([(dst[0], full)])
    torch.cuda.synchronize()
    assert torch.equal(dst[0], full)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_cast_transpose_throttled_equals_the_full_rate_pass(dt):
    """drn_cast_transpose_throttled (at most N workgroups resident, grid-stride loop over the tiles: the input cast when it runs beside
    the query encoder in the two-branch step) writes the same bits as drn_cast_transpose."""
    from drn_amd import ops
    code = ops.BF16 if dt == torch.bfloat16 else ops.F32
    x = torch.randn(1024, 520, device="cuda:0")
    a, aT = ops.cast_transpose(x, code)
    ops.CAST_THROTTLE = 7
    try:
        b, bT = ops.cast_transpose(x, code)
    finally:
        ops.CAST_THROTTLE = 0
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(aT, bT)
    assert torch.equal(aT, x.to(dt).t().contiguous())
