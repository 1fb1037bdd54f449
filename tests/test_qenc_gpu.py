"""Fused query encoder (drn_amd.functional._QueryEncoderFn: drn_qe_* + drn_lstm_step_* + drn_skinny_group + drn_outer_wgrad kernels) against
the oracle's QueryEncoder (oracle/drn_oracle.py, model/language_module.py:9-62) in fp64 on the CPU: the three attention
commands and the gradient of every parameter, including the dense embedding gradient with its zero padding row."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double()
    err = float((a - b).abs().max())
    # the softmax is shift invariant, so d/d(cmd_inter2logits.bias) is exactly 0: only rounding noise on both sides
    scale = max(float(b.abs().max()), 5e-2 if what.endswith("cmd_inter2logits.bias") else 1e-6)
    assert err <= tol * scale, "%s: %.3e > %.1e * %.3g" % (what, err, tol, scale)


@pytest.mark.parametrize("B,L,H,E,lens", [
    (6, 7, 64, 20, [7, 6, 4, 2, 1, 1]),
    (32, 8, 512, 300, None),
    (3, 12, 128, 300, [12, 12, 12]),
    (2, 1, 64, 24, [1, 1]),
])
def test_query_encoder_matches_oracle(B, L, H, E, lens):
    from drn_amd.model.language_module import QueryEncoder
    from oracle import drn_oracle as O
    V = 50
    g = torch.Generator().manual_seed(B * 100 + L)
    if lens is None:
        lens = sorted(torch.randint(1, L + 1, (B,), generator=g).tolist(), reverse=True)
        lens[0] = L
    lengths = torch.tensor(lens, dtype=torch.int64)
    tokens = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        tokens[b, :lens[b]] = torch.randint(1, V + 1, (lens[b],), generator=g)
    tokens[0, 0] = tokens[1, 0]                       # a word used by two clips: the embedding gradient must sum
    ref = O.QueryEncoder(V, hidden_dim=H, embed_dim=E).double()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) * (p.shape[-1] ** -0.5 if p.dim() > 1 else 0.1))
        ref.embedding.weight.mul_(E ** 0.5)            # unit-variance word vectors (GloVe-like)
        ref.embedding.weight[0].zero_()
    w = [torch.randn(B, 2 * H, generator=g, dtype=torch.float64) for _ in range(3)]
    cr = ref(tokens, lengths)
    sum((c * wi).sum() for c, wi in zip(cr, w)).backward()

    dev = "cuda:0"
    mod = QueryEncoder(V, hidden_dim=H, embed_dim=E)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(dev)
    ch = mod(tokens.to(dev), lengths.to(dev))
    assert len(ch) == 3 and all(c.shape == (B, 2 * H) for c in ch)
    sum((c * wi.float().to(dev)).sum() for c, wi in zip(ch, w)).backward()
    for t in range(3):
        _close(ch[t], cr[t], 3e-5, "cmd%d" % t)
    refp = dict(ref.named_parameters())
    for k, p in mod.named_parameters():
        if k.startswith("textualAttention"):
            assert p.grad is None                      # declared by the reference, never used
            continue
        _close(p.grad, refp[k].grad, 2e-4, k)
    assert float(mod.embedding.weight.grad[0].abs().max()) == 0.0          # padding_idx row
    unused = [v for v in range(1, V + 1) if v not in set(tokens.flatten().tolist())]
    assert float(mod.embedding.weight.grad[unused].abs().max()) == 0.0


def test_query_encoder_partial_command_gradients():
    """Only one of the three commands feeds the loss: the other two arrive as None in backward."""
    from drn_amd.model.language_module import QueryEncoder
    from oracle import drn_oracle as O
    B, L, H, E, V = 4, 5, 64, 20, 30
    g = torch.Generator().manual_seed(7)
    lengths = torch.tensor([5, 3, 2, 1])
    tokens = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        tokens[b, :lengths[b]] = torch.randint(1, V + 1, (int(lengths[b]),), generator=g)
    ref = O.QueryEncoder(V, hidden_dim=H, embed_dim=E).double()
    ref(tokens, lengths)[1].square().sum().backward()
    mod = QueryEncoder(V, hidden_dim=H, embed_dim=E)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to("cuda:0")
    mod(tokens.to("cuda:0"), lengths.to("cuda:0"))[1].square().sum().backward()
    refp = dict(ref.named_parameters())
    for k, p in mod.named_parameters():
        if refp[k].grad is None:
            continue
        _close(p.grad, refp[k].grad, 2e-4, k)


def test_query_encoder_rejects_overlong_queries():
    from drn_amd._lib import DrnError
    from drn_amd.model.language_module import QueryEncoder
    mod = QueryEncoder(20, hidden_dim=64, embed_dim=20).to("cuda:0")
    tokens = torch.ones(2, 65, dtype=torch.int64, device="cuda:0")
    with pytest.raises(DrnError):
        mod(tokens, torch.tensor([65, 65], device="cuda:0"))


@pytest.mark.parametrize("M,N,K,relu,bias", [(32, 512, 2048, True, True), (32, 3072, 512, False, True), (5, 16, 64, False, False),
                                             (64, 4096, 1024, False, True), (33, 48, 192, True, True), (32, 512, 3072, False, False)])
def test_skinny_linear(M, N, K, relu, bias):
    """drn_skinny_linear (exact-fp32 MFMA, K split over workgroups) against fp64."""
    from drn_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to("cuda:0")
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to("cuda:0")
    b = torch.randn(N, generator=g).to("cuda:0") if bias else None
    y = ops.skinny_linear(x, W, b, relu)
    want = x.double() @ W.double().t() + (b.double() if bias else 0)
    want = want.clamp_min(0) if relu else want
    _close(y, want.cpu(), 2e-6, "skinny")


@pytest.mark.parametrize("M,N,K", [(64, 300, 4096), (37, 48, 136), (16, 1024, 512)])
def test_skinny_group_bf16_rows(M, N, K):
    """drn_skinny_group with bf16 X rows (DrnSkinnyDesc.x_dtype): weights rounded to bf16 in registers, fp32 accumulation on the bf16
    MFMA, K split over workgroups as in the fp32 path -- against fp64 on the same (rounded) operands."""
    from drn_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(M + N)
    X = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    W = torch.randn(N, K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    y = ops.skinny_linear(X, W, bias)
    want = X.double() @ W.to(torch.bfloat16).double().t() + bias.double()
    _close(y, want.cpu(), 2e-5, "bf16-row product")


def test_gate_linear_function_matches_torch():
    from drn_amd import functional as DF
    g = torch.Generator().manual_seed(3)
    for N, K in [(4096, 1024), (500, 1024), (256, 1024)]:          # 500: not a multiple of 16 (ragged last column tile)
        lin = torch.nn.Linear(K, N).to("cuda:0")
        ref = torch.nn.Linear(K, N).double()
        ref.load_state_dict({k: v.double().cpu() for k, v in lin.state_dict().items()})
        x = torch.randn(32, K, generator=g)
        xr = x.double().requires_grad_()
        xh = x.to("cuda:0").requires_grad_()
        w = torch.randn(32, N, generator=g)
        (ref(xr) * w.double()).sum().backward()
        (DF.linear(xh, lin) * w.to("cuda:0")).sum().backward()
        _close(xh.grad, xr.grad, 1e-5, "dx")
        _close(lin.weight.grad, ref.weight.grad, 1e-5, "dW")
        _close(lin.bias.grad, ref.bias.grad, 1e-5, "db")


def test_skinny_group_ragged_and_masked():
    """drn_skinny_group: several problems of different shapes in one launch -- N not a multiple of 16, K = 300 (a multiple
    of 4 only), long-K ones (split over workgroups, last-arriver reduce) next to short ones, bias / ReLU / mask epilogues, a strided X -- against
    fp64, bit-identical from launch to launch; a short-K-only group as well."""
    from drn_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(11)
    shapes = [(32, 4096, 300, False, False, False), (64, 300, 4096, False, False, False), (32, 512, 3072, False, False, True),
              (7, 50, 2048, True, True, False), (32, 1024, 4096, True, False, False), (1, 1, 4, False, True, False)]
    probs, wants = [], []
    for M, N, K, bias, relu, mask in shapes:
        wide = torch.randn(M, K + 8, generator=g).to(dev)
        x = wide[:, 4:4 + K]                                        # row stride K + 8, 16-byte aligned start
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev) if bias else None
        mk = torch.randn(M, N, generator=g).to(dev) if mask else None
        want = x.double() @ W.double().t() + (b.double() if bias else 0)
        want = want.clamp_min(0) if relu else want
        if mask:
            want = want * (mk.double() > 0)
        probs.append(dict(X=x, W=W, bias=b, relu=relu, mask=mk))
        wants.append(want.cpu())
    first = None
    for rep in range(4):                                  # repeated: the K-split tiles' arrival counters re-arm themselves
        outs = ops.skinny_group(probs)
        for o, wnt, sh in zip(outs, wants, shapes):
            _close(o, wnt, 3e-6, "skinny_group %s rep %d" % (sh, rep))
        if first is None:
            first = [o.clone() for o in outs]
        assert all(torch.equal(a, b) for a, b in zip(first, outs)), "K-split reduce must be order-independent (deterministic)"
    assert int(ops._counters(torch.device(dev)).abs().sum()) == 0
    short = [0, 5]
    outs = ops.skinny_group([probs[i] for i in short])
    for o, i in zip(outs, short):
        _close(o, wants[i], 3e-6, "skinny_group short-K %s" % (shapes[i],))
    y = ops.skinny_rows(torch.randn(200, 300, generator=g).to(dev), probs[0]["W"])
    assert y.shape == (200, 4096)


@pytest.mark.parametrize("lowp", [False, True])
def test_outer_wgrad_group(lowp):
    """drn_outer_wgrad: dW = dY^T X and db = db2 = column sums for a group of problems (M = 32 and M = 256 rows, K = 300,
    column-sliced dY / X views, a bias-only problem) against fp64.  lowp (the bf16 model): the operands are rounded to bf16 on their
    way into the MFMA (two roundings of 2^-9, fp32 accumulation over M <= 256 rows); the column sums stay exact."""
    from drn_amd import ops
    tol_w = 2e-2 if lowp else 3e-6
    dev = "cuda:0"
    g = torch.Generator().manual_seed(12)
    probs, checks = [], []
    big = torch.randn(256, 4096, generator=g).to(dev)
    emb = torch.randn(256, 300, generator=g).to(dev)
    hp = torch.randn(256, 1024, generator=g).to(dev)
    for dY, X in [(big[:, :2048], emb), (big[:, 2048:], hp[:, 512:]), (torch.randn(32, 4096, generator=g).to(dev), torch.randn(32, 1024, generator=g).to(dev)),
                  (torch.randn(32, 50, generator=g).to(dev), torch.randn(32, 20, generator=g).to(dev)), (torch.randn(5, 1, generator=g).to(dev), None)]:
        N = dY.shape[1]
        dW = torch.full((N, X.shape[1]), float("nan"), device=dev) if X is not None else None
        db, db2 = torch.full((N,), float("nan"), device=dev), torch.full((N,), float("nan"), device=dev)
        probs.append(dict(dY=dY, X=X, dW=dW, db=db, db2=db2))
        checks.append((dY, X, dW, db, db2))
    probs.append(dict(dY=probs[2]["dY"], X=probs[2]["X"], dW=torch.empty(4096, 1024, device=dev)))      # no bias requested
    ops.outer_wgrad(probs, lowp=lowp)
    for dY, X, dW, db, db2 in checks:
        if X is not None:
            _close(dW, (dY.double().t() @ X.double()).cpu(), tol_w, "dW %s" % (tuple(dW.shape),))
        want = dY.double().sum(0).cpu()
        _close(db, want, 3e-6, "db")
        assert torch.equal(db, db2)
    _close(probs[-1]["dW"], (checks[2][0].double().t() @ checks[2][1].double()).cpu(), tol_w, "dW without bias")


def test_query_side_with_gate_projections_matches_oracle():
    """mainModel.encode_query: the query encoder and the three per-level gate projections qInput{t} (main_model.py:47-50)
    as ONE autograd node, against the oracle's modules in fp64 -- gates and every parameter gradient."""
    from drn_amd import functional as DF
    from drn_amd.model.language_module import QueryEncoder
    from oracle import drn_oracle as O
    B, L, H, E, V = 32, 8, 512, 300, 60
    Cs = [4096, 256, 512]
    g = torch.Generator().manual_seed(21)
    lens = sorted(torch.randint(3, L + 1, (B,), generator=g).tolist(), reverse=True)
    lens[0] = L
    lengths = torch.tensor(lens, dtype=torch.int64)
    tokens = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        tokens[b, :lens[b]] = torch.randint(1, V + 1, (lens[b],), generator=g)
    ref = O.QueryEncoder(V, hidden_dim=H, embed_dim=E).double()
    lin_r = [torch.nn.Linear(2 * H, c).double() for c in Cs]
    with torch.no_grad():
        for p in list(ref.parameters()) + [p for l in lin_r for p in l.parameters()]:
            p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) * (p.shape[-1] ** -0.5 if p.dim() > 1 else 0.1))
        ref.embedding.weight.mul_(E ** 0.5)
        ref.embedding.weight[0].zero_()
    w = [torch.randn(B, c, generator=g, dtype=torch.float64) for c in Cs]
    gr = [lin_r[t](c) for t, c in enumerate(ref(tokens, lengths))]
    sum((a * wi).sum() for a, wi in zip(gr, w)).backward()
    dev = "cuda:0"
    mod = QueryEncoder(V, hidden_dim=H, embed_dim=E)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(dev)
    lin_h = [torch.nn.Linear(2 * H, c).to(dev) for c in Cs]
    for lh, lr in zip(lin_h, lin_r):
        lh.load_state_dict({k: v.float() for k, v in lr.state_dict().items()})
    gh = DF.query_encoder(tokens.to(dev), lengths.to(dev), mod, lin_h)
    sum((a * wi.float().to(dev)).sum() for a, wi in zip(gh, w)).backward()
    for t in range(3):
        _close(gh[t], gr[t], 3e-5, "gate%d" % t)
        _close(lin_h[t].weight.grad, lin_r[t].weight.grad, 2e-4, "qInput%d.weight" % t)
        _close(lin_h[t].bias.grad, lin_r[t].bias.grad, 2e-4, "qInput%d.bias" % t)
    refp = dict(ref.named_parameters())
    for k, p in mod.named_parameters():
        if not k.startswith("textualAttention"):
            _close(p.grad, refp[k].grad, 2e-4, k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gate_projections_as_their_own_node_equal_the_fused_query_side(dtype):
    """mainModel.project_gates(encode_commands(...)) -- the cut the four-phase multi-GPU step makes so that the projections' gradients
    travel during the query encoder's backward -- against encode_query(...) (one node): gates and every parameter gradient of the
    query side, bit for bit."""
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    dev = "cuda:0"
    tok, qlen = [b.to(dev) for b in synthetic_batch(8, 32, 64, seed=3)[:2]]
    res = []
    for split in (False, True):
        m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 1)), compute_dtype=dtype)
        m.load_state_dict(seeded_state_dict(m, 0))
        m = m.to(dev).train()
        gates = m.project_gates(m.encode_commands(tok, qlen)) if split else m.encode_query(tok, qlen)
        g = torch.Generator(device=dev).manual_seed(5)
        loss = sum((gt * torch.randn(gt.shape, generator=g, device=dev)).sum() for gt in gates)
        loss.backward()
        torch.cuda.synchronize()
        res.append(([gt.detach().clone() for gt in gates],
                    {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    assert res[0][1].keys() == res[1][1].keys() and len(res[0][1]) > 20
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
