"""Fused query encoder (drn_amd.functional._QueryEncoderFn: drn_qe_* + drn_lstm_step_* kernels + library GEMMs) against
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


def test_gate_linear_function_matches_torch():
    from drn_amd import functional as DF
    g = torch.Generator().manual_seed(3)
    for N, K in [(4096, 1024), (500, 1024), (256, 1024)]:          # 500: not a multiple of 16 -> library path
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
