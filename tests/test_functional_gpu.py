"""GPU parity of each fused autograd stage (drn_amd.functional) against plain PyTorch fp64 on the same inputs:
forward values AND gradients, stage by stage, so no ReLU sign flip of an upstream layer can leak in.
fp32 compute: 2e-5 relative to the tensor scale; bf16 compute: 3e-2."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float32: 3e-5, torch.bfloat16: 4e-2}


def close(got, ref, tol, what, scale=None):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    err = float((got - ref).abs().max())
    scale = max(float(ref.abs().max()), 1e-3) if scale is None else scale
    assert err <= tol * scale, "%s: max|err| %.3e > %.1e * %.3g" % (what, err, tol, scale)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64) * scale


def ref_block(x, w, gamma, beta, stride, dt, eps=1e-5):
    """fp64 conv -> train-mode BN -> ReLU.  In bf16 mode the conv output is rounded to bf16 (straight-through
    gradient) exactly where the kernel stores it, with statistics taken before rounding as the GEMM epilogue
    does, so both sides take the same ReLU decisions and the comparison is well-posed."""
    raw = F.conv1d(x, w, stride=stride, padding=(w.shape[2] - 1) // 2)
    mean = raw.mean(dim=(0, 2), keepdim=True)
    var = raw.var(dim=(0, 2), unbiased=False, keepdim=True)
    rq = raw + (raw.bfloat16().double() - raw).detach() if dt == torch.bfloat16 else raw
    y = (rq - mean) / torch.sqrt(var + eps) * gamma[None, :, None] + beta[None, :, None]
    n = raw.numel() / raw.shape[1]
    return F.relu(y), mean.reshape(-1), var.reshape(-1) * n / (n - 1)


def q(t, dt):
    return t.bfloat16().double() if dt == torch.bfloat16 else t


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,stride,gate,up", [(3, 1, False, False), (3, 2, True, False), (1, 1, False, True), (3, 1, True, False)])
def test_conv_block_fwd_bwd(dt, k, stride, gate, up):
    from drn_amd import functional as DF
    B, L, Cin, Cout = 3, 32, 64, 128
    w = q(rnd(Cout, Cin, k, seed=11) / np.sqrt(Cin * k), dt).requires_grad_()
    gamma = (rnd(Cout, seed=1).abs() + 0.5).requires_grad_()
    beta = (rnd(Cout, seed=2) * 0.3).requires_grad_()
    x = q(rnd(B, Cin, L, seed=3), dt)
    Lo = (L + 2 * ((k - 1) // 2) - k) // stride + 1
    qg = rnd(B, Cout, seed=4) if gate else None
    u = q(rnd(B, Cout, Lo // 2, seed=5), dt) if up else None
    # ---- reference (fp64)
    xr = x.clone().requires_grad_()
    qr = qg.clone().requires_grad_() if gate else None
    ur = u.clone().requires_grad_() if up else None
    y, mean, uvar = ref_block(xr, w, gamma, beta, stride, dt)
    if up:
        y = y + ur.repeat_interleave(2, dim=-1)
    w1, w2 = rnd(B, Cout, Lo, seed=6), rnd(B, Cout, Lo, seed=7)
    loss = (y * w1).sum()
    if gate:
        loss = loss + ((y * qr[:, :, None]) * w2).sum()
    loss.backward()
    # ---- HIP
    conv_h = nn.Conv1d(Cin, Cout, k, stride=stride, padding=(k - 1) // 2, bias=False).to(DEV)
    bn_h = nn.BatchNorm1d(Cout).to(DEV)
    with torch.no_grad():
        conv_h.weight.copy_(w.float()); bn_h.weight.copy_(gamma.float()); bn_h.bias.copy_(beta.float())
    xh = x.permute(0, 2, 1).contiguous().to(DEV, dt).requires_grad_()
    qh = qg.float().to(DEV).requires_grad_() if gate else None
    uh = u.permute(0, 2, 1).contiguous().to(DEV, dt).requires_grad_() if up else None
    outs, gated = DF.conv_block([xh], conv_h, bn_h, True, dt, gate=qh, up=uh)
    lh = (outs[0].float() * w1.permute(0, 2, 1).float().to(DEV)).sum()
    if gate:
        lh = lh + (gated.float() * w2.permute(0, 2, 1).float().to(DEV)).sum()
    lh.backward()
    tol = TOL[dt]
    close(outs[0].permute(0, 2, 1), y, tol, "out")
    if gate:
        close(gated.permute(0, 2, 1), y * qg[:, :, None], tol, "gated")
        close(qh.grad, qr.grad, tol * 3, "dgate")
    if up:
        close(uh.grad.permute(0, 2, 1), ur.grad, tol, "dup")
    close(xh.grad.permute(0, 2, 1), xr.grad, tol * 3, "dx")
    close(conv_h.weight.grad, w.grad, tol * 3, "dW")
    close(bn_h.weight.grad, gamma.grad, tol * 3, "dgamma")
    close(bn_h.bias.grad, beta.grad, tol * 3, "dbeta")
    close(bn_h.running_mean, 0.1 * mean, tol, "running_mean")
    close(bn_h.running_var, 0.9 + 0.1 * uvar, tol, "running_var")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_conv_block_three_levels_shared(dt):
    """Shared module applied per level: per-level batch statistics, running stats updated 3x in order (model/fcos.py:93-102)."""
    from drn_amd import functional as DF
    B, C = 2, 64
    w = q(rnd(C, C, 3, seed=30) / np.sqrt(3 * C), dt).requires_grad_()
    cb = rnd(C, seed=31) * 0.2
    gamma, beta = torch.ones(C, dtype=torch.float64, requires_grad=True), torch.zeros(C, dtype=torch.float64, requires_grad=True)
    xs = [q(rnd(B, C, L, seed=10 + i), dt) for i, L in enumerate((32, 16, 8))]
    xr = [x.clone().requires_grad_() for x in xs]
    ws = [rnd(B, C, L, seed=20 + i) for i, L in enumerate((32, 16, 8))]
    rm, rv, loss = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64), 0
    for x, wgt in zip(xr, ws):
        y, mean, uvar = ref_block(x, w, gamma, beta, 1, dt)
        loss = loss + (y * wgt).sum()
        rm, rv = 0.9 * rm + 0.1 * (mean.detach() + cb), 0.9 * rv + 0.1 * uvar.detach()
    loss.backward()
    conv_h, bn_h = nn.Conv1d(C, C, 3, padding=1).to(DEV), nn.BatchNorm1d(C).to(DEV)
    with torch.no_grad():
        conv_h.weight.copy_(w.float()); conv_h.bias.copy_(cb.float())
    xh = [x.permute(0, 2, 1).contiguous().to(DEV, dt).requires_grad_() for x in xs]
    outs, _ = DF.conv_block(xh, conv_h, bn_h, True, dt)
    sum((o.float() * wgt.permute(0, 2, 1).float().to(DEV)).sum() for o, wgt in zip(outs, ws)).backward()
    tol = TOL[dt]
    for i in range(3):
        close(xh[i].grad.permute(0, 2, 1), xr[i].grad, tol * 3, "dx level %d" % i)
    close(conv_h.weight.grad, w.grad, tol * 3, "dW")
    close(bn_h.weight.grad, gamma.grad, tol * 3, "dgamma")
    assert conv_h.bias.grad is None or float(conv_h.bias.grad.abs().max()) == 0.0   # a conv bias in front of train-mode BN: zero gradient (None)
    close(bn_h.running_mean, rm, tol, "running_mean (the conv bias shifts it)")
    close(bn_h.running_var, rv, tol, "running_var")
    DF.flush_bn_counters()          # counter increments are batched; modules flush at the end of their forward
    assert int(bn_h.num_batches_tracked) == 3


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_input_stage_fwd_bwd(dt):
    from drn_amd import functional as DF
    B, T, D, P = 2, 16, 64, 256
    torch.manual_seed(0)
    fc, pt = nn.Linear(D, D).double(), nn.Linear(3, P).double()
    feats = torch.rand(B, T, D, generator=torch.Generator().manual_seed(1), dtype=torch.float64).float().double()
    if dt == torch.bfloat16:
        feats = feats.bfloat16().double()
        with torch.no_grad():
            fc.weight.copy_(fc.weight.bfloat16().double())
    pos = rnd(B, T, 3, seed=2).float().double()
    q = rnd(B, D, seed=3).float().double().requires_grad_()
    ref = torch.cat([fc(feats) * q[:, None, :], pt(pos)], dim=2)
    w = rnd(B, T, D + P, seed=4)
    (ref * w).sum().backward()
    fch, pth = nn.Linear(D, D).to(DEV), nn.Linear(3, P).to(DEV)
    with torch.no_grad():
        fch.weight.copy_(fc.weight.float()); fch.bias.copy_(fc.bias.float())
        pth.weight.copy_(pt.weight.float()); pth.bias.copy_(pt.bias.float())
    qh = q.detach().float().to(DEV).requires_grad_()
    prep = DF.input_prep(feats.float().to(DEV), pos.float().to(DEV), fch, dt)
    g0 = DF.input_stage(prep, fch, qh, pth)
    (g0.float() * w.float().to(DEV)).sum().backward()
    tol = TOL[dt]
    close(g0, ref, tol, "G0")
    close(qh.grad, q.grad, tol * 3, "dgate0")
    close(fch.weight.grad, fc.weight.grad, tol * 3, "dW prop_fc")
    close(fch.bias.grad, fc.bias.grad, tol * 3, "db prop_fc")
    close(pth.weight.grad, pt.weight.grad, tol * 3, "dW pos")
    close(pth.bias.grad, pt.bias.grad, tol * 3, "db pos")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,stride,T", [(3, 1, 32), (3, 2, 32), (1, 1, 24), (3, 1, 19)])
def test_input_stage_into_conv_with_tail(dt, k, stride, T):
    """input_stage -> conv block (model/backbone.py:28-33 at level 0) with the EmbedTail link: the conv's backward takes the
    position-embedding gradients through the conv and leaves those columns out of its input gradient; every gradient of both
    stages against fp64 autograd."""
    from drn_amd import functional as DF
    B, D, P, Cout = 3, 64, 32, 64
    torch.manual_seed(0)
    fc, pt = nn.Linear(D, D).double(), nn.Linear(3, P).double()
    feats = q(torch.rand(B, T, D, generator=torch.Generator().manual_seed(1), dtype=torch.float64).float().double(), dt)
    with torch.no_grad():
        fc.weight.copy_(q(fc.weight, dt))
    pos = rnd(B, T, 3, seed=2).float().double()
    gate = rnd(B, D, seed=3).float().double().requires_grad_()
    w = q(rnd(Cout, D + P, k, seed=11) / np.sqrt((D + P) * k), dt).requires_grad_()
    gamma = (rnd(Cout, seed=1).abs() + 0.5).requires_grad_()
    beta = (rnd(Cout, seed=2) * 0.3).requires_grad_()
    g0r = torch.cat([fc(feats) * gate[:, None, :], pt(pos)], dim=2)
    if dt == torch.bfloat16:
        g0r = g0r + (g0r.bfloat16().double() - g0r).detach()
    y, _, _ = ref_block(g0r.permute(0, 2, 1), w, gamma, beta, stride, dt)
    w1 = rnd(*y.shape, seed=6)
    (y * w1).sum().backward()
    fch, pth = nn.Linear(D, D).to(DEV), nn.Linear(3, P).to(DEV)
    conv_h = nn.Conv1d(D + P, Cout, k, stride=stride, padding=(k - 1) // 2, bias=False).to(DEV)
    bn_h = nn.BatchNorm1d(Cout).to(DEV)
    with torch.no_grad():
        fch.weight.copy_(fc.weight.float()); fch.bias.copy_(fc.bias.float())
        pth.weight.copy_(pt.weight.float()); pth.bias.copy_(pt.bias.float())
        conv_h.weight.copy_(w.float()); bn_h.weight.copy_(gamma.float()); bn_h.bias.copy_(beta.float())
    qh = gate.detach().float().to(DEV).requires_grad_()
    prep = DF.input_prep(feats.float().to(DEV), pos.float().to(DEV), fch, dt)
    g0, tail = DF.input_stage(prep, fch, qh, pth, with_tail=True)
    outs, _ = DF.conv_block([g0], conv_h, bn_h, True, dt, tail=tail)
    (outs[0].float() * w1.permute(0, 2, 1).float().to(DEV)).sum().backward()
    assert tail.dW is None and tail.db is None          # taken by the input stage's backward
    tol = TOL[dt]
    close(outs[0].permute(0, 2, 1), y, tol, "conv out")
    close(pth.weight.grad, pt.weight.grad, tol * 3, "dW pos (through the conv)")
    # (a 1-tap conv in front of a train-mode BN cancels a constant input shift: the true bias gradient is ~0 there, a sum of
    # terms of the weight gradient's magnitude -- that magnitude is the scale)
    wscale = float(pt.weight.grad.abs().max())
    close(pth.bias.grad, pt.bias.grad, tol * 3, "db pos (through the conv)", scale=wscale)
    close(qh.grad, gate.grad, tol * 3, "dgate0")
    close(fch.weight.grad, fc.weight.grad, tol * 3, "dW prop_fc")
    close(fch.bias.grad, fc.bias.grad, tol * 3, "db prop_fc")
    close(conv_h.weight.grad, w.grad, tol * 3, "dW conv")
    # and the same numbers as the path without the link (full input gradient + pos_embed_bwd) to rounding
    for m in (fch, pth, conv_h, bn_h):
        m.zero_grad(set_to_none=True)
    qh2 = gate.detach().float().to(DEV).requires_grad_()
    g0b = DF.input_stage(prep, fch, qh2, pth)
    outs2, _ = DF.conv_block([g0b], conv_h, bn_h, True, dt)
    (outs2[0].float() * w1.permute(0, 2, 1).float().to(DEV)).sum().backward()
    close(pth.weight.grad, pt.weight.grad, tol * 3, "dW pos (plain path)")
    close(pth.bias.grad, pt.bias.grad, tol * 3, "db pos (plain path)", scale=wscale)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,Ls", [(2, 64, (32, 16, 8)), (3, 512, (64, 32, 16)), (3, 128, (50, 25, 13))])
def test_head_out_fwd_bwd(dt, B, C, Ls):
    """cls_logits + exp(scale*bbox_pred) on the two halves of the stacked tower output, 3 levels (model/fcos.py:96-100); the
    second shape has the model's channel count (runs of 8 rows inside one level: the sliding-window paths), the third has
    sequence lengths that are no multiples of 8 (runs across clip and level boundaries: masks and the row-by-row paths)."""
    from drn_amd import functional as DF
    torch.manual_seed(0)
    cls, box = nn.Conv1d(C, 1, 3, padding=1).double(), nn.Conv1d(C, 2, 3, padding=1).double()
    scales = (rnd(3, seed=1) * 0.1 + 1.0).requires_grad_()
    xs = [rnd(B, 2 * C, L, seed=2 + i) * (8.0 / np.sqrt(C)) for i, L in enumerate(Ls)]
    if dt == torch.bfloat16:
        xs = [x.bfloat16().double() for x in xs]
    xr = [x.clone().requires_grad_() for x in xs]
    lo = [cls(x[:, :C]) for x in xr]
    rg = [torch.exp(scales[l] * box(x[:, C:])) for l, x in enumerate(xr)]
    flat = lambda ts: torch.cat([t.permute(0, 2, 1).reshape(-1, t.size(1)) for t in ts])
    w1, w2 = rnd(B * sum(Ls), 1, seed=8), rnd(B * sum(Ls), 2, seed=9)
    ((flat(lo) * w1).sum() + (flat(rg) * w2).sum()).backward()
    ch, bh = nn.Conv1d(C, 1, 3, padding=1).to(DEV), nn.Conv1d(C, 2, 3, padding=1).to(DEV)
    with torch.no_grad():
        ch.weight.copy_(cls.weight.float()); ch.bias.copy_(cls.bias.float())
        bh.weight.copy_(box.weight.float()); bh.bias.copy_(box.bias.float())
    sh = scales.detach().float().to(DEV).requires_grad_()
    xh = [x.permute(0, 2, 1).contiguous().to(DEV, dt).requires_grad_() for x in xs]
    logits, reg = DF.head_out(xh, [(ch, None), (bh, sh)], cols=[0, C], dtype=dt)
    ((logits * w1.float().to(DEV)).sum() + (reg * w2.float().to(DEV)).sum()).backward()
    tol = TOL[dt]
    close(logits, flat(lo), tol, "logits")
    close(reg, flat(rg), tol, "reg")
    for i in range(3):
        close(xh[i].grad.permute(0, 2, 1), xr[i].grad, tol * 3, "dx level %d" % i)
    close(ch.weight.grad, cls.weight.grad, tol * 3, "dW cls")
    close(bh.weight.grad, box.weight.grad, tol * 3, "dW box")
    close(ch.bias.grad, cls.bias.grad, tol * 3, "db cls")
    close(bh.bias.grad, box.bias.grad, tol * 3, "db box")
    close(sh.grad, scales.grad, tol * 3, "dscale")


@pytest.mark.parametrize("use_bn,use_relu", [(True, False), (False, True), (False, False)])
def test_conv_factory_variants_match_torch(use_bn, use_relu):
    """The factory combinations DRN never instantiates (model/basic_blocks.py:5-33: no BatchNorm -> biased conv, no ReLU) against
    the same nn.Sequential in fp64: outputs, input / weight / bias gradients, state_dict keys."""
    from drn_amd.model.basic_blocks import conv_with_kaiming_uniform
    torch.manual_seed(3)
    B, L, Cin, Cout, k, stride = 3, 40, 64, 72, 3, 2
    blk = conv_with_kaiming_uniform(use_bn, use_relu)(Cin, Cout, k, stride).to(DEV).train()
    mods = [torch.nn.Conv1d(Cin, Cout, k, stride=stride, padding=1, bias=not use_bn)]
    if use_bn:
        mods.append(torch.nn.BatchNorm1d(Cout))
    if use_relu:
        mods.append(torch.nn.ReLU())
    ref = (torch.nn.Sequential(*mods) if len(mods) > 1 else mods[0]).double().train()
    assert list(ref.state_dict().keys()) == list(blk.state_dict().keys())
    ref.load_state_dict({k_: v.double().cpu() for k_, v in blk.state_dict().items()})
    x = torch.randn(B, Cin, L)
    xd = x.to(DEV).requires_grad_()
    xr = x.double().requires_grad_()
    y, yr = blk(xd), ref(xr)
    g = torch.randn(yr.shape)
    y.backward(g.to(DEV))
    yr.backward(g.double())
    scale = float(yr.abs().max())
    assert float((y.cpu().double() - yr).abs().max()) <= 3e-5 * scale
    assert float((xd.grad.cpu().double() - xr.grad).abs().max()) <= 3e-5 * float(xr.grad.abs().max())
    for (n1, p1), (n2, p2) in zip(blk.named_parameters(), ref.named_parameters()):
        assert n1 == n2 and float((p1.grad.cpu().double() - p2.grad).abs().max()) <= 5e-5 * max(float(p2.grad.abs().max()), 1e-3), n1


@pytest.mark.parametrize("top", ["maxpool", "p6p7", "p6p7_c5"])
def test_fpn_top_blocks(top):
    """model/FPN.py:72-103: the extra pyramid levels of the optional top blocks (unused by DRN's own config)."""
    from drn_amd.model.FPN import FPN, LastLevelMaxPool, LastLevelP6P7
    from drn_amd.model.basic_blocks import conv_with_kaiming_uniform
    torch.manual_seed(4)
    B, C = 2, 64
    chans, Ls = [32, 48, 96], [32, 16, 8]
    tb = LastLevelMaxPool() if top == "maxpool" else LastLevelP6P7(C if top == "p6p7" else chans[-1], C)
    fpn = FPN(chans, C, conv_with_kaiming_uniform(True, True), top_blocks=tb).to(DEV).train()
    xs = [torch.randn(B, c, L) for c, L in zip(chans, Ls)]
    outs = fpn([x.to(DEV) for x in xs])
    assert len(outs) == (4 if top == "maxpool" else 5)
    p3 = outs[2].detach().cpu().double()
    if top == "maxpool":
        want = torch.nn.functional.max_pool2d(p3, 1, 2, 0)
        assert outs[3].shape == want.shape and float((outs[3].cpu().double() - want).abs().max()) == 0.0
    else:
        src = p3 if top == "p6p7" else xs[-1].double()
        p6 = torch.nn.functional.conv1d(src, tb.p6.weight.detach().cpu().double(), tb.p6.bias.detach().cpu().double(), 2, 1)
        p7 = torch.nn.functional.conv1d(torch.relu(p6), tb.p7.weight.detach().cpu().double(), tb.p7.bias.detach().cpu().double(), 2, 1)
        for got, want in ((outs[3], p6), (outs[4], p7)):
            assert got.shape == want.shape and float((got.cpu().double() - want).abs().max()) <= 5e-5 * float(want.abs().max())
        sum(o.float().sum() for o in outs).backward()
        assert tb.p6.weight.grad is not None and tb.p7.bias.grad is not None
