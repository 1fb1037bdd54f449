"""GPU parity of the fused FCOS loss kernels (drn_fcos_loss_fwd/bwd) against the CPU oracle's autograd."""
import numpy as np
import pytest
import torch

from oracle import drn_oracle as O

pytestmark = pytest.mark.gpu


def make_case(B, T, seed, matched, loc0=False):
    g = torch.Generator().manual_seed(seed)
    Ls = [T, T // 2, T // 4]
    strides = [1, 2, 4]
    logits = [torch.randn(B, 1, L, generator=g) for L in Ls]
    reg = [torch.exp(torch.randn(B, 2, L, generator=g)) * (1.5 + l) for l, L in enumerate(Ls)]
    iou = [torch.randn(B, 1, L, generator=g) for L in Ls]
    gt = torch.stack([torch.rand(B, generator=g) * 0.4, 0.5 + torch.rand(B, generator=g) * 0.4], 1)
    if matched:     # GT equal to one prediction per clip -> tIoU>0.9 positives with exact ties
        rows = []
        for b in range(B):
            # loc0: location index 0 of level 0, the one model/loss.py:180-181 clamps; its raw start (0.5 - reg) / 32 is made
            # negative for every clip, so the clamp decides the target of a tIoU > 0.9 positive and zeroes d tIoU / d start there
            t = 0 if loc0 and b % 2 == 0 else (3 + 5 * b) % T
            if t == 0:
                reg[0][b, 0, 0] = 0.9 + 0.3 * b
            loc = t + 0.5
            s, e = max((loc - reg[0][b, 0, t].item()) / 32.0, 0.0), min((loc + reg[0][b, 1, t].item()) / 32.0, 1.0)
            if loc0:        # shrink the GT by 1.5 % per side: no exact min / max ties, so d tIoU / d pred is O(1 / width) -- with an
                s, e = s + 0.015 * (e - s), e - 0.015 * (e - s)     # exact match it vanishes and the clamp's gradient mask would go unseen
            rows.append([s, e])
        gt = torch.tensor(rows, dtype=torch.float64).float()
    locs = [O.FCOSModule.locations_for(L, s, "cpu") for L, s in zip(Ls, strides)]
    return Ls, strides, logits, reg, iou, gt, locs


@pytest.mark.parametrize("B,T,stage,matched,seed", [(2, 32, 1, False, 0), (3, 32, 3, True, 1), (4, 64, 3, True, 2),
                                                    (2, 32, 3, False, 3), (5, 256, 3, True, 4),
                                                    (4, 32, 3, "loc0", 5), (3, 64, 2, "loc0", 6), (6, 256, 3, "loc0", 7)])
def test_loss_fwd_bwd(B, T, stage, matched, seed):
    from drn_amd import functional as DF
    Ls, strides, logits, reg, iou, gt, locs = make_case(B, T, seed, bool(matched), loc0=matched == "loc0")
    if matched == "loc0":
        # the case is only worth its name if the clamp changes a positive's prediction: raw start < 0 at location 0, tIoU > 0.9 there
        for b in range(0, B, 2):
            raw = (0.5 - reg[0][b, 0, 0].item()) / 32.0
            assert raw < 0.0
            pred = torch.tensor([[0.0, min((0.5 + reg[0][b, 1, 0].item()) / 32.0, 1.0)]])
            assert float(O.segment_tiou(pred, gt[b:b + 1])) > 0.9
            unclamped = torch.tensor([[raw, pred[0, 1].item()]])
            assert float(O.segment_tiou(unclamped, gt[b:b + 1])) < float(O.segment_tiou(pred, gt[b:b + 1])) - 1e-3
    cfg = {"fcos_loss_gamma": 2.0, "fcos_loss_alpha": 0.25}
    lr = [x.clone().requires_grad_() for x in logits]
    rr = [x.clone().requires_grad_() for x in reg]
    ir = [x.clone().requires_grad_() for x in iou]
    lc, lg, li = O.FCOSLoss(cfg)(locs, lr, rr, gt, ir, stage == 1)
    w = torch.tensor([0.7, 1.3, 2.1])
    tot = w[0] * lc + w[1] * lg
    if li.requires_grad:
        tot = tot + w[2] * li.reshape(())
    tot.backward()
    flat = lambda ts: torch.cat([t.permute(0, 2, 1).reshape(-1, t.size(1)) for t in ts])
    dev = torch.device("cuda:0")
    L_, R_, I_ = (flat(x).to(dev).requires_grad_() for x in (logits, reg, iou))
    levels = [(Ls[i], float(strides[i]), float(O.SIZES_OF_INTEREST[i][0]), float(O.SIZES_OF_INTEREST[i][1])) for i in range(3)]
    l_cls, l_reg, l_iou, counts, all3 = DF.fcos_loss(L_, R_, I_, gt.to(dev), levels, B, 2.0, 0.25, 32.0, stage != 1)
    assert l_cls.shape == l_reg.shape == l_iou.shape == (1,)
    losses = torch.cat([l_cls, l_reg, l_iou])
    (losses * w.to(dev)).sum().backward()
    got = losses.detach().cpu().numpy()
    np.testing.assert_allclose(got[0], lc.item(), atol=1e-5)
    np.testing.assert_allclose(got[1], lg.item(), atol=1e-5)
    if stage != 1:
        np.testing.assert_allclose(got[2], float(li.reshape(-1)[0]), atol=1e-5)
        if matched:
            assert counts[1].item() >= B
    gl = flat([x.grad for x in lr])
    gr = flat([x.grad if x.grad is not None else torch.zeros_like(x) for x in rr])
    np.testing.assert_allclose(L_.grad.cpu().numpy(), gl.numpy(), atol=1e-6)
    np.testing.assert_allclose(R_.grad.cpu().numpy(), gr.numpy(), atol=2e-6, rtol=1e-4)
    if stage != 1 and ir[0].grad is not None:
        gi = flat([x.grad for x in ir])
        np.testing.assert_allclose(I_.grad.cpu().numpy(), gi.numpy(), atol=1e-6)


@pytest.mark.gpu
def test_loss_total_matches_python_sum():
    """DF.loss_total (one reduction over the (3,) view) vs sum(loss_dict.values()) of the reference loop (main.py:225):
    same value, same input gradients; a plain dict falls back to the python sum."""
    from drn_amd import functional as DF
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, Ls, strides = 3, (16, 8, 4), (2, 4, 8)
    R = B * sum(Ls)
    gt = torch.tensor([[3.0, 20.0], [0.0, 9.0], [11.0, 30.0]])
    levels = [(Ls[i], float(strides[i]), float(O.SIZES_OF_INTEREST[i][0]), float(O.SIZES_OF_INTEREST[i][1])) for i in range(3)]
    base = [torch.randn(R, 1, generator=g), torch.randn(R, 2, generator=g).abs() + 0.1, torch.randn(R, 1, generator=g)]
    grads = []
    for fused in (True, False):
        L_, R_, I_ = (x.clone().to(dev).requires_grad_() for x in base)
        l_cls, l_reg, l_iou, _, all3 = DF.fcos_loss(L_, R_, I_, gt.to(dev), levels, B, 2.0, 0.25, 32.0, True)
        d = DF.LossDict(loss_cls=l_cls, loss_reg=l_reg, loss_iou=l_iou)
        if fused:
            d.total = all3
            total = DF.loss_total(d)
            assert total.data_ptr() == all3.data_ptr(), "the fused branch must hand out the kernel's own total"
        else:
            total = DF.loss_total(dict(d))
        (2.5 * total).sum().backward()
        grads.append((float(total.reshape(-1)[0]), L_.grad.clone(), R_.grad.clone(), I_.grad.clone()))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * max(1.0, abs(grads[1][0]))
    for a, b in zip(grads[0][1:], grads[1][1:]):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)
