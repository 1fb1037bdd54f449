"""GPU parity of the stand-alone layers the reference exports beside the fused path: IOULoss with its `weight` branch
(model/layers/iou_loss.py:5-24, drn_iou_loss_fwd / drn_iou_loss_bwd) and LGP(use_bn=False) (model/LGP.py:5-27), against the
goldens recorded from the reference (tests/golden/gen_golden.py layers) and against the oracle on larger inputs."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import load_golden
from drn_amd.utils.synthetic import seeded_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["none", "pos", "zero"])
def test_iou_loss_module_matches_reference_golden(tag):
    from drn_amd.model.layers import IOULoss
    g = load_golden("layers")
    dev = torch.device("cuda:0")
    p = torch.from_numpy(g["iou/pred"]).to(dev).requires_grad_()
    t = torch.from_numpy(g["iou/target"]).to(dev).requires_grad_()
    w = {"none": None, "pos": torch.from_numpy(g["iou/weight_pos"]).to(dev), "zero": torch.zeros(p.shape[0], device=dev)}[tag]
    loss = IOULoss()(p, t, w)
    (loss * 1.7).backward()
    np.testing.assert_allclose(loss.detach().cpu().numpy().reshape(1), g["iou/%s/loss" % tag], rtol=2e-6)
    np.testing.assert_allclose(p.grad.cpu().numpy(), g["iou/%s/dpred" % tag], rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(t.grad.cpu().numpy(), g["iou/%s/dtarget" % tag], rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("N,weighted,seed", [(1, False, 0), (1, True, 1), (1000, True, 2), (14336, False, 3), (100001, True, 4)])
def test_iou_loss_abi_matches_oracle(N, weighted, seed):
    """Through the C-ABI, against the oracle in float64, ties in min() included (every 7th row: pred == target on one side)."""
    from drn_amd import _lib
    from oracle import drn_oracle as O
    gen = torch.Generator().manual_seed(seed)
    pred = torch.rand(N, 2, generator=gen) * 8 + 0.1
    target = torch.rand(N, 2, generator=gen) * 8 + 0.1
    target[::7, 0] = pred[::7, 0]
    weight = torch.rand(N, generator=gen) if weighted else None
    p64, t64 = pred.double().requires_grad_(), target.double().requires_grad_()
    ref = O.iou_loss(p64, t64, weight.double() if weighted else None)
    (ref * 0.5).backward()
    dev = torch.device("cuda:0")
    pg, tg = pred.to(dev), target.to(dev)
    wg = weight.to(dev) if weighted else None
    out2 = torch.empty(2, device=dev)
    gout = torch.tensor([0.5], device=dev)
    dp, dt = torch.empty_like(pg), torch.empty_like(tg)
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
    _lib.check(L.drn_iou_loss_fwd(vp(pg), vp(tg), vp(wg), ctypes.c_int64(N), vp(out2), st), "fwd")
    _lib.check(L.drn_iou_loss_bwd(vp(pg), vp(tg), vp(wg), ctypes.c_int64(N), vp(out2), vp(gout), vp(dp), vp(dt), st), "bwd")
    torch.cuda.synchronize()
    assert abs(float(out2[0]) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert (float(out2[1]) > 0) == weighted
    scale = float(p64.grad.abs().max())
    np.testing.assert_allclose(dp.cpu().numpy(), p64.grad.float().numpy(), rtol=1e-4, atol=1e-6 * scale)
    np.testing.assert_allclose(dt.cpu().numpy(), t64.grad.float().numpy(), rtol=1e-4, atol=1e-6 * scale)


def test_iou_loss_empty_input_raises_like_the_reference():
    from drn_amd._lib import DrnError
    from drn_amd.model.layers import IOULoss
    e = torch.empty(0, 2, device="cuda:0")
    with pytest.raises(DrnError):
        IOULoss()(e, e)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_lgp_without_bn_flag_matches_reference_golden(mode):
    from drn_amd.model.LGP import LGP
    g = load_golden("layers")
    pre = "lgp_nobn/%s/" % mode
    net = LGP(input_dim=64, query_dim=64, use_bn=False)
    assert net.query_fc[0].bias is not None and isinstance(net.query_fc[1], torch.nn.BatchNorm1d)
    net.load_state_dict(seeded_state_dict(net, seed=4))
    with torch.no_grad():
        net.query_fc[0].bias.copy_(torch.from_numpy(g[pre + "bias"]))
        net.query_fc[1].running_mean.copy_(torch.from_numpy(g[pre + "rm0"]))
        net.query_fc[1].running_var.copy_(torch.from_numpy(g[pre + "rv0"]))
    net = net.to("cuda:0").train(mode == "train")
    x = torch.from_numpy(g[pre + "x"]).cuda().requires_grad_()
    q = torch.from_numpy(g[pre + "q"]).cuda().requires_grad_()
    y = net(x, q)

    def close(got, ref, tol=2e-5):
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref, atol=tol * max(1.0, float(np.abs(ref).max())), rtol=0)
    close(y, g[pre + "y"])
    np.testing.assert_allclose(net.query_fc[1].running_mean.cpu().numpy(), g[pre + "rm"], atol=1e-6)
    np.testing.assert_allclose(net.query_fc[1].running_var.cpu().numpy(), g[pre + "rv"], atol=1e-6)
    if mode == "train":               # (backward through an eval-mode BatchNorm is not provided by the HIP path)
        (y * torch.from_numpy(g[pre + "w"]).cuda()).sum().backward()
        close(x.grad, g[pre + "dx"])
        close(q.grad, g[pre + "dq"], 1e-4)
        close(net.query_fc[0].weight.grad, g[pre + "dw"], 1e-4)
        close(net.query_fc[1].weight.grad, g[pre + "dgamma"], 1e-4)
        close(net.query_fc[1].bias.grad, g[pre + "dbeta"], 1e-4)
        assert float(net.query_fc[0].bias.grad.abs().max()) == 0.0 and float(np.abs(g[pre + "dbias"]).max()) < 1e-5
