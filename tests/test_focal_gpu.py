"""drn_focal_fwd / drn_focal_bwd -- the C-ABI twin of the reference's only FFI (fcos_core._C.sigmoid_focalloss_forward /
_backward, model/layers/sigmoid_focal_loss.py:18-33) -- through ctypes against the oracle's class-general formula
(oracle.sigmoid_focal_loss_sum = sigmoid_focal_loss.py:40-52), element-wise, any number of classes."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import drn_oracle as O

pytestmark = pytest.mark.gpu


def oracle_elementwise(logits, targets, gamma, alpha):
    """Per-element losses and their gradient under d_losses from the oracle's formula (float64 for the gradient check)."""
    x = logits.double().requires_grad_()
    C = x.shape[1]
    cls = torch.arange(1, C + 1, dtype=targets.dtype)[None]
    t = targets[:, None]
    p = torch.sigmoid(x)
    pos = (t == cls).double()
    neg = ((t != cls) & (t >= 0)).double()
    loss = -pos * alpha * (1 - p) ** gamma * torch.log(p) - neg * (1 - alpha) * p ** gamma * torch.log(1 - p)
    return x, loss


@pytest.mark.parametrize("N,C,seed", [(1, 1, 0), (14336, 1, 1), (777, 2, 2), (1000, 20, 3), (4097, 80, 4)])
def test_focal_matches_oracle(N, C, seed):
    from drn_amd import _lib
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(N, C, generator=g) * 4.0
    targets = torch.randint(-1, C + 1, (N,), generator=g, dtype=torch.int32)      # -1 = ignored, 0 = background
    d_losses = torch.randn(N, C, generator=g)
    gamma, alpha = 2.0, 0.25
    # the oracle's summed focal loss is the sum of these elements (pins this helper to oracle/drn_oracle.py)
    x, ref = oracle_elementwise(logits, targets, gamma, alpha)
    assert abs(ref.sum().item() - O.sigmoid_focal_loss_sum(logits.double(), targets, gamma, alpha).item()) <= 1e-9 * max(1.0, ref.abs().sum().item())
    (ref * d_losses.double()).sum().backward()
    dev = torch.device("cuda:0")
    lg, tg, dl = logits.to(dev), targets.to(dev), d_losses.to(dev)
    out, dx = torch.empty_like(lg), torch.empty_like(lg)
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(L.drn_focal_fwd(vp(lg), vp(tg), ctypes.c_int64(N), C, ctypes.c_float(gamma), ctypes.c_float(alpha), vp(out), st), "fwd")
    _lib.check(L.drn_focal_bwd(vp(lg), vp(tg), vp(dl), ctypes.c_int64(N), C, ctypes.c_float(gamma), ctypes.c_float(alpha), vp(dx), st), "bwd")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().float().numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(dx.cpu().numpy(), x.grad.float().numpy(), rtol=1e-5, atol=1e-7)


def test_focal_large_logits_stay_finite():
    """|logit| > 88: the in-repo CPU formula gives inf/nan (log(0)); the stable form (what the CUDA kernel of fcos_core does)
    stays finite and equals the analytic limit: loss -> |x| * weight on the wrong side, 0 on the right side."""
    from drn_amd.model.layers import SigmoidFocalLoss, sigmoid_focal_loss
    dev = torch.device("cuda:0")
    x = torch.tensor([[-120.0], [120.0], [-95.0], [95.0], [0.0]], device=dev, requires_grad=True)
    t = torch.tensor([1, 0, 0, 1, 1], dtype=torch.int32, device=dev)
    per = sigmoid_focal_loss(x, t, 2.0, 0.25)
    want = torch.tensor([[0.25 * 120.0], [0.75 * 120.0], [0.0], [0.0], [0.25 * 0.25 * float(np.log(2.0))]])
    np.testing.assert_allclose(per.detach().cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-30)
    tot = SigmoidFocalLoss(2.0, 0.25)(x, t)
    tot.backward()
    assert torch.isfinite(x.grad).all()
    np.testing.assert_allclose(x.grad[:2, 0].cpu().numpy(), [-0.25, 0.75], rtol=1e-6)


def test_focal_module_matches_fused_loss_kernel():
    """The stand-alone layer and the fused whole-loss kernel (drn_fcos_loss_fwd) agree on loss_cls for C = 1."""
    from drn_amd import functional as DF
    from drn_amd.model.layers import SigmoidFocalLoss
    dev = torch.device("cuda:0")
    B, T = 3, 32
    g = torch.Generator().manual_seed(5)
    Ls = [T, T // 2, T // 4]
    R = B * sum(Ls)
    logits = torch.randn(R, 1, generator=g).to(dev)
    reg = torch.exp(torch.randn(R, 2, generator=g)).to(dev)
    gt = torch.tensor([[0.1, 0.6], [0.3, 0.9], [0.0, 0.4]], device=dev)
    levels = [(Ls[i], float(2 ** i), float(O.SIZES_OF_INTEREST[i][0]), float(O.SIZES_OF_INTEREST[i][1])) for i in range(3)]
    from drn_amd import ops
    out5 = torch.empty(6, device=dev)
    labels = torch.empty(R, device=dev)
    ops.fcos_loss_fwd(ops.loss_levels(levels), B, logits, reg, None, gt, 2.0, 0.25, 32.0, 0, out5, labels=labels)
    n_pos = int(out5[3].item())
    assert n_pos == int(labels.sum().item()) and n_pos > 0
    focal = SigmoidFocalLoss(2.0, 0.25)(logits, labels.to(torch.int32))
    np.testing.assert_allclose(out5[0].item(), focal.item() / (n_pos + B), rtol=2e-6)
