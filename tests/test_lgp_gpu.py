"""GPU parity of language-guided pooling (drn_amd.model.LGP) against the golden recorded from the reference's
model/LGP.py and against the CPU oracle on a larger shape."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from drn_amd.utils.synthetic import seeded_state_dict

pytestmark = pytest.mark.gpu


def test_lgp_matches_reference_golden():
    from drn_amd.model.LGP import LGP
    g = load_golden("lgp")
    net = LGP(input_dim=64, query_dim=64)
    net.load_state_dict(seeded_state_dict(net, seed=3))
    net = net.to("cuda:0").train()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_()
    q = torch.from_numpy(g["q"]).cuda().requires_grad_()
    y = net(x, q)
    (y * torch.from_numpy(g["w"]).cuda()).sum().backward()
    def close(got, ref, tol=2e-5):          # 2e-5 of the tensor's scale (fp32 noise floor; never pure relative)
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref, atol=tol * max(1.0, float(np.abs(ref).max())), rtol=0)
    close(y, g["y"])
    close(x.grad, g["dx"])
    close(q.grad, g["dq"], 1e-4)
    close(net.query_fc[0].weight.grad, g["dw"], 1e-4)
    close(net.query_fc[1].weight.grad, g["dgamma"], 1e-4)
    close(net.query_fc[1].bias.grad, g["dbeta"], 1e-4)
    np.testing.assert_allclose(net.query_fc[1].running_mean.cpu().numpy(), g["rm"], atol=1e-6)
    np.testing.assert_allclose(net.query_fc[1].running_var.cpu().numpy(), g["rv"], atol=1e-6)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 3e-5), (torch.bfloat16, 3e-2)])
def test_lgp_matches_oracle_1024(dt, tol):
    from drn_amd.model.LGP import LGP
    from oracle import drn_oracle as O
    B, C, t = 8, 1024, 32
    ref = O.LGP(input_dim=C, query_dim=C)
    ref.load_state_dict(seeded_state_dict(ref, seed=5))
    ref = ref.double().train()
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, t, generator=gen).to(dt).double()
    q = torch.randn(B, C, generator=gen).double()
    xr, qr = x.clone().requires_grad_(), q.clone().requires_grad_()
    y = ref(xr, qr)
    w = torch.randn(y.shape, generator=gen).double()
    (y * w).sum().backward()
    net = LGP(input_dim=C, query_dim=C)
    net.load_state_dict(seeded_state_dict(net, seed=5))
    net = net.cuda().train()
    net.compute_dtype = dt
    xh = x.float().cuda().requires_grad_()
    qh = q.float().cuda().requires_grad_()
    yh = net(xh, qh)
    (yh.float() * w.float().cuda()).sum().backward()

    def close(a, b, what, f=1.0):
        err = float((a.detach().double().cpu() - b.detach()).abs().max())
        assert err <= tol * f * max(float(b.abs().max()), 1e-3), (what, err)
    close(yh, y, "out")
    close(xh.grad, xr.grad, "dx")
    close(qh.grad, qr.grad, "dq", 3)
    close(net.query_fc[0].weight.grad, ref.query_fc[0].weight.grad, "dW", 3)
    close(net.query_fc[1].weight.grad, ref.query_fc[1].weight.grad, "dgamma", 3)
