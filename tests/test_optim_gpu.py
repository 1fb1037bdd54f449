"""Fused clip+Adam HIP kernels vs torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (main.py:140,238-243)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_norm", [0.5, 1e9])
def test_fused_adam_matches_torch(max_norm):
    from drn_amd.dist import GradReducer
    from drn_amd.optim import FusedAdam
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    shapes = [(300, 64, 3), (64,), (1,), (4096, 33), (7,), (50000,), (2, 3, 5)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    red = GradReducer(pa, world_size=1, bucket_bytes=200 << 10)      # several buckets
    assert len(red.buckets) >= 2
    fused = FusedAdam(red, lr=1e-2, max_norm=max_norm)
    ref = torch.optim.Adam(pb, lr=1e-2)
    for it in range(4):
        grads = [torch.randn(s, generator=g).to(dev) * (0.1 + it) for s in shapes]
        red.zero()
        for p, q, gr in zip(pa, pb, grads):
            p.grad = gr.clone()          # finish() moves hand-set gradients into the flat bucket
            q.grad = gr.clone()
        red.finish()
        fused.step()
        tn = torch.nn.utils.clip_grad_norm_(pb, max_norm)
        ref.step()
        assert abs(float(fused.total_norm()) - float(tn)) <= 1e-4 * float(tn)
        for p, q in zip(pa, pb):
            assert torch.allclose(p, q, atol=2e-6, rtol=1e-5), float((p - q).abs().max())
