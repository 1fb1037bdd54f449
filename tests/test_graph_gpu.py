"""A hipGraph-replayed training step must produce the same parameters as the eager step (same kernels, same order)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def make(seed_model=0):
    from drn_amd.dist import GradReducer
    from drn_amd.model import mainModel
    from drn_amd.optim import FusedAdam
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    dev = "cuda:0"
    m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 3)), compute_dtype=torch.bfloat16)
    m.load_state_dict(seeded_state_dict(m, seed_model))
    m = m.to(dev).train()
    params = [p for p in m.parameters() if p.requires_grad]
    red = GradReducer(params, world_size=1)
    opt = FusedAdam(red, lr=1e-3, max_norm=0.5)
    batch = [b.to(dev) for b in synthetic_batch(4, 32, 64, seed=1)]

    def step():
        red.zero()
        _, losses = m(*batch)
        sum(losses.values()).backward()
        red.finish()
        opt.step()
        return losses

    return m, step


def test_graph_replay_matches_eager():
    from drn_amd.graph import GraphedStep
    n = 6
    m1, step1 = make()
    eager_losses = []
    for _ in range(n):
        eager_losses.append(float(step1()["loss_cls"]))
    m2, step2 = make()
    g = GraphedStep(step2, warmup=2).capture()          # 2 eager warm-up steps ran; capture itself executes nothing
    graph_losses = []
    for _ in range(n - 2):
        graph_losses.append(float(g()["loss_cls"]))
    torch.cuda.synchronize()
    assert eager_losses[0] != eager_losses[-1], "training made no progress: %s" % eager_losses
    # replay k is training step 2+k
    for k, v in enumerate(graph_losses):
        assert abs(v - eager_losses[2 + k]) <= 2e-3 * max(1.0, abs(v)), (k, graph_losses, eager_losses)
    for (k1, p1), (k2, p2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        if p1.is_floating_point():
            assert torch.allclose(p1, p2, atol=2e-3, rtol=2e-3), (k1, float((p1 - p2).abs().max()))
