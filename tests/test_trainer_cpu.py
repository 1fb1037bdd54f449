"""Trainer counterpart (drn_amd/trainer.py): stage plan, checkpoint format and key-filtered resume on the CPU, and the
torch-optimizer path of Trainer.train_step driven by the ORACLE model against the loss trajectories recorded from the
reference model under main.py's loop (tests/golden/traj_s{1,2,3}.npz, generator: gen_golden.py trajectory)."""
import os

import numpy as np
import pytest
import torch

from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
from drn_amd import trainer as T

HERE = os.path.dirname(os.path.abspath(__file__))


def traj(stage):
    """stage: 1, 2, 3 or "1_lowlr" (stage 1 at lr = 1e-5, where the trajectory is not chaotic)."""
    return np.load(os.path.join(HERE, "golden", "traj_s%s.npz" % stage))


def traj_batches(g):
    out = []
    for i, seed in enumerate((1, 2)):
        b = list(synthetic_batch(int(g["B"]), int(g["T"]), int(g["D"]), seed=seed))
        b[4] = torch.from_numpy(g["gt%d" % i])
        out.append(b)
    return out


# Adam's first update is lr * g / (|g| + 1e-8): every parameter moves by +-lr whatever the size of its gradient, including
# the ones whose gradient is analytically zero (a conv bias in front of a train-mode BatchNorm) and therefore pure rounding
# noise -- so two correct implementations drift apart geometrically at stage 1's lr = 1e-3.  The gate widens per step.
STEP_RTOL = [2e-5, 2e-4, 2e-3, 2e-2]


def check_trajectory(got, g, state_dict, scale=1.0):
    want = g["losses"]
    for it in range(len(want)):
        np.testing.assert_allclose(got[it], want[it], rtol=STEP_RTOL[it] * scale, atol=2e-5 * scale, err_msg="step %d" % it)
    for key in g.files:
        if key.startswith("cs/"):
            v = state_dict[key[3:]].double()
            assert abs(v.abs().sum().item() - g[key][1]) <= 2e-3 * scale * max(g[key][1], 1e-3), key


def oracle_model(stage, D=64):
    from oracle import drn_oracle as O
    m = O.mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", D, stage)))
    m.load_state_dict(seeded_state_dict(m, 0))
    return m


@pytest.mark.parametrize("stage", [1, 2, 3])
def test_stage_plan(stage):
    m = oracle_model(stage)
    params, lr, epochs, which = T.stage_plan(m, stage, 1e-3)
    names = {id(p): n for n, p in m.named_parameters()}
    learned = sorted(names[id(p)] for p in params)
    if stage == 1:
        assert lr == 1e-3 and epochs == 10 and which == "sum"
        assert not any("iou_scores" in n or "mix_fc" in n for n in learned)
        assert all(not p.requires_grad for n, p in m.named_parameters() if "iou_scores" in n or "mix_fc" in n)
        assert len(learned) == sum(1 for n, _ in m.named_parameters() if "iou_scores" not in n and "mix_fc" not in n)
    elif stage == 2:
        assert lr == pytest.approx(1e-5) and which == "loss_iou"
        assert all(n.startswith(("fcos.head.iou_scores", "fcos.head.mix_fc")) for n in learned) and len(learned) == 10
        assert all(p.requires_grad for p in m.parameters())              # the trunk still receives gradients (clip quirk)
    else:
        assert lr == pytest.approx(1e-7) and which == "sum" and len(learned) == len(list(m.parameters()))


@pytest.mark.parametrize("stage", [1, 2, 3, "1_lowlr"])
def test_torch_path_on_oracle_follows_reference_trajectory(stage):
    g = traj(stage)
    stage = int(g["stage"])
    m = oracle_model(stage)
    tr = T.Trainer(m, stage, lr=float(g["lr"]), clip_gradient=0.5, fused=False)
    batches = traj_batches(g)
    got = []
    for it in range(int(g["steps"])):
        ld = tr.train_step(batches[it % 2])
        got.append([float(ld[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    check_trajectory(np.array(got), g, m.state_dict())


def test_checkpoint_format_and_filtered_resume(tmp_path):
    m = oracle_model(1)
    state = {"epoch": 3, "state_dict": T.checkpoint_state_dict(m), "loss": 1.25, "top1": 41.5, "top5": 80.25}
    path = T.save_checkpoint(state, str(tmp_path / "snap"), "Charades", 2, 41.5, 80.25)
    assert os.path.basename(path) == "model_Charades_epoch2_top1_41.500_top5_80.250_model_best.pth.tar"     # main.py:369-373
    ck = torch.load(path)
    assert set(ck) == {"epoch", "state_dict", "loss", "top1", "top5"}
    assert all(k.startswith("module.") for k in ck["state_dict"])                                           # nn.DataParallel keys
    ck["state_dict"]["module.not_in_model"] = torch.zeros(1)                                               # ignored (main.py:108)
    dropped = "module.fcos.head.iou_scores.3.weight"
    del ck["state_dict"][dropped]                                                                           # kept from the model
    torch.save(ck, path)
    m2 = oracle_model(3)
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(1.0)
    keep = m2.fcos.head.iou_scores[3].weight.detach().clone()
    epoch, picked = T.load_checkpoint(m2, path)
    assert epoch == 3 and "not_in_model" not in picked and "fcos.head.iou_scores.3.weight" not in picked
    assert torch.equal(m2.fcos.head.iou_scores[3].weight, keep)
    assert torch.equal(m2.prop_fc.weight, m.prop_fc.weight)


def test_glove_init(tmp_path):
    m = oracle_model(1)
    table = torch.randn_like(m.query_encoder.embedding.weight)
    p = str(tmp_path / "glove_weights")
    assert not T.init_glove(m, p)
    torch.save(table, p)
    assert T.init_glove(m, p) and torch.equal(m.query_encoder.embedding.weight, table)
