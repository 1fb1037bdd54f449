"""drn_amd.trainer on the MI355X (HIP model; fused clip+Adam over the flat buckets in stages 1 and 3, torch's optimizer in
stage 2) against the loss trajectories recorded from the reference model under main.py's loop, plus an end-to-end
fit() -> evaluate() -> checkpoint round trip on the committed mini dataset."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_trainer_cpu import check_trajectory, traj, traj_batches      # noqa: E402  (same goldens, same gate)


def hip_model(stage, D=64, cfg=None):
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict
    m = mainModel(VOCAB_SIZE, as_namespace(cfg or default_cfg("TINY", D, stage)))
    m.load_state_dict(seeded_state_dict(m, 0))
    return m.to("cuda:0")


@pytest.mark.parametrize("stage", [1, 2, 3, "1_lowlr"])
def test_hip_trainer_follows_reference_trajectory(stage):
    from drn_amd import trainer as T
    g = traj(stage)
    chaotic = stage == 1          # lr = 1e-3: rounding-level gradient differences are amplified ~30x per step (see STEP_RTOL)
    stage = int(g["stage"])
    m = hip_model(stage)
    tr = T.Trainer(m, stage, lr=float(g["lr"]), clip_gradient=0.5)
    assert tr.fused == (stage != 2)
    batches = [[t.to("cuda:0") for t in b] for b in traj_batches(g)]
    got = []
    for it in range(int(g["steps"])):
        ld = tr.train_step(batches[it % 2])
        got.append([float(ld[k].detach().reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
    from drn_amd import functional as DF
    DF.flush_bn_counters()
    if chaotic:                   # only the first two steps are comparable at that learning rate
        np.testing.assert_allclose(np.array(got)[0], g["losses"][0], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(np.array(got)[1], g["losses"][1], rtol=5e-3, atol=1e-4)
        assert np.isfinite(np.array(got)).all()
    else:
        check_trajectory(np.array(got), g, {k: v.cpu() for k, v in m.state_dict().items()}, scale=5.0)


def test_fit_evaluate_checkpoint_on_mini_dataset(tmp_path):
    from torch.utils.data import DataLoader
    from drn_amd import trainer as T
    from drn_amd.data import CharadesSTA, collate_data
    from drn_amd.utils.synthetic import default_cfg
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.join(here, "golden", "charades_mini")
    cfg = default_cfg("TINY", 12, 3)
    cfg["feature_type"] = "C3D"
    cfg["C3D"] = {"feature_root": "./features", "feature_dim": 12, "ft_window_size": 16, "ft_overlap": 0.5}
    cfg["props_file_path"] = "./data/dataset/Charades/mini_props.txt"
    tok = lambda s: s.split()
    train = DataLoader(CharadesSTA(cfg, "train", root, tok), batch_size=4, shuffle=False, collate_fn=collate_data)
    test = DataLoader(CharadesSTA(cfg, "test", root, tok), batch_size=4, shuffle=False, collate_fn=collate_data)
    m = hip_model(3, cfg=cfg)
    tr = T.Trainer(m, 3, lr=1e-3)
    hist = tr.fit(train, test, n_epoch=2, snapshot_pref=str(tmp_path / "snap"), id2word=None)
    assert len(hist) == 2 and all(np.isfinite(h["train_loss"]) and 0.0 <= h["top1"] <= h["top5"] <= 100.0 for h in hist)
    val_loss, topks, accs, results = tr.evaluate(test)
    assert topks == [1, 5] and len(results) > 0
    rec = next(iter(results.values()))[0]
    assert set(rec) == {"query", "gt", "node_predictions", "edge_predictions", "level"} and len(rec["node_predictions"][0]) == 3
    files = os.listdir(str(tmp_path / "snap")) if os.path.isdir(str(tmp_path / "snap")) else []
    if files:                                                                # a best checkpoint exists once R@k > 0
        m2 = hip_model(3, cfg=cfg)
        T.load_checkpoint(m2, os.path.join(str(tmp_path / "snap"), files[0]), map_location="cuda:0")
