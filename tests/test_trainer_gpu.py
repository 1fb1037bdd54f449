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


def _varying_batches(n, B, T, D, seed=3):
    """n synthetic batches whose longest query differs (3, 5, 7 words -> padded lengths 4 and 8), the last one ragged (B - 1 clips)."""
    from drn_amd.utils.synthetic import VOCAB_SIZE, synthetic_batch
    out = []
    for i in range(n):
        nb = B - 1 if i == n - 1 else B
        b = list(synthetic_batch(nb, T, D, seed=seed + i))
        g = torch.Generator().manual_seed(100 + i)
        lmax = 3 + (i * 2) % 6                                  # 3, 5, 7, 3, ...
        lens = torch.sort(torch.randint(2, lmax + 1, (nb,), generator=g), descending=True)[0]
        lens[0] = lmax                                          # (sorted descending: the first clip carries the longest query)
        tok = torch.zeros((nb, lmax), dtype=torch.int64)
        for r in range(nb):
            tok[r, :int(lens[r])] = torch.randint(1, VOCAB_SIZE + 1, (int(lens[r]),), generator=g)
        b[0], b[1] = tok, lens
        out.append(b)
    return out


@pytest.mark.parametrize("forked", [False, True])
@pytest.mark.parametrize("dtype,stage", [(torch.bfloat16, 1), (torch.float32, 3)])
def test_graph_trainer_is_bit_identical_to_eager(dtype, stage, forked):
    """Trainer(graph=True) -- train.py's default -- replays each step as a hipGraph from static input buffers, tokens padded to
    the geometry's query length; fed batches of varying query length and a ragged last batch it must leave EVERY parameter and
    buffer bit-identical to the eager trainer, and report the same losses (main.py:198-252 is the loop both implement)."""
    from drn_amd import functional as DF
    from drn_amd import trainer as T
    B, Tp, D, n = 4, 32, 64, 13
    batches = _varying_batches(n, B, Tp, D)
    order = [0, 1, 2, 0, 1, 2, 0, 1, 2, 12, 0, 1, 2]           # every geometry passes warm-up, capture and replay; one ragged step
    runs = []
    for graph in (False, True):
        m = hip_model(stage)
        m.set_compute_dtype(dtype)
        tr = T.Trainer(m, stage, lr=1e-4 if stage == 1 else 1.0, clip_gradient=0.5, graph=graph, lq_bucket=4,
                       forked=forked)                            # (stage 3 divides lr by 1e4; forked: one graph, two branches)
        assert tr.graph == graph and tr.forked == (forked and graph)
        losses = []
        for i in order:
            b = batches[i]
            args = b if graph else [t.to("cuda:0") for t in b]   # graph mode takes the host batch as the loader yields it
            ld = tr.train_step(args)
            losses.append([float(ld[k].detach().reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
        torch.cuda.synchronize()
        DF.flush_bn_counters()
        if graph:
            assert sum(s.graph is not None for s in tr._slots.values()) >= 2, "no step was ever replayed"
        runs.append((losses, {k: v.detach().clone() for k, v in m.state_dict().items()}))
    (l0, s0), (l1, s1) = runs
    assert np.array_equal(np.array(l0), np.array(l1)), (l0, l1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), (k, float((s0[k].float() - s1[k].float()).abs().max()))


def test_train_epoch_prefetches_host_batches_without_changing_anything():
    """Trainer.train_epoch in graph mode moves the NEXT host batch to the device on a copy stream while the current step runs
    (one batch of look-ahead); the epoch must leave the model exactly where the same batches handed over device-resident leave
    it, whatever the memory kind (pinned or pageable) and with geometries alternating from step to step."""
    from drn_amd import functional as DF
    from drn_amd import trainer as T
    B, Tp, D, n = 4, 32, 64, 6
    raw = _varying_batches(n, B, Tp, D)
    as_loader = lambda bs: [(["v"] * B, b[3], b[2], b[4], b[0], b[1], b[5], b[6]) for b in bs]     # collate_data's 8-tuple
    # "bf16 host": the features handed over in the compute dtype, as drn_amd.data.collate_data(feature_dtype=torch.bfloat16)
    # builds them in the DataLoader workers (train.py's default for a bf16 model): rounded on the host by the rule the step's
    # cast kernel applies on the device, so half the bytes cross PCIe and nothing changes
    kinds = {"device": [[t.to("cuda:0") if torch.is_tensor(t) else t for t in b] for b in raw],
             "pinned": [[t.pin_memory() if torch.is_tensor(t) else t for t in b] for b in raw],
             "pageable": raw,
             "bf16 host": [[(t.bfloat16() if i == 2 else t).pin_memory() if torch.is_tensor(t) else t for i, t in enumerate(b)] for b in raw],
             "bf16 device": [[(t.bfloat16() if i == 2 else t).to("cuda:0") if torch.is_tensor(t) else t for i, t in enumerate(b)] for b in raw]}
    states, means = {}, {}
    for kind, bs in kinds.items():
        m = hip_model(1)
        m.set_compute_dtype(torch.bfloat16)
        tr = T.Trainer(m, 1, lr=1e-4, clip_gradient=0.5, graph=True, lq_bucket=4)
        means[kind] = [tr.train_epoch(as_loader(bs)) for _ in range(3)]      # warm-up, capture, replay
        torch.cuda.synchronize()
        DF.flush_bn_counters()
        assert any(s.graph is not None for s in tr._slots.values())
        states[kind] = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for kind in ("pinned", "pageable", "bf16 host", "bf16 device"):
        assert means[kind] == means["device"], (kind, means[kind], means["device"])
        for k in states["device"]:
            assert torch.equal(states[kind][k], states["device"][k]), (kind, k)


def test_graph_trainer_over_a_real_dataloader_with_pinning_workers(tmp_path):
    """train.py's loop as a user runs it: Trainer(graph=True).train_epoch over DataLoader(num_workers > 0, pin_memory=True) with
    the bf16 hand-over of the features.  The DataLoader's pin_memory THREAD calls hipHostMalloc while the first steps are being
    captured -- in the default (global) capture error mode that fails the capture; the captures run thread_local."""
    import functools
    from torch.utils.data import DataLoader
    from drn_amd import functional as DF
    from drn_amd import trainer as T
    from drn_amd.data import CharadesSTA, collate_data
    from drn_amd.utils.synthetic import default_cfg
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.join(here, "golden", "charades_mini")
    cfg = default_cfg("TINY", 12, 1)
    cfg["feature_type"] = "C3D"
    cfg["C3D"] = {"feature_root": "./features", "feature_dim": 12, "ft_window_size": 16, "ft_overlap": 0.5}
    cfg["props_file_path"] = "./data/dataset/Charades/mini_props.txt"
    ds = CharadesSTA(cfg, "train", root, lambda s: s.split())
    runs = {}
    for mode in ("loader", "plain"):
        m = hip_model(1, cfg=cfg)
        m.set_compute_dtype(torch.bfloat16)
        tr = T.Trainer(m, 1, lr=1e-4, graph=True)
        collate = functools.partial(collate_data, feature_dtype=torch.bfloat16)
        if mode == "loader":
            loader = DataLoader(ds, batch_size=4, shuffle=False, collate_fn=collate, num_workers=2, pin_memory=True, drop_last=True)
        else:
            loader = list(DataLoader(ds, batch_size=4, shuffle=False, collate_fn=collate, drop_last=True))
        means = [tr.train_epoch(loader, e) for e in range(4)]          # warm-up, capture (while the pin thread works), replay
        torch.cuda.synchronize()
        DF.flush_bn_counters()
        assert any(s.graph is not None for s in tr._slots.values()), "no step was ever captured"
        assert all(np.isfinite(means))
        runs[mode] = (means, {k: v.detach().clone() for k, v in m.state_dict().items()})
    assert runs["loader"][0] == runs["plain"][0]
    for k, v in runs["plain"][1].items():
        assert torch.equal(v, runs["loader"][1][k]), k
