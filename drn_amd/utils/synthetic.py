"""Synthetic configs, inputs and seeded weights for tests and bench.py.

Everything is derived from numpy `default_rng` seeds so the golden generator
(which imports the reference), the CPU oracle and the HIP path all see the
same weights and inputs without storing them (182 MB at D=4096).
Shapes and distributions follow SURVEY.md section 8(d).
"""
import argparse
import zlib

import numpy as np
import torch

VOCAB_SIZE = 1301  # data/dataset/Charades vocab


def default_cfg(feature_type="C3D", feature_dim=None, stage=1):
    """The keys of data/default_config.yaml['Charades'] that the hot path reads
    (SURVEY section 5), plus the stage flags set by main.py's CLI."""
    dims = {"MFnet": 768, "C3D": 4096, "I3D": 2048}
    cfg = {
        "feature_type": feature_type,
        "first_output_dim": 256,
        "fpn_feature_dim": 512,
        "fpn_stride": [1, 2, 4],
        "lstm_layers": 1,
        "fcos_conv_layers": 1,
        "fcos_prior_prob": 0.01,
        "fcos_loss_alpha": 0.25,
        "fcos_loss_gamma": 2.0,
        "fcos_inference_thr": 0.05,
        "fcos_pre_nms_top_n": 32,
        "fcos_nms_thr": 0.6,
        "fcos_num_class": 2,
        "test_detections_per_img": 32,
        "is_first_stage": stage == 1,
        "is_second_stage": stage == 2,
        "is_third_stage": stage == 3,
    }
    for k, v in dims.items():
        cfg[k] = {"feature_dim": v}
    if feature_dim is not None:
        cfg[feature_type] = {"feature_dim": int(feature_dim)}
    return cfg


def as_namespace(cfg):
    return argparse.Namespace(**cfg)


def _rng(seed, key):
    return np.random.default_rng([int(seed), zlib.crc32(key.encode())])


def seeded_tensor(key, shape, dtype, seed):
    """Deterministic value for one state_dict entry, independent of key order."""
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=dtype)
    if key.endswith("running_mean"):
        return torch.zeros(shape, dtype=dtype)
    if key.endswith("running_var"):
        return torch.ones(shape, dtype=dtype)
    g = _rng(seed, key)
    if key.endswith("embedding.weight"):
        w = g.uniform(-0.5, 0.5, size=shape).astype(np.float32)
        w[0] = 0.0                                    # padding_idx=0
        return torch.from_numpy(w)
    if key.endswith(".scale"):
        return torch.from_numpy(g.uniform(0.8, 1.2, size=shape).astype(np.float32))
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        b = np.sqrt(3.0 / fan_in)
        return torch.from_numpy(g.uniform(-b, b, size=shape).astype(np.float32))
    if key.endswith("weight"):                        # BN gamma
        return torch.from_numpy(g.uniform(0.5, 1.5, size=shape).astype(np.float32))
    return torch.from_numpy(g.uniform(-0.1, 0.1, size=shape).astype(np.float32))   # biases / BN beta


def seeded_state_dict(model, seed=0):
    """Seeded replacement for every entry of model.state_dict() (same keys/shapes)."""
    return {k: seeded_tensor(k, v.shape, v.dtype, seed) for k, v in model.state_dict().items()}


def synthetic_batch(B, T, D, seed=1, device="cpu"):
    """The 7 positional inputs of mainModel.forward (model/main_model.py:42-43)."""
    g = np.random.default_rng(int(seed))
    feats = g.random((B, T, D), dtype=np.float32)
    t = np.arange(T, dtype=np.float64)
    pse = np.broadcast_to(np.stack([t / T, (t + 1) / T], -1)[None], (B, T, 2)).copy()
    s = g.uniform(0.0, 0.5, size=B)
    w = g.uniform(0.1, 0.5, size=B)
    gt = np.stack([s, s + w], -1)
    lens = np.sort(g.integers(3, 9, size=B))[::-1].copy()
    tokens = np.zeros((B, int(lens.max())), dtype=np.int64)
    for b in range(B):
        tokens[b, :lens[b]] = g.integers(1, VOCAB_SIZE + 1, size=lens[b])
    dev = torch.device(device)
    return (torch.from_numpy(tokens).to(dev), torch.from_numpy(lens.astype(np.int64)),
            torch.from_numpy(feats).to(dev), torch.from_numpy(pse).to(dev),
            torch.from_numpy(gt).to(dev), torch.full((B,), T, dtype=torch.int64),
            torch.full((B,), 16 * T, dtype=torch.int64))


def planted_batches(n, B, T, D, seed=0, noise=0.3):
    """n synthetic batches in drn_amd.data.collate_data's 8-tuple format whose FEATURES CARRY THE ANSWER: every clip has one
    ground-truth segment, proposals inside it hold a fixed pattern vector (+ noise), the others noise only -- a planted signal a
    model can learn to localise in a few hundred steps (there is no Charades-STA feature file in this repository: the only R@1
    evidence obtainable is relative, between two runs of this task).  Uniform proposals (t/T, (t+1)/T) as the reference's
    MAN-32 props file gives; segment lengths 0.15-0.6 of the clip; random queries (they carry no information)."""
    g = np.random.default_rng(int(seed))
    pattern = np.random.default_rng(12345).standard_normal(D).astype(np.float32)
    t = np.arange(T, dtype=np.float64)
    pse1 = np.stack([t / T, (t + 1) / T], -1)
    centres = (t + 0.5) / T
    out = []
    for i in range(n):
        s = g.uniform(0.0, 0.4, size=B)
        w = g.uniform(0.15, 0.6, size=B)
        e = np.minimum(s + w, 1.0)
        inside = ((centres[None, :] >= s[:, None]) & (centres[None, :] <= e[:, None])).astype(np.float32)
        feats = np.abs(noise * g.standard_normal((B, T, D)).astype(np.float32) + inside[:, :, None] * pattern[None, None, :])
        lens = np.sort(g.integers(3, 9, size=B))[::-1].copy()
        tokens = np.zeros((B, int(lens.max())), dtype=np.int64)
        for b in range(B):
            tokens[b, :lens[b]] = g.integers(1, VOCAB_SIZE + 1, size=lens[b])
        names = ["clip%d_%d" % (i, b) for b in range(B)]
        out.append((names, torch.from_numpy(np.broadcast_to(pse1[None], (B, T, 2)).copy()), torch.from_numpy(feats),
                    torch.from_numpy(np.stack([s, e], -1)), torch.from_numpy(tokens), torch.from_numpy(lens.astype(np.int64)),
                    torch.full((B,), T, dtype=torch.int64), torch.full((B,), 16 * T, dtype=torch.int64)))
    return out
