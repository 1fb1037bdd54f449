"""drn_amd.data (CharadesSTA reader + collate_data) against batches recorded from the reference's dataset.py on the
committed synthetic mini dataset (tests/golden/charades_mini, tests/golden/dataset.npz; generator:
tests/golden/gen_golden.py dataset): proposal/feature window arithmetic, duration-normalised ground truth, end frames
past the video, fewer stored segments than frames, sort-by-query-length collation."""
import os

import numpy as np
import pytest
import torch

from drn_amd.data import CharadesSTA, collate_data, default_tokenizer

HERE = os.path.dirname(os.path.abspath(__file__))
MINI = os.path.join(HERE, "golden", "charades_mini")
GOLD = np.load(os.path.join(HERE, "golden", "dataset.npz"))
CFG = {"feature_type": "C3D", "C3D": {"feature_root": "./features", "feature_dim": 12, "ft_window_size": 16, "ft_overlap": 0.5},
       "props_file_path": "./data/dataset/Charades/mini_props.txt"}


@pytest.mark.parametrize("split", ["train", "test"])
def test_batches_match_reference(split):
    ds = CharadesSTA(CFG, split=split, root=MINI, tokenizer=lambda s: s.split())
    names, pse, feats, gt, tok, qlen, nprops, nframes = collate_data([ds[i] for i in range(len(ds))])
    assert names == GOLD[split + "/names"].tolist()
    assert pse.dtype == torch.float64 and gt.dtype == torch.float64 and tok.dtype == torch.int64 and qlen.dtype == torch.int64
    np.testing.assert_array_equal(pse.numpy(), GOLD[split + "/props_s_e"])
    np.testing.assert_array_equal(feats.numpy(), GOLD[split + "/feats"])
    np.testing.assert_array_equal(gt.numpy(), GOLD[split + "/gt"])
    np.testing.assert_array_equal(tok.numpy(), GOLD[split + "/tokens"])
    np.testing.assert_array_equal(qlen.numpy(), GOLD[split + "/qlen"])
    np.testing.assert_array_equal(nprops.numpy(), GOLD[split + "/nprops"])
    np.testing.assert_array_equal(nframes.numpy(), GOLD[split + "/nframes"])
    assert (np.diff(qlen.numpy()) <= 0).all()                     # pack_padded_sequence order (language_module.py:42)


def test_fallback_tokenizer_on_charades_style_sentences():
    tok = default_tokenizer()
    assert tok("person opens the door") == ["person", "opens", "the", "door"]
    assert tok("a person sits down , then eats") == ["a", "person", "sits", "down", ",", "then", "eats"]


def test_dataloader_round_trip():
    from torch.utils.data import DataLoader
    ds = CharadesSTA(CFG, split="train", root=MINI, tokenizer=lambda s: s.split())
    batches = list(DataLoader(ds, batch_size=3, shuffle=False, collate_fn=collate_data))
    assert sum(len(b[0]) for b in batches) == len(ds)
    for b in batches:
        assert b[2].shape[:2] == b[1].shape[:2] and b[2].shape[2] == 12 and b[4].shape[1] == int(b[5].max())


def test_collate_in_the_compute_dtype_rounds_like_the_device_cast():
    """collate_data(feature_dtype=torch.bfloat16) -- what train.py hands a bf16 model -- equals the fp32 collate rounded to
    nearest-even, element for element (the rule drn_cast_transpose applies on the device), and changes nothing else."""
    import torch
    from drn_amd.data import collate_data
    g = torch.Generator().manual_seed(0)
    batch = []
    for i, (nprops, qlen) in enumerate([(5, 3), (7, 6), (4, 6)]):
        batch.append(("v%d" % i, torch.rand(nprops, 2, generator=g, dtype=torch.float64), torch.randn(nprops, 24, generator=g) * 3,
                      (0.1 * i, 0.5 + 0.1 * i), torch.randint(1, 50, (qlen,), generator=g), qlen, nprops, 100 + i))
    a = collate_data(batch)
    b = collate_data(batch, feature_dtype=torch.bfloat16)
    assert b[2].dtype == torch.bfloat16 and torch.equal(b[2], a[2].bfloat16())
    for i in (1, 3, 4, 5, 6, 7):
        assert torch.equal(a[i], b[i])
    assert a[0] == b[0]
