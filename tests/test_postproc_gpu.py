"""drn_postprocess (one kernel per batch) against the reference-shaped host loop of the same module
(drn_amd/model/inference.py: forward_for_single_feature_map / select_over_all_levels, model/inference.py:51-199) on random
head outputs: more candidates than top_n on the fine level, a clip with no candidate at all, first and later stages."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _heads(B, Ls, seed, empty_clip=None):
    from drn_amd.model.fcos import FCOSHead
    g = torch.Generator().manual_seed(seed)
    R = B * sum(Ls)
    logits = (torch.randn(R, 1, generator=g) * 2.0 - 1.0)
    reg = torch.exp(torch.randn(R, 2, generator=g))
    iou = torch.randn(R, 1, generator=g)
    if empty_clip is not None:
        r = 0
        for L in Ls:
            logits[r + empty_clip * L: r + (empty_clip + 1) * L] = -20.0
            r += B * L
    geo = [(B, L) for L in Ls]
    dev = "cuda:0"
    return [FCOSHead.split_levels(t.to(dev), geo) for t in (logits, reg, iou)]


@pytest.mark.parametrize("first_stage", [True, False])
@pytest.mark.parametrize("B,Ls,empty", [(4, [64, 32, 16], 2), (3, [256, 128, 64], None), (2, [1024, 512, 256], 0)])
def test_device_postprocessor_matches_host_loop(first_stage, B, Ls, empty):
    from drn_amd.model.inference import FCOSPostProcessor
    box_cls, box_reg, iou = _heads(B, Ls, seed=B + len(Ls) + int(first_stage), empty_clip=empty)
    pp = FCOSPostProcessor(0.05, 32, 0.6, 32, 0, 2, first_stage, False)
    strides = [1.0, 2.0, 4.0]
    pp.strides = strides
    locations = [torch.arange(L, device="cuda:0", dtype=torch.float32) * s + s / 2 for L, s in zip(Ls, strides)]
    got = pp.forward_flat(locations, box_cls, box_reg, iou)
    sampled = [pp.forward_for_single_feature_map(l, o, b, i, s)
               for i, (l, o, b, s) in enumerate(zip(locations, box_cls, box_reg, iou))]
    want = pp.select_over_all_levels(list(zip(*sampled)))
    assert len(got) == len(want) == B
    for b in range(B):
        assert got[b]["detections"].shape == want[b]["detections"].shape, b
        for d in (got[b], want[b]):
            assert len(d["scores"]) == len(d["detections"]) == len(d["locations"])

        def rows(d):
            a = np.concatenate([d["detections"].cpu().numpy(), d["scores"].cpu().numpy()[:, None], d["locations"].cpu().numpy()[:, None]], 1)
            return a[np.lexsort((a[:, 0], a[:, 3], a[:, 2]))]
        np.testing.assert_allclose(rows(got[b]), rows(want[b]), atol=2e-6, rtol=0)
        lv_g = sorted(x for l in got[b]["level"] for x in l)
        lv_w = sorted(x for l in want[b]["level"] for x in l)
        assert lv_g == lv_w
    if empty is not None:
        assert got[empty]["level"] == [[-1]] and got[empty]["detections"].tolist() == [[0.0, 1.0]]
    assert max(len(d["detections"]) for d in got) > 32          # top_n per LEVEL, so a clip can carry more than 32
