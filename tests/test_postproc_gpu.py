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


@pytest.mark.parametrize("first_stage", [True, False])
@pytest.mark.parametrize("B,T", [(32, 256), (16, 1024)])
def test_device_postprocessor_matches_oracle_indices_and_levels(first_stage, B, T):
    """drn_postprocess against the ORACLE's post-processor (oracle/drn_oracle.py FCOSPostProcessor = model/inference.py:11-215)
    at the BASELINE sequence lengths: per clip and level the SET of kept location indices and the level tags must be
    identical (integer work: bit-exact), detections / scores within 2e-6 (sigmoid / sqrt on two devices)."""
    from oracle import drn_oracle as O
    from drn_amd.model.inference import FCOSPostProcessor
    Ls = [T, T // 2, T // 4]
    box_cls, box_reg, iou = _heads(B, Ls, seed=T + B + int(first_stage), empty_clip=1)
    pp = FCOSPostProcessor(0.05, 32, 0.6, 32, 0, 2, first_stage, False)
    strides = [1.0, 2.0, 4.0]
    pp.strides = strides
    locations = [torch.arange(L, device="cuda:0", dtype=torch.float32) * s + s / 2 for L, s in zip(Ls, strides)]
    got = pp.forward_flat(locations, box_cls, box_reg, iou)
    ref = O.FCOSPostProcessor({"fcos_inference_thr": 0.05, "fcos_pre_nms_top_n": 32, "is_first_stage": first_stage})
    want = ref([l.cpu() for l in locations], [t.cpu() for t in box_cls], [t.cpu() for t in box_reg], [t.cpu() for t in iou])
    assert len(got) == len(want) == B
    truncated = 0
    for b in range(B):
        lv_g = [x for l in got[b]["level"] for x in l]
        lv_w = [x for l in want[b]["level"] for x in l]
        assert lv_g == lv_w, b                                             # same count per level, levels in order
        lg, lw = got[b]["locations"].cpu().numpy() * 32, want[b]["locations"].numpy() * 32
        for lvl in set(lv_g):
            sel = np.array(lv_g) == lvl
            assert sorted(lg[sel].tolist()) == sorted(lw[sel].tolist()), (b, lvl)      # kept location indices, exact
            truncated += int(sel.sum() == 32)

        def rows(d):
            a = np.concatenate([d["detections"].cpu().numpy(), d["scores"].cpu().numpy()[:, None], d["locations"].cpu().numpy()[:, None]], 1)
            return a[np.lexsort((a[:, 0], a[:, 3]))]
        np.testing.assert_allclose(rows(got[b]), rows(want[b]), atol=2e-6, rtol=0)
    assert got[1]["level"] == [[-1]] and want[1]["level"] == [[-1]]
    assert truncated > 0                                                   # the per-level top-n truncation was exercised


@pytest.mark.parametrize("B,T", [(32, 256), (16, 1024), (3, 64)])
def test_target_labels_match_oracle_bit_exact(B, T):
    """Target assignment (model/loss.py:40-127) of drn_fcos_loss_fwd against oracle.fcos_targets: the label of every location
    and the positive count, bit-exact, including ground truths whose boundaries fall exactly on location centres (min(l, r) ==
    0 is NOT inside) and whose max(l, r) sits exactly on a size-of-interest bound (inclusive)."""
    from oracle import drn_oracle as O
    from drn_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(B + T)
    Ls, strides = [T, T // 2, T // 4], [1.0, 2.0, 4.0]
    s = torch.rand(B, generator=g) * 0.5
    gt = torch.stack([s, s + 0.05 + torch.rand(B, generator=g) * 0.4], 1)
    gt[0] = torch.tensor([2.5, 8.5]) / 32.0          # both boundaries on level-0 centres: l == 0 at 2.5 and r == 0 at 8.5 are outside
    gt[1] = torch.tensor([1.0, 13.0]) / 32.0         # level-1 centre 7: l == r == 6; level-0 centres 6.5 / 7.5: max(l, r) = 6.5 > 6
    gt[2] = torch.tensor([0.0, 22.0]) / 32.0         # level-1 centre 11: max(l, r) == 11, the inclusive upper bound of level 1
    locs = [torch.arange(L, dtype=torch.float32) * st + st / 2 for L, st in zip(Ls, strides)]
    want, _ = O.fcos_targets(locs, gt)
    R = B * sum(Ls)
    logits = torch.randn(R, 1, generator=g).to(dev)
    reg = torch.exp(torch.randn(R, 2, generator=g)).to(dev)
    levels = [(Ls[i], strides[i], float(O.SIZES_OF_INTEREST[i][0]), float(O.SIZES_OF_INTEREST[i][1])) for i in range(3)]
    out5 = torch.empty(6, device=dev)
    labels = torch.empty(R, device=dev)
    ops.fcos_loss_fwd(ops.loss_levels(levels), B, logits, reg, None, gt.to(dev), 2.0, 0.25, 32.0, 0, out5, labels=labels)
    assert torch.equal(labels.cpu(), want.float()), int((labels.cpu() != want.float()).sum())
    assert int(out5[3].item()) == int(want.sum().item()) > 0
    per_level = want.split([B * L for L in Ls])
    assert all(int(p.sum()) > 0 for p in per_level)                        # every pyramid level owns some positives


def test_results_entries_batched_copy_equals_per_clip_records():
    """metrics.results_entries concatenates all clips' detections / scores on the device and copies once per field: same records
    as results_entry clip by clip (main.py:324-348)."""
    from drn_amd.metrics import results_entries, results_entry
    g = torch.Generator().manual_seed(0)
    boxes = []
    for n in (3, 1, 5, 2):
        boxes.append({"detections": torch.rand(n, 2, generator=g).cuda(), "scores": torch.rand(n, generator=g).cuda(), "labels": [],
                      "level": [[0] * (n - 1), [1]], "locations": torch.rand(n, generator=g).cuda()})
    queries = ["q%d" % i for i in range(4)]
    gts = torch.rand(4, 2, generator=g).numpy()
    assert results_entries(queries, gts, boxes) == [results_entry(q, t, b) for q, t, b in zip(queries, gts, boxes)]
