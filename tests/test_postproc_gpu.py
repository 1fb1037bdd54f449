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


def _host_first_hits(det, scores, counts, gts, ious, K):
    """The reference evaluator (drn_amd.metrics.PostProcessRunner, pinned to utils/evaluate_utils.py by tests/test_metrics_cpu.py)
    applied clip by clip: position of the first NMS survivor among the first K that hits, else K."""
    import numpy as np
    from drn_amd.metrics import PostProcessRunner as R
    out = np.full((det.shape[0], len(ious)), K, dtype=np.int64)
    for b in range(det.shape[0]):
        n = int(counts[b].sum())
        preds = [[0.0, 1.0, 1.0]] if n == 0 else np.concatenate([det[b, :n], scores[b, :n, None]], axis=1).tolist()
        preds = sorted(preds, key=lambda x: x[-1], reverse=True)
        for q, iou in enumerate(ious):
            picks = R.nms_temporal([p[0] for p in preds], [p[1] for p in preds], [p[-1] for p in preds], iou - 0.05)
            for pos, i in enumerate(picks[:K]):
                if R.calculate_IoU((gts[b][0], gts[b][1]), (preds[i][0], preds[i][1])) >= iou:
                    out[b, q] = pos
                    break
    return out


@pytest.mark.parametrize("gt_dtype", [torch.float64, torch.float32])
def test_device_recall_matches_the_host_evaluator(gt_dtype):
    """drn_eval_recall (temporal NMS + first hit among the survivors, on the device) against the host evaluator on random
    detections: score ties (visited from the later prediction), zero-length segments (0/0 in the NMS overlap), duplicates,
    clips without any detection (the (0, 1) default), ground truths that nothing reaches."""
    import numpy as np
    from drn_amd import ops
    g = torch.Generator().manual_seed(5)
    B, R, nl, K = 64, 96, 3, 5
    ious = [0.3, 0.5, 0.7]
    a = torch.rand(B, R, generator=g)
    w = torch.rand(B, R, generator=g) * 0.5
    det = torch.stack([a.clamp(0, 1), (a + w).clamp(0, 1)], dim=2).float()
    scores = (torch.randint(0, 12, (B, R), generator=g).float() / 12.0 + 0.05).sqrt()        # many exact ties
    det[:, 5] = det[:, 4]                                    # duplicates
    det[:, 7, 1] = det[:, 7, 0]                              # empty segments
    det[:, 8] = det[:, 7]                                    # two equal empty segments: 0/0
    counts = torch.randint(0, 33, (B, nl), generator=g).int()
    counts[0] = 0                                            # no detection at all
    counts[1] = torch.tensor([1, 0, 0])
    gts = torch.sort(torch.rand(B, 2, generator=g, dtype=torch.float64), dim=1)[0]
    gts[2] = torch.tensor([0.0, 1.0])
    gts = gts.to(gt_dtype)
    dev = "cuda:0"
    fh = ops.eval_recall(det.to(dev), scores.to(dev), counts.to(dev), gts.to(dev), torch.tensor(ious, dtype=torch.float64, device=dev), K)
    want = _host_first_hits(det.numpy(), scores.numpy(), counts.numpy(), gts.double().numpy(), ious, K)
    got = fh.cpu().numpy()
    assert np.array_equal(got, want), np.argwhere(got != want)[:10]
    assert (want < K).any() and (want == K).any() and (want > 0).any()


def test_fast_evaluate_equals_the_record_building_path():
    """Trainer.evaluate(with_results=False) -- device post-processor buffers -> drn_eval_recall, no per-batch host sync -- returns
    the loss and Recall@k of the default path that builds the reference's raw-results records and runs the host evaluator
    (main.py:270-366), on a model whose classifier passes ~every location so that top-n truncation and NMS have work to do."""
    from drn_amd import trainer as T
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict, synthetic_batch
    for stage in (1, 3):
        cfg = default_cfg("TINY", 64, stage)
        m = mainModel(VOCAB_SIZE, as_namespace(cfg))
        m.load_state_dict(seeded_state_dict(m, 0))
        with torch.no_grad():
            m.fcos.head.cls_logits.bias.fill_(0.5)
            m.fcos.head.cls_logits.weight.mul_(30.0)
        m = m.to("cuda:0")
        tr = T.Trainer(m, stage, lr=1e-3)
        batches = []
        for i in range(5):
            tok, qlen, feats, pse, gt, nprops, nframes = synthetic_batch(8, 64, 64, seed=40 + i)
            gt = pse[:, 5 + i, :].clone()                     # a proposal as ground truth: some queries are answerable
            batches.append((["v%d" % (i % 3)] * 8, pse, feats, gt, tok, qlen, nprops, nframes))
        iou_topk = {"iou": [0.3, 0.5, 0.7], "topk": [1, 5]}
        l0, k0, a0, res = tr.evaluate(batches, iou_topk=iou_topk)
        l1, k1, a1, none = tr.evaluate(batches, iou_topk=iou_topk, with_results=False)
        assert none is None and res and k0 == k1
        assert a0 == a1, (a0, a1)
        assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0))
        assert sum(len(v) for v in res.values()) == 40
        n_preds = [len(e["node_predictions"]) for v in res.values() for e in v]
        assert max(n_preds) > 32, "the classifier should pass enough locations for top-n / NMS to matter"
