"""drn_amd.metrics (R@k / IoU, temporal NMS) against outputs recorded from the reference's PostProcessRunner
(utils/evaluate_utils.py, fixture tests/golden/metrics.json made by tests/golden/gen_golden.py metrics)."""
import json
import os

import pytest

from drn_amd.metrics import PostProcessRunner

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.json")))


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: "nms" if c["temporal_nms"] else "nonms")
def test_run_evaluate_matches_reference(case):
    runner = PostProcessRunner(json.loads(json.dumps(GOLD["results"])))
    topks, acc = runner.run_evaluate(iou_topk_dict=case["iou_topk"], temporal_nms=case["temporal_nms"])
    assert list(topks) == case["topks"]
    assert acc == case["accuracy"]                       # counts / counts: exact
    got = {vid: [{"node_predictions": it["node_predictions"], "level": it["level"]} for it in items]
           for vid, items in runner.viz_processed_results.items()}
    assert got == case["last_setting_picks"]             # surviving predictions (order included) and their level tags


def test_nms_temporal_and_iou_match_reference():
    for c in GOLD["nms_cases"]:
        assert PostProcessRunner.nms_temporal(c["x1"], c["x2"], c["s"], c["overlap"]) == c["pick"]
    for c in GOLD["iou_cases"]:
        assert PostProcessRunner.calculate_IoU(c["a"], c["b"]) == c["iou"]
    assert PostProcessRunner.calculate_IoU((0.1, 0.2), (0.6, 0.9)) < 0      # un-clamped, evaluate_utils.py:228-232
    assert PostProcessRunner.nms_temporal([], [], [], 0.45) == []


def test_dead_branches_raise():
    r = PostProcessRunner({"v": [{"query": "q", "gt": [0.1, 0.4], "node_predictions": [[0.1, 0.4, 0.9]], "level": [[0]]}]})
    with pytest.raises(NotImplementedError):
        r.run_evaluate({"iou": [0.5], "topk": [1]}, do_merge=True)
    assert r.run_evaluate({"iou": [0.5], "topk": [1]})[1] == [1.0]


def test_results_entries_equals_per_clip_records():
    """The batched record builder (one device->host copy per field) writes what results_entry writes clip by clip."""
    import torch
    from drn_amd.metrics import results_entries, results_entry
    g = torch.Generator().manual_seed(0)
    boxes = []
    for n in (3, 1, 5, 2):
        boxes.append({"detections": torch.rand(n, 2, generator=g), "scores": torch.rand(n, generator=g), "labels": [],
                      "level": [[0] * (n - 1), [1]], "locations": torch.rand(n, generator=g)})
    queries = ["q%d" % i for i in range(4)]
    gts = torch.rand(4, 2, generator=g).numpy()
    assert results_entries(queries, gts, boxes) == [results_entry(q, t, b) for q, t, b in zip(queries, gts, boxes)]
