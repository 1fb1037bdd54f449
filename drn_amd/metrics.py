"""Recall@k / temporal-IoU metric of the DRN evaluator (reference: utils/evaluate_utils.py:13-16,91-215,328-354 as driven
by main.py:324-364) -- SURVEY row 8f-2.  Host-side: a few dozen (start, end, score) triples per query.

Input format (main.py:336-348): {video: [{"query": str, "gt": [s, e], "node_predictions": [[s, e, score], ...],
"level": [[l]*n_l, ...]}, ...]}.  Reproduced reference behaviour, quirks included:
  * predictions are sorted by score, descending, STABLE (ties keep their input order);
  * the NMS threshold is iou_thresh - 0.05 and NMS visits candidates from the highest score down, resolving score ties
    in favour of the LATER index (ascending stable sort walked from its end);
  * `calculate_IoU` is not clamped: disjoint segments give a negative value (never >= a positive threshold);
  * "level" is carried along UNSORTED, so the per-pick level recorded for visualisation indexes the original order.
The graph-merging branch (`do_merge=True`) is dead in the reference (it calls a method with a keyword it does not accept,
evaluate_utils.py:336, and reads a hard-coded pickle, :69) and is not provided."""
import json
from copy import deepcopy

import numpy as np


class PostProcessRunner(object):
    def __init__(self, raw_results):
        self.raw_results = raw_results if isinstance(raw_results, dict) else json.load(open(raw_results, "r"))
        self.processed_results = None
        self.viz_processed_results = None

    # -- evaluate_utils.py:91-107
    def _postprocess_raw_results_no_merge(self):
        processed = {}
        for vid, items in self.raw_results.items():
            for it in items:
                preds = sorted(it["node_predictions"], key=lambda x: x[-1], reverse=True)
                processed.setdefault(vid, []).append({"query": it["query"], "gt": it["gt"], "node_predictions": preds,
                                                      "level": it["level"]})
        self.processed_results = processed

    # -- evaluate_utils.py:186-212
    @staticmethod
    def nms_temporal(x1, x2, s, overlap):
        assert len(x1) == len(s) and len(x2) == len(s)
        if len(x1) == 0:
            return []
        x1, x2, s = np.asarray(x1, dtype=np.float64), np.asarray(x2, dtype=np.float64), np.asarray(s, dtype=np.float64)
        length = x2 - x1
        order = list(np.argsort(s, kind="stable"))        # ascending, ties in input order; visited from the end
        pick = []
        while order:
            i = order.pop()
            pick.append(int(i))
            if not order:
                break
            rest = np.asarray(order)
            inter = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
            with np.errstate(invalid="ignore", divide="ignore"):      # two empty segments: 0/0 = NaN, "not <= overlap" -> dropped, as in the reference
                o = inter / (length[i] + length[rest] - inter)
            order = [int(j) for j, keep in zip(rest, o <= overlap) if keep]
        return pick

    # -- evaluate_utils.py:228-232 (un-clamped)
    @staticmethod
    def calculate_IoU(i0, i1):
        union = (min(i0[0], i1[0]), max(i0[1], i1[1]))
        inter = (max(i0[0], i1[0]), min(i0[1], i1[1]))
        return 1.0 * (inter[1] - inter[0]) / (union[1] - union[0])

    # -- evaluate_utils.py:131-184
    def compute_IoU_recall_top_n_ours(self, top_n, iou_thresh, nms=False):
        correct, total = 0.0, 0.0
        picked = {}
        for vid, items in self.processed_results.items():
            for it in items:
                total += 1
                gt = it["gt"]
                preds = it["node_predictions"]
                if nms:
                    picks = self.nms_temporal([p[0] for p in preds], [p[1] for p in preds], [p[-1] for p in preds],
                                              iou_thresh - 0.05)
                else:
                    picks = list(range(len(preds)))
                merged_level = np.array([x for lv in it["level"] for x in lv])
                # (an independent record as the reference's deepcopy makes, without copying the predictions NMS drops: the deepcopy
                # of every item for every (IoU, top-k) pair was 70 % of Trainer.evaluate's time)
                rec = {k: deepcopy(v) for k, v in it.items() if k not in ("node_predictions", "level")}
                rec["node_predictions"] = [list(preds[i]) for i in picks] if nms else [list(p) for p in preds]
                rec["level"] = merged_level[picks].tolist()
                picked.setdefault(vid, []).append(rec)
                for i in picks[:top_n] if top_n < len(picks) else picks:
                    if self.calculate_IoU((gt[0], gt[1]), (preds[i][0], preds[i][1])) >= iou_thresh:
                        correct += 1
                        break
        self.viz_processed_results = picked
        return correct, total, correct / total

    # -- evaluate_utils.py:328-354
    def run_evaluate(self, iou_topk_dict, do_merge=False, update_score=False, score_weight=1.0, temporal_nms=False, viz_nms=True,
                     do_viz=""):
        assert isinstance(iou_topk_dict, dict)
        if do_merge:
            raise NotImplementedError("do_merge=True is dead code in the reference (evaluate_utils.py:336 raises TypeError)")
        if do_viz:
            raise NotImplementedError("plotly visualisation (evaluate_utils.py:240-326) is out of scope")
        self._postprocess_raw_results_no_merge()
        accs = []
        for iou_thresh in iou_topk_dict["iou"]:
            for topk in iou_topk_dict["topk"]:
                accs.append(self.compute_IoU_recall_top_n_ours(topk, iou_thresh, temporal_nms)[2])
        return iou_topk_dict["topk"], accs


def recall_from_first_hits(first_hits, ious, topks):
    """accs in run_evaluate's order (for iou: for topk) from the per-query first-hit positions of ops.eval_recall:
    first_hits (n_queries, len(ious)) integer array."""
    fh = np.asarray(first_hits).reshape(-1, len(ious))
    total = max(fh.shape[0], 1)
    return [float((fh[:, q] < k).sum()) / total for q in range(len(ious)) for k in topks]


def results_entries(queries, gts, boxes):
    """results_entry for a whole batch with ONE device->host copy per field (all clips' detections / scores concatenated on
    the device first) instead of two per clip."""
    import torch
    dets = [b["detections"].detach().float() for b in boxes]
    scs = [b["scores"].detach().float() for b in boxes]
    if not dets or not dets[0].is_cuda:
        return [results_entry(q, g, b) for q, g, b in zip(queries, gts, boxes)]
    lens = [int(d.shape[0]) for d in dets]
    det = torch.cat(dets).cpu().numpy()
    sc = torch.cat(scs).cpu().numpy()
    out, o = [], 0
    for q, g, b, n in zip(queries, gts, boxes, lens):
        preds = np.concatenate([det[o:o + n], sc[o:o + n, None]], axis=1).tolist()
        out.append({"query": q, "gt": [float(g[0]), float(g[1])], "node_predictions": preds, "edge_predictions": preds,
                    "level": b["level"]})
        o += n
    return out


def results_entry(query, gt, box):
    """One main.py:324-348 record from a post-processor dict (drn_amd.model.inference) and its ground truth."""
    det = box["detections"].detach().float().cpu().numpy()
    sc = box["scores"].detach().float().cpu().numpy()
    preds = np.concatenate([det, sc[:, None]], axis=1).tolist()
    return {"query": query, "gt": [float(gt[0]), float(gt[1])], "node_predictions": preds, "edge_predictions": preds,
            "level": box["level"]}
