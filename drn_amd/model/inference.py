"""Eval-time post-processing (reference: model/inference.py:11-237): sigmoid, 0.05 threshold before the
IoU-score product, per-level top-k, segment decoding / 32, clamp to [0, 1], score = sqrt(cls[*iou]), merge levels.
On the HIP path (head outputs in their flat loss layout) all of it is ONE kernel, drn_postprocess, and one
device->host copy of the per-(clip, level) counts; the reference's per-clip host loop remains as the generic path for
callers that hand over plain per-level tensors."""
import torch

from .. import ops


class DeviceDetections(object):
    """What drn_postprocess leaves on the device for a whole batch, un-sliced: det (B, R, 2), scores (B, R), locs (B, R),
    counts (B, levels) -- clip b's detections are det[b, :counts[b].sum()], level after level.  Returned instead of the
    per-clip dicts when FCOSPostProcessor.device_only is set (drn_amd.trainer.Trainer.evaluate(with_results=False)): the
    per-batch host copy of the counts -- the eval path's one synchronisation -- and ~100 tensor slices per batch go away, and
    drn_amd.metrics.device_first_hits computes Recall@k from these buffers where they are."""

    def __init__(self, det, scores, locs, counts):
        self.det, self.scores, self.locs, self.counts = det, scores, locs, counts

    def __len__(self):
        return int(self.det.shape[0])


class FCOSPostProcessor(torch.nn.Module):
    def __init__(self, pre_nms_thresh, pre_nms_top_n, nms_thresh, fpn_post_nms_top_n, min_size, num_classes,
                 is_first_stage, is_second_stage):
        super(FCOSPostProcessor, self).__init__()
        self.pre_nms_thresh = pre_nms_thresh
        self.pre_nms_top_n = pre_nms_top_n
        self.nms_thresh = nms_thresh
        self.fpn_post_nms_top_n = fpn_post_nms_top_n
        self.min_size = min_size
        self.num_classes = num_classes
        self.downsample_scale = 32
        self.is_first_stage = is_first_stage
        self.is_second_stage = is_second_stage
        self.strides = None          # set by FCOSModule (fpn_stride); otherwise recovered from the locations
        self.device_only = False     # HIP path: hand back DeviceDetections (no host copy, no per-clip dicts)

    def forward_for_single_feature_map(self, locations, box_cls, box_regression, level, iou_scores):
        N = box_cls.shape[0]
        cls = box_cls.float().permute(0, 2, 1).sigmoid()                 # (N, L, C)
        iou = iou_scores.float().permute(0, 2, 1).sigmoid()
        reg = box_regression.float().permute(0, 2, 1)
        cand = cls > self.pre_nms_thresh                                 # before the iou product (inference.py:71-79)
        top_n = cand.reshape(N, -1).sum(1).clamp(max=self.pre_nms_top_n).tolist()
        if not self.is_first_stage:
            cls = cls * iou
        results = []
        for i in range(N):
            idx = cand[i].nonzero()
            scores = cls[i][cand[i]]
            where = idx[:, 0]
            breg, bloc, labels = reg[i][where], locations[where], idx[:, 1] + 1
            if scores.numel() > top_n[i]:
                scores, keep = scores.topk(top_n[i], sorted=False)
                breg, bloc, labels = breg[keep], bloc[keep], labels[keep]
            det = torch.stack([bloc - breg[:, 0], bloc + breg[:, 1]], dim=1) / self.downsample_scale
            det = det.clamp(min=0, max=1)
            det = det[(det[:, 1] - det[:, 0]) >= self.min_size]
            results.append({"detections": det, "labels": labels, "scores": torch.sqrt(scores), "level": [level],
                            "locations": bloc / 32})
        return results

    def forward_flat(self, locations, box_cls, box_regression, iou_scores):
        """HIP path: box_cls / box_regression / iou_scores are LevelLists sharing flat (R, n) fp32 buffers."""
        B = box_cls[0].shape[0]
        nl = len(box_cls)
        strides = [float(2 * loc[0]) for loc in locations] if self.strides is None else self.strides[:nl]
        levels = ops.loss_levels([(int(c.shape[2]), float(strides[i]), 0.0, 0.0) for i, c in enumerate(box_cls)])
        iou = None if self.is_first_stage else iou_scores.flat
        det, scores, locs, counts = ops.postprocess(levels, B, box_cls.flat, box_regression.flat, iou, self.pre_nms_thresh,
                                                    self.pre_nms_top_n, float(self.downsample_scale))
        if self.device_only:
            return DeviceDetections(det, scores, locs, counts)
        counts = counts.tolist()                                           # the only host sync of the eval path
        results = []
        for b in range(B):
            n = sum(counts[b])
            if n == 0:                                                     # inference.py:192-197
                dev = det.device
                results.append({"detections": torch.tensor([[0.0, 1.0]], device=dev), "labels": [],
                                "scores": torch.tensor([1.0], device=dev), "level": [[-1]],
                                "locations": torch.tensor([0.5], device=dev)})
            else:
                results.append({"detections": det[b, :n], "labels": [], "scores": scores[b, :n],
                                "level": [[l] * c for l, c in enumerate(counts[b])], "locations": locs[b, :n]})
        return results

    def forward(self, locations, box_cls, box_regression, iou_scores):
        # (the one-kernel path is for DRN's single foreground channel; more channels -- fcos_num_class > 2 -- take the reference's
        # per-level procedure below, candidates over (location, class), on the device with torch ops)
        if all(getattr(x, "flat", None) is not None and x.flat.is_cuda for x in (box_cls, box_regression)) and \
                box_cls.flat.shape[1] == 1 and \
                (self.is_first_stage or getattr(iou_scores, "flat", None) is not None) and self.min_size == 0:
            return self.forward_flat(locations, box_cls, box_regression, iou_scores)
        sampled = [self.forward_for_single_feature_map(l, o, b, i, s)
                   for i, (l, o, b, s) in enumerate(zip(locations, box_cls, box_regression, iou_scores))]
        return self.select_over_all_levels(list(zip(*sampled)))

    def select_over_all_levels(self, boxlists):
        results = []
        for dicts in boxlists:
            dev = dicts[0]["detections"].device
            dets = [d["detections"] for d in dicts if len(d["detections"]) != 0]
            if len(dets) == 0:                                           # inference.py:192-197
                res = {"detections": torch.tensor([[0.0, 1.0]], device=dev), "labels": [],
                       "scores": torch.tensor([1.0], device=dev), "level": [[-1]],
                       "locations": torch.tensor([0.5], device=dev)}
            else:
                res = {"detections": torch.cat(dets, dim=0), "labels": [],
                       "scores": torch.cat([d["scores"] for d in dicts if len(d["scores"]) != 0], dim=0),
                       "level": [d["level"] * len(d["detections"]) for d in dicts if len(d["level"]) != 0],
                       "locations": torch.cat([d["locations"] for d in dicts if len(d["locations"]) != 0], dim=0)}
            results.append(res)
        return results


def make_fcos_postprocessor(config):
    return FCOSPostProcessor(pre_nms_thresh=config["fcos_inference_thr"], pre_nms_top_n=config["fcos_pre_nms_top_n"],
                             nms_thresh=config["fcos_nms_thr"], fpn_post_nms_top_n=config["test_detections_per_img"],
                             min_size=0, num_classes=config["fcos_num_class"], is_first_stage=config['is_first_stage'],
                             is_second_stage=config['is_second_stage'])
