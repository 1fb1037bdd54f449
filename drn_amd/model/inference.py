"""Eval-time post-processing (reference: model/inference.py:11-237): sigmoid, 0.05 threshold before the
IoU-score product, per-level top-k, segment decoding / 32, clamp to [0, 1], score = sqrt(cls[*iou]), merge levels.
Host-side control flow over a few hundred tiny device values per clip, as in the reference (eval only)."""
import torch


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

    def forward(self, locations, box_cls, box_regression, iou_scores):
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
