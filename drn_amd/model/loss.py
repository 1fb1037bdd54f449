"""FCOS-style losses of DRN (reference: model/loss.py:22-262, model/layers/iou_loss.py, sigmoid_focal_loss.py)."""
import torch

from .. import functional as DF

INF = 100000000
TARGET_SCALE = 32.0          # model/loss.py:98,178 (the reference hard-codes 32 proposals)


class FCOSLossComputation(object):
    """Same call signature as the reference (model/loss.py:134).  Target assignment, focal loss, IoU loss and
    the stage-2/3 IoU-score loss run in one fused HIP kernel (forward) + one (backward); positive counts stay
    on the device, so unlike the reference there is no host sync (`nonzero`, `.item()`, model/loss.py:194,209).

    Deviations that do not change values: with no tIoU>0.9 positives the reference returns
    an int64 CPU `tensor([0])`; here `loss_iou` is a float32 device zero with zero gradient."""

    object_sizes_of_interest = [[-1, 6], [5.6, 11], [11, INF]]           # model/loss.py:47-51

    def __init__(self, cfg):
        self.gamma = cfg["fcos_loss_gamma"]
        self.alpha = cfg["fcos_loss_alpha"]
        self.fpn_strides = None
        self.total_points = []
        self.last_counts = None

    @staticmethod
    def _flat(lst):
        flat = getattr(lst, "flat", None)
        if flat is not None:
            return flat
        return torch.cat([t.permute(0, 2, 1).reshape(-1, t.size(1)) for t in lst], dim=0)

    def __call__(self, locations, box_cls, box_regression, targets, iou_scores, is_first_stage=True):
        B = box_cls[0].size(0)
        if B == 1 and not is_first_stage:
            # model/loss.py:186,192: `squeeze()` drops the batch dimension of the IoU scores when there is one clip, and indexing the
            # (sumL,) result with the (1, sumL) mask raises -- in train AND eval mode (model/fcos.py:176 calls the evaluator there
            # too), with or without a tIoU > 0.9 positive.  Same exception, same text (tests/golden/errors.json, recorded from the
            # reference): the kernels below would compute the "intended" value, which the reference never produced.
            raise IndexError("too many indices for tensor of dimension 1")
        if self.fpn_strides is not None:
            strides = [float(s) for s in self.fpn_strides[:len(locations)]]
        else:                                                            # generic callers: read them back (host sync)
            strides = [float(2 * l[0]) for l in locations]
        levels = [(int(c.size(2)), strides[i], float(self.object_sizes_of_interest[i][0]),
                   float(self.object_sizes_of_interest[i][1])) for i, c in enumerate(box_cls)]
        logits, reg = self._flat(box_cls), self._flat(box_regression)
        iou = None if is_first_stage else self._flat(iou_scores)
        extra = None
        if logits.shape[1] > 1:
            # more than one foreground channel (fcos_num_class > 2): DRN's targets carry class 1 only (model/loss.py:103-131), so
            # channel 0 is what the fused kernel computes and every other channel is background at every location -- their focal
            # terms come from the class-general kernel (drn_focal_fwd / _bwd) with all-zero labels and join the same
            # normalisation, sum / (n_pos + N) (model/loss.py:209-213)
            from .layers.sigmoid_focal_loss import sigmoid_focal_loss
            zeros = torch.zeros(logits.shape[0], dtype=torch.int32, device=logits.device)
            extra = sigmoid_focal_loss(logits[:, 1:].contiguous(), zeros, self.gamma, self.alpha).sum()
            logits = logits[:, :1].contiguous()
        l_cls, l_reg, l_iou, counts, total = DF.fcos_loss(logits, reg, iou, targets, levels, B, self.gamma, self.alpha, TARGET_SCALE,
                                      not is_first_stage)
        if extra is not None:
            extra = (extra / (counts[0].detach() + float(B))).reshape(1)
            l_cls = l_cls + extra
            total = total + extra
        self.last_counts = counts
        self.last_total = total
        if is_first_stage:
            return l_cls, l_reg, l_iou.detach()    # stage 1: the kernel writes 0 there (loss.py:239, `torch.tensor([0])`)
        return l_cls, l_reg, l_iou


def make_fcos_loss_evaluator(cfg):
    return FCOSLossComputation(cfg)
