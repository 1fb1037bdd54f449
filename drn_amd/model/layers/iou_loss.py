"""Stand-alone IoU regression loss (reference: model/layers/iou_loss.py:5-24, exported from model/layers/__init__.py).

DRN's own training step computes the same expression inside the fused whole-loss kernel (drn_amd/model/loss.py, positives
only, plain mean); this module is the 1:1 replacement of the reference layer for callers that use it on its own, the
`weight` branch included: pred / target (N, 2) = (left, right) distances ->
    weight given and weight.sum() > 0:  (losses * weight).sum() / weight.sum()      else: losses.mean()
on the HIP kernels `drn_iou_loss_fwd` / `drn_iou_loss_bwd` (include/drn_hip.h).  The weighted / unweighted decision is taken
on the device (no host sync).  Gradients flow to `pred` and `target`; `weight` is a constant (in FCOS it is a target-derived
centerness map)."""
import torch
from torch import nn
from torch.autograd.function import once_differentiable

from ... import ops
from ..._lib import DrnError


class _IOULoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight):
        pred, target = pred.contiguous().float(), target.contiguous().float()
        if pred.numel() == 0:
            raise DrnError("IOULoss: no rows (the reference asserts losses.numel() != 0, model/layers/iou_loss.py:23)")
        w = weight.contiguous().float().reshape(-1) if weight is not None else None
        out2 = ops.iou_loss_fwd(pred, target, w)
        ctx.save_for_backward(pred, target, out2, *( [w] if w is not None else []))
        return out2[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        pred, target, out2 = ctx.saved_tensors[:3]
        w = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        dpred, dtarget = ops.iou_loss_bwd(pred, target, w, out2, g.contiguous().float().reshape(1),
                                          want_pred=ctx.needs_input_grad[0], want_target=ctx.needs_input_grad[1])
        return dpred, dtarget, None


class IOULoss(nn.Module):
    def forward(self, pred, target, weight=None):
        return _IOULoss.apply(pred, target, weight)
