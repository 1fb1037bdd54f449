"""Class-general sigmoid focal loss (reference: model/layers/sigmoid_focal_loss.py:9-69).

The reference calls the third-party pybind extension `fcos_core._C.sigmoid_focalloss_forward/backward`; here the same two
functions are the C-ABI entry points `drn_focal_fwd` / `drn_focal_bwd` (include/drn_hip.h), same argument meaning:
logits (N, C) fp32, int32 targets (N,) with 0 = background and c in 1..C = class c, gamma, alpha -> per-element losses.
DRN's own training step uses the fused whole-loss kernel instead (drn_amd/model/loss.py); this module is the 1:1
replacement of the reference layer for callers that use it on its own."""
import torch
from torch import nn
from torch.autograd.function import once_differentiable

from ... import ops


class _SigmoidFocalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        logits = logits.contiguous().float()
        targets = targets.contiguous().to(torch.int32)
        ctx.save_for_backward(logits, targets)
        ctx.gamma, ctx.alpha = float(gamma), float(alpha)
        return ops.focal_fwd(logits, targets, ctx.gamma, ctx.alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        logits, targets = ctx.saved_tensors
        return ops.focal_bwd(logits, targets, d_loss.contiguous().float(), ctx.gamma, ctx.alpha), None, None, None


sigmoid_focal_loss = _SigmoidFocalLoss.apply


class SigmoidFocalLoss(nn.Module):
    def __init__(self, gamma, alpha):
        super(SigmoidFocalLoss, self).__init__()
        self.gamma = gamma
        self.alpha = alpha

    def forward(self, logits, targets):
        return sigmoid_focal_loss(logits, targets, self.gamma, self.alpha).sum()      # sigmoid_focal_loss.py:68

    def __repr__(self):
        return "%s(gamma=%s, alpha=%s)" % (self.__class__.__name__, self.gamma, self.alpha)
