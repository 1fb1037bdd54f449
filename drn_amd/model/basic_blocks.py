"""conv -> BatchNorm1d -> ReLU blocks (reference: model/basic_blocks.py:5-33)."""
import torch
from torch import nn

from .. import functional as DF


class ConvBlock(nn.Sequential):
    """nn.Sequential(Conv1d(bias=False), BatchNorm1d, ReLU) as a parameter holder (keys `0.weight`, `1.*`);
    forward runs the fused implicit-GEMM + BN + ReLU HIP path on channels-last activations."""

    compute_dtype = torch.float32

    def forward_nlc(self, xs, gate=None, up=None):
        """xs: list of (B, L, Cin) channels-last level inputs -> (list of outputs, gated output or None)."""
        return DF.conv_block(xs, self[0], self[1], self.training, self.compute_dtype, gate=gate, up=up)

    def forward(self, x):
        out, _ = self.forward_nlc([DF.as_nlc(x, self.compute_dtype)])
        DF.flush_bn_counters()
        return out[0].permute(0, 2, 1)


def conv_with_kaiming_uniform(use_bn=True, use_relu=True, use_dropout=False):
    """Same factory signature as the reference (model/basic_blocks.py:5).  Only the combination the DRN
    model instantiates (BN + ReLU, no dropout; model/main_model.py:28) has a HIP path."""
    if not (use_bn and use_relu) or use_dropout:
        raise NotImplementedError("drn_amd implements conv+BN+ReLU blocks only (the combination DRN uses)")

    def make_conv(in_channels, out_channels, kernel_size=3, stride=1, dilation=1):
        if dilation != 1:
            raise NotImplementedError("dilation != 1 is not used by DRN")
        conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                         padding=(kernel_size - 1) // 2, bias=False)
        nn.init.kaiming_uniform_(conv.weight, a=1)
        return ConvBlock(conv, nn.BatchNorm1d(out_channels), nn.ReLU(inplace=True))

    return make_conv
