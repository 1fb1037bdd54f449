"""conv [-> BatchNorm1d] [-> ReLU] [-> Dropout] blocks (reference: model/basic_blocks.py:5-33)."""
import torch
from torch import nn

from .. import functional as DF
from .._lib import DrnError


def conv_bn(block, who):
    """(conv, BatchNorm1d) of a conv -> BN -> ReLU block, the only kind the channels-last model paths run (DRN builds nothing
    else, model/main_model.py:28): the no-BN / no-ReLU / Dropout variants of the factory raise here instead of being
    mis-indexed or silently run without their Dropout."""
    if not isinstance(block, nn.Sequential) or len(block) < 2 or not isinstance(block[0], nn.Conv1d) \
            or not isinstance(block[1], nn.BatchNorm1d):
        raise DrnError("%s: needs conv -> BatchNorm1d -> ReLU blocks (conv_with_kaiming_uniform(True, True)); got %s"
                       % (who, type(block).__name__))
    if not getattr(block, "relu", True) or not any(isinstance(m, nn.ReLU) for m in block):
        raise DrnError("%s: a block without ReLU is only served by its own forward()" % who)
    if block.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in block):
        raise DrnError("%s: Dropout is not applied on the channels-last model path; call the block itself" % who)
    return block[0], block[1]


class ConvBlock(nn.Sequential):
    """nn.Sequential(Conv1d(bias=False), BatchNorm1d[, ReLU]) as a parameter holder (keys `0.weight`, `1.*`);
    forward runs the fused implicit-GEMM + BN (+ ReLU) HIP path on channels-last activations."""

    compute_dtype = torch.float32
    relu = True

    def forward_nlc(self, xs, gate=None, up=None):
        """xs: list of (B, L, Cin) channels-last level inputs -> (list of outputs, gated output or None).  The channels-last
        path has no Dropout stage (DRN builds its blocks without one, model/main_model.py:28): a block made with
        use_dropout=True is served by `forward` only and refuses to run here in training mode rather than skip it silently."""
        if self.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in self):
            raise DrnError("ConvBlock.forward_nlc: Dropout is not applied on the channels-last model path; call the block itself")
        return DF.conv_block(xs, self[0], self[1], self.training, self.compute_dtype, gate=gate, up=up, relu=self.relu)

    def forward(self, x):
        out, _ = self.forward_nlc([DF.as_nlc(x, self.compute_dtype)])
        DF.flush_bn_counters()
        y = out[0].permute(0, 2, 1)
        for m in list(self)[2:]:                              # Dropout, when the factory was asked for it
            if isinstance(m, nn.Dropout):
                y = m(y)
        return y


class PlainConvBlock(nn.Sequential):
    """nn.Sequential(Conv1d(bias=True)[, ReLU][, Dropout]) -- the use_bn=False variants of the reference factory (same keys);
    forward runs the implicit-GEMM HIP path with the bias in the epilogue."""

    compute_dtype = torch.float32

    def forward_nlc(self, x):
        relu = any(isinstance(m, nn.ReLU) for m in self)
        return DF.plain_conv(x, self[0], self.compute_dtype, relu=relu)

    def forward(self, x):
        y = self.forward_nlc(DF.as_nlc(x, self.compute_dtype)).permute(0, 2, 1)
        for m in self:
            if isinstance(m, nn.Dropout):
                y = m(y)
        return y


class PlainConv(nn.Conv1d):
    """A bare nn.Conv1d (what the reference factory returns for use_bn=use_relu=use_dropout=False; keys `weight`, `bias`) whose
    forward runs the implicit-GEMM HIP path."""

    compute_dtype = torch.float32

    def forward_nlc(self, x, relu=False):
        return DF.plain_conv(x, self, self.compute_dtype, relu=relu)

    def forward(self, x):
        return self.forward_nlc(DF.as_nlc(x, self.compute_dtype)).permute(0, 2, 1)


def conv_with_kaiming_uniform(use_bn=True, use_relu=True, use_dropout=False):
    """Same factory signature and module layout (state_dict keys) as the reference (model/basic_blocks.py:5-33): every
    combination of BatchNorm / ReLU / Dropout.  DRN itself instantiates BN + ReLU only (model/main_model.py:28), and the
    model paths (Backbone / FPN / FCOSHead `forward_nlc`) accept exactly that kind of block: they index it as (conv, BatchNorm)
    and raise DrnError for the use_bn=False variants, which -- like Dropout -- are served by the blocks' own `forward`."""

    def make_conv(in_channels, out_channels, kernel_size=3, stride=1, dilation=1):
        if dilation != 1:
            raise DrnError("dilation != 1 has no HIP path (no caller in the reference passes one: model/backbone.py:12, "
                           "model/FPN.py:36-37)")
        pad = (kernel_size - 1) // 2
        if use_bn:
            conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=pad, bias=False)
            nn.init.kaiming_uniform_(conv.weight, a=1)
            mods = [conv, nn.BatchNorm1d(out_channels)]
            if use_relu:
                mods.append(nn.ReLU(inplace=True))
            if use_dropout:
                mods.append(nn.Dropout(p=0.5))
            blk = ConvBlock(*mods)
            blk.relu = use_relu
            return blk
        if use_relu or use_dropout:
            conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=pad, bias=True)
            nn.init.kaiming_uniform_(conv.weight, a=1)
            mods = [conv] + ([nn.ReLU(inplace=True)] if use_relu else []) + ([nn.Dropout(p=0.5)] if use_dropout else [])
            return PlainConvBlock(*mods)
        conv = PlainConv(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=pad, bias=True)
        nn.init.kaiming_uniform_(conv.weight, a=1)
        return conv

    return make_conv
