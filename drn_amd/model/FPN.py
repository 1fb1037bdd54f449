"""Temporal feature pyramid (reference: model/FPN.py:7-103, top blocks included)."""
import torch
from torch import nn

from .. import functional as DF
from .basic_blocks import conv_bn


class FPN(nn.Module):
    compute_dtype = torch.float32

    def __init__(self, in_channels_list, out_channels, conv_block, top_blocks=None):
        super(FPN, self).__init__()
        self.inner_blocks, self.layer_blocks = [], []
        for idx, in_channels in enumerate(in_channels_list, 1):
            inner, layer = "fpn_inner{}".format(idx), "fpn_layer{}".format(idx)
            if in_channels == 0:
                continue
            self.add_module(inner, conv_block(in_channels, out_channels, 1))
            self.add_module(layer, conv_block(out_channels, out_channels, 3, 1))
            self.inner_blocks.append(inner)
            self.layer_blocks.append(layer)
        self.top_blocks = top_blocks                # None for DRN (model/main_model.py:30); LastLevelMaxPool / LastLevelP6P7

    def forward_nlc(self, feats):
        """feats: channels-last (B, L_l, C_l), highest resolution first.  The three lateral 1x1 convs run as ONE grouped
        implicit-GEMM launch, their BN-apply passes resolve the top-down chain last_l = lateral_l + nearest_x2(last_{l+1})
        coarse to fine (FPN.py:54-68), then the three output convs run as one grouped launch (FPN.py:56,69)."""
        inner = [conv_bn(getattr(self, nm), "FPN." + nm) for nm in self.inner_blocks]
        layer = [conv_bn(getattr(self, nm), "FPN." + nm) for nm in self.layer_blocks]
        dt = self.compute_dtype
        last = DF.multi_conv_block(list(feats), inner, self.training, dt, chain_up=True)
        results = DF.multi_conv_block(last, layer, self.training, dt, chain_up=False)
        if isinstance(self.top_blocks, LastLevelP6P7):          # model/FPN.py:72-74
            results = list(results) + self.top_blocks.forward_nlc(feats[-1], results[-1], dt)
        elif isinstance(self.top_blocks, LastLevelMaxPool):     # model/FPN.py:75-77
            results = list(results) + self.top_blocks.forward_nlc(results[-1])
        return results

    def forward(self, x):
        outs = self.forward_nlc([DF.as_nlc(f, self.compute_dtype) for f in x])
        DF.flush_bn_counters()
        return tuple(o.permute(0, 2, 1) for o in outs)


class LastLevelMaxPool(nn.Module):
    """model/FPN.py:81-83 applies max_pool2d(x, kernel 1, stride 2) to a 3-D (B, C, L) tensor, which F.max_pool2d reads as an
    unbatched (C, H, W) image: a kernel-1 pool is a subsample, here of BOTH trailing dims -> (B, ceil(C/2), ceil(L/2)).
    Reproduced as the same strided view (no kernel)."""

    def forward_nlc(self, x):
        return [x[:, ::2, ::2]]

    def forward(self, x):
        return [x[:, ::2, ::2]]


class LastLevelP6P7(nn.Module):
    """model/FPN.py:86-103: P6 = conv(k3, s2)(c5 or p5), P7 = conv(k3, s2)(relu(P6)); same parameter names (p6, p7)."""

    def __init__(self, in_channels, out_channels):
        super(LastLevelP6P7, self).__init__()
        self.p6 = nn.Conv1d(in_channels, out_channels, 3, 2, 1)
        self.p7 = nn.Conv1d(out_channels, out_channels, 3, 2, 1)
        for module in [self.p6, self.p7]:
            nn.init.kaiming_uniform_(module.weight, a=1)
            nn.init.constant_(module.bias, 0)
        self.use_P5 = in_channels == out_channels

    def forward_nlc(self, c5, p5, dtype):
        x = p5 if self.use_P5 else c5
        p6 = DF.plain_conv(x, self.p6, dtype)
        return [p6, DF.plain_conv(torch.relu(p6), self.p7, dtype)]

    def forward(self, c5, p5):
        dt = torch.float32
        return [o.permute(0, 2, 1) for o in self.forward_nlc(DF.as_nlc(c5, dt), DF.as_nlc(p5, dt), dt)]
