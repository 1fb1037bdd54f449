"""Temporal feature pyramid (reference: model/FPN.py:7-78, top_blocks=None)."""
import torch
from torch import nn

from .. import functional as DF


class FPN(nn.Module):
    compute_dtype = torch.float32

    def __init__(self, in_channels_list, out_channels, conv_block, top_blocks=None):
        super(FPN, self).__init__()
        if top_blocks is not None:
            raise NotImplementedError("top_blocks is unused by DRN (model/main_model.py:30)")
        self.inner_blocks, self.layer_blocks = [], []
        for idx, in_channels in enumerate(in_channels_list, 1):
            inner, layer = "fpn_inner{}".format(idx), "fpn_layer{}".format(idx)
            if in_channels == 0:
                continue
            self.add_module(inner, conv_block(in_channels, out_channels, 1))
            self.add_module(layer, conv_block(out_channels, out_channels, 3, 1))
            self.inner_blocks.append(inner)
            self.layer_blocks.append(layer)
        self.top_blocks = None

    def forward_nlc(self, feats):
        """feats: channels-last (B, L_l, C_l), highest resolution first.  The three lateral 1x1 convs run as ONE grouped
        implicit-GEMM launch, their BN-apply passes resolve the top-down chain last_l = lateral_l + nearest_x2(last_{l+1})
        coarse to fine (FPN.py:54-68), then the three output convs run as one grouped launch (FPN.py:56,69)."""
        inner = [(getattr(self, nm)[0], getattr(self, nm)[1]) for nm in self.inner_blocks]
        layer = [(getattr(self, nm)[0], getattr(self, nm)[1]) for nm in self.layer_blocks]
        dt = self.compute_dtype
        last = DF.multi_conv_block(list(feats), inner, self.training, dt, chain_up=True)
        return DF.multi_conv_block(last, layer, self.training, dt, chain_up=False)

    def forward(self, x):
        outs = self.forward_nlc([DF.as_nlc(f, self.compute_dtype) for f in x])
        DF.flush_bn_counters()
        return tuple(o.permute(0, 2, 1) for o in outs)
