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
        """feats: channels-last (B, L_l, C_l), highest resolution first.  The nearest-x2 upsample + add of the
        top-down path (FPN.py:63-68) is fused into the lateral block's BN-apply pass."""
        last, _ = getattr(self, self.inner_blocks[-1]).forward_nlc([feats[-1]])
        last = last[0]
        results = [getattr(self, self.layer_blocks[-1]).forward_nlc([last])[0][0]]
        for feat, inner, layer in zip(feats[:-1][::-1], self.inner_blocks[:-1][::-1], self.layer_blocks[:-1][::-1]):
            last = getattr(self, inner).forward_nlc([feat], up=last)[0][0]
            results.insert(0, getattr(self, layer).forward_nlc([last])[0][0])
        return results

    def forward(self, x):
        outs = self.forward_nlc([DF.as_nlc(f, self.compute_dtype) for f in x])
        DF.flush_bn_counters()
        return tuple(o.permute(0, 2, 1) for o in outs)
