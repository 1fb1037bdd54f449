"""Query-gated temporal conv backbone (reference: model/backbone.py:4-36)."""
import types

import torch
import torch.nn as nn

from .. import functional as DF
from .basic_blocks import conv_bn


class Backbone(nn.Module):
    compute_dtype = torch.float32

    def __init__(self, channels_list, conv_block):
        super(Backbone, self).__init__()
        self.num_layers = len(channels_list)
        self.blocks = []
        for idx, (cin, cout, k, stride) in enumerate(channels_list):
            name = "forward_conv{}".format(idx)
            self.add_module(name, conv_block(cin, cout, kernel_size=k, stride=stride))
            self.blocks.append(name)

    def forward_from_stage(self, g0, gates, tail=None):
        """g0: (B, T, D+P) channels-last = cat(q0 * prop_fc(x), position feats) (drn_amd.functional.input_stage);
        gates[i]: (B, C_i) fp32.  The gate of level i+1 is fused into level i's BN-apply pass.  g0 may arrive in float32
        inside a bfloat16 model (feature dim not a 16-byte multiple in bf16): conv0 then runs on the exact-f32 kernels
        and its outputs are cast to the model's compute dtype.  tail: the EmbedTail input_stage returned with g0 (conv0's
        backward then produces the position-embedding gradients itself and skips those columns of its input gradient)."""
        outs, x = [], g0
        dt = self.compute_dtype
        for idx in range(self.num_layers):
            nxt = gates[idx + 1] if idx + 1 < self.num_layers else None
            conv, bn = conv_bn(getattr(self, self.blocks[idx]), "Backbone." + self.blocks[idx])
            if idx == 0 and x.shape[2] > conv.weight.shape[1]:
                # g0 carries zero-padded feature channels (mainModel, bf16 with a feature dim that is not a 16-byte multiple):
                # conv0's weight gets zero input channels at the same places -- [features | zeros | position embedding]
                W, extra = conv.weight, x.shape[2] - conv.weight.shape[1]
                P = tail.P if tail is not None else 256
                Wp = torch.cat([W[:, :W.shape[1] - P], W.new_zeros((W.shape[0], extra, W.shape[2])), W[:, W.shape[1] - P:]], dim=1)
                conv = types.SimpleNamespace(weight=Wp, bias=conv.bias, stride=conv.stride)
            out, gated = DF.conv_block([x], conv, bn, self.training, x.dtype if idx == 0 else dt, gate=nxt,
                                       tail=tail if idx == 0 else None)
            outs.append(DF.cast_act(out[0], dt))
            x = DF.cast_act(gated, dt) if gated is not None else None
        return outs

    def forward(self, x, query_fts, position_fts):
        """Reference signature (model/backbone.py:17): x (B, C, T), query_fts[i] (B, C_i), position_fts[0] (B, P, T).
        The level-0 gate + concat is plain tensor glue here; mainModel uses the fused input stage instead."""
        dt = self.compute_dtype
        g0 = torch.cat([DF.as_nlc(x, dt) * query_fts[0].to(dt)[:, None, :], DF.as_nlc(position_fts[0], dt)], dim=2)
        outs = self.forward_from_stage(g0, [q.float() for q in query_fts])
        DF.flush_bn_counters()
        return tuple(o.permute(0, 2, 1) for o in outs)
