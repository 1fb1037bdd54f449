"""DRN top-level model (reference: model/main_model.py:13-81): same constructor/forward signature, attribute
names and state_dict keys, so main.py-style trainers and reference checkpoints work unchanged."""
import types

import torch
import torch.nn.functional as F
import torch.nn as nn

from .. import functional as DF
from .. import ops
from .._lib import DrnError
from .backbone import Backbone
from .basic_blocks import conv_with_kaiming_uniform
from .fcos import FCOSHead, build_fcos
from .FPN import FPN
from .language_module import QueryEncoder


class mainModel(nn.Module):
    def __init__(self, vocab_size, dataset_configs, hidden_dim=512, embed_dim=300, bidirection=True,
                 graph_node_features=1024, compute_dtype=torch.float32):
        super(mainModel, self).__init__()
        dataset_configs = vars(dataset_configs) if not isinstance(dataset_configs, dict) else dataset_configs
        self.first_output_dim = dataset_configs["first_output_dim"]
        self.fpn_feature_dim = dataset_configs["fpn_feature_dim"]
        self.feature_dim = dataset_configs[dataset_configs['feature_type']]['feature_dim']
        self.query_encoder = QueryEncoder(vocab_size, hidden_dim, embed_dim, dataset_configs["lstm_layers"], bidirection)
        channels_list = [
            (self.feature_dim + 256, self.first_output_dim, 3, 1),
            (self.first_output_dim, self.first_output_dim * 2, 3, 2),
            ((self.first_output_dim * 2), self.first_output_dim * 4, 3, 2),
        ]
        conv_func = conv_with_kaiming_uniform(use_bn=True, use_relu=True)
        self.backbone_net = Backbone(channels_list, conv_func)
        self.fpn = FPN([256, 512, 1024], 512, conv_func)
        self.fcos = build_fcos(dataset_configs, self.fpn_feature_dim)
        self.prop_fc = nn.Linear(self.feature_dim, self.feature_dim)
        self.position_transform = nn.Linear(3, 256)
        for t in range(len(channels_list)):
            setattr(self, "qInput%d" % t, nn.Linear(1024, self.feature_dim if t == 0 else channels_list[t - 1][1]))
        self.set_compute_dtype(compute_dtype)
        self.taps = None      # set to a dict to record intermediate activations (tests / debugging)
        # the re-laid GEMM copies of this model's weights live in a store the model owns (not in process-wide tables): they
        # die with the model, and an optimizer only looks at / refreshes the copies of the model it trains
        self.weight_copies = DF.WeightCopies().adopt(self)

    def set_compute_dtype(self, dtype):
        """torch.float32: exact-f32 MFMA (<=1e-4 parity with the reference); torch.bfloat16: bf16 storage + fp32 accumulate."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise DrnError("compute dtype must be float32 or bfloat16")
        for m in self.modules():
            if hasattr(type(m), "compute_dtype"):
                m.compute_dtype = dtype
        self.compute_dtype = dtype
        return self

    def encode_commands(self, query_tokens, query_length):
        """The query encoder alone: its three (B, 2H) commands (model/language_module.py:38-63), one autograd node."""
        return list(DF.query_encoder(query_tokens, query_length, self.query_encoder, None, lowp=self.compute_dtype == torch.bfloat16))

    def project_gates(self, cmds):
        """The per-level gate projections on given commands (model/main_model.py:47-50), one autograd node (DF.gate_projections):
        encode_query == project_gates(encode_commands(...)), value for value, with the commands as a cut point for schedules that
        exchange the projections' gradients early (drn_amd.graph.TwoPhaseStep)."""
        return DF.gate_projections(cmds, [getattr(self, "qInput%d" % i) for i in range(3)], lowp=self.compute_dtype == torch.bfloat16)

    def gate_parameters(self):
        return [p for t in range(len(self.backbone_net.blocks)) for p in getattr(self, "qInput%d" % t).parameters()]

    def encoder_parameters(self):
        return list(self.query_encoder.parameters())

    def encode_query(self, query_tokens, query_length):
        """Query encoder + per-level gate projections (model/main_model.py:47-50): three (B, C_l) fp32 gate tensors."""
        n = len(self.backbone_net.blocks)
        if n == 3:          # the whole query side, gate projections included, as one autograd node
            return list(DF.query_encoder(query_tokens, query_length, self.query_encoder,
                                         [getattr(self, "qInput%d" % i) for i in range(n)],
                                         lowp=self.compute_dtype == torch.bfloat16))
        query_features = self.query_encoder(query_tokens, query_length)
        return [DF.linear(query_features[i], getattr(self, "qInput%d" % i)) for i in range(n)]

    # The step splits at the tensors that leave the "front" (query encoder, gate projections, prop_fc, position
    # embedding): g0 and the gates of levels 1.. .  Backward runs the trunk (backbone, FPN, heads, losses) first, so its
    # gradients can be all-reduced while the front's backward -- which holds the largest gradient, prop_fc.weight -- is
    # still running (drn_amd.graph.TwoPhaseStep).
    def query_parameters(self):
        """Query encoder + the per-level gate projections (main_model.py:36-40,47-50)."""
        mods = [self.query_encoder] + [getattr(self, "qInput%d" % t) for t in range(len(self.backbone_net.blocks))]
        return [p for m in mods for p in m.parameters()]

    def unused_parameters(self):
        """Parameters the reference constructs but never calls: QueryEncoder.textualAttention (language_module.py:16,65-74;
        6.3 M of them).  Their .grad stays None, so in the reference clip_grad_norm_ and Adam.step never touch them
        (main.py:140,238-243: both skip parameters without a gradient).  `learned_parameters` leaves them out of the flat
        gradient buckets -- same semantics, but no zero gradients to all-reduce and no Adam pass over zero moments."""
        return list(self.query_encoder.textualAttention.parameters())

    def learned_parameters(self):
        """What an optimizer should hold: every parameter with requires_grad that the forward pass can reach."""
        dead = set(id(p) for p in self.unused_parameters())
        return [p for p in self.parameters() if p.requires_grad and id(p) not in dead]

    def input_parameters(self):
        """prop_fc + position_transform: the input stage (main_model.py:33-34,51-59)."""
        return list(self.prop_fc.parameters()) + list(self.position_transform.parameters())

    def front_parameters(self):
        return self.query_parameters() + self.input_parameters()

    def grad_stack_groups(self):
        """For GradReducer(adjacent=...): parameters whose gradients are computed as one stacked tensor."""
        return self.fcos.head.grad_stack_groups()

    def trunk_parameters(self):
        front = set(id(p) for p in self.front_parameters())
        return [p for p in self.parameters() if id(p) not in front]

    def prepare_input(self, props_features, props_start_end):
        """The part of the input stage that does not depend on the query (cast / transposed copy of the features, position
        features, the warmed GEMM copy of the prop_fc weight): drn_amd.graph.DualStreamStep runs it beside the query encoder."""
        if not props_features.is_cuda:
            raise DrnError("drn_amd.mainModel runs on an MI355X only (inputs on %s); no CPU fallback" % props_features.device)
        dt = self.compute_dtype
        want_wgrad = self.prop_fc.weight.requires_grad and torch.is_grad_enabled()
        pad = self._front_pad(props_features.shape[2])
        if pad:
            # bf16 rows must be 16-byte multiples for the MFMA kernels' LDS staging.  A feature dim that is not (D = 500, the
            # ActivityNet C3D-PCA convention, BASELINE configs[4]) runs the front -- prop_fc, gating, conv0 -- on a ZERO-PADDED
            # width Dp = 512: features, prop_fc's weight / bias, the level-0 gate and conv0's input channels get zero columns, so
            # every padded activation is exactly 0 and the real ones are what they were; torch's pad / cat nodes hand the real
            # slices of the gradients back to the parameters.  (Round 3 kept these two layers on the exact-f32 kernels inside the
            # bf16 model instead: 1/16 of the MFMA rate for the two largest GEMMs of the step.)
            props_features = F.pad(props_features, (0, pad))
            fc = types.SimpleNamespace(weight=F.pad(self.prop_fc.weight, (0, pad, 0, pad)), bias=F.pad(self.prop_fc.bias, (0, pad)))
        else:
            fc = self.prop_fc
        prep = DF.input_prep(props_features, props_start_end, fc, dt, want_wgrad, split_gate=getattr(self, "split_gate", False),
                             position_transform=self.position_transform)
        prep.fc = fc
        return prep

    def _front_pad(self, D):
        """Zero columns appended to the feature dim inside a bf16 model (0 when its rows already are 16-byte multiples)."""
        return (-D) % 64 if (self.compute_dtype == torch.bfloat16 and D % 8) else 0

    def forward_front(self, query_tokens, query_length, props_features, props_start_end, gates=None, prep=None):
        """-> (g0 (B, T, D+P) channels-last, gates): query encoder, gate projections, prop_fc + gating + position embedding.
        `gates`: reuse already computed gate tensors (e.g. detached ones) instead of running the query encoder; `prep`: the
        result of prepare_input() on the same features."""
        if prep is None:
            prep = self.prepare_input(props_features, props_start_end)
        if gates is None:
            gates = self.encode_query(query_tokens, query_length)
        pad = self._front_pad(self.feature_dim)
        gate0 = F.pad(gates[0], (0, pad)) if pad else gates[0]
        g0, tail = DF.input_stage(prep, prep.fc, gate0, self.position_transform, with_tail=True)
        g0._drn_tail = tail          # rides on the tensor object to forward_trunk (a caller that replaces g0 simply loses it)
        return g0, gates

    def forward_trunk(self, g0, gates, gt_start_end):
        """Backbone (gates[1:] only: level 0 is already applied), FPN, heads, losses / post-processor."""
        backbone_feats = self.backbone_net.forward_from_stage(g0, gates, tail=getattr(g0, "_drn_tail", None))
        feats = self.fpn.forward_nlc(backbone_feats)
        head = self.fcos.head
        logits, reg, iou, geo = head.forward_nlc(feats)
        box_cls, box_reg, iou_scores = (FCOSHead.split_levels(t, geo) for t in (logits, reg, iou))
        if self.taps is not None:
            for i, f in enumerate(backbone_feats):
                self.taps["backbone_net.forward_conv%d" % i] = f.permute(0, 2, 1)
            for i, f in enumerate(feats):
                self.taps["fpn.fpn_layer%d" % (i + 1)] = f.permute(0, 2, 1)
            self.taps["head"] = (box_cls, box_reg, [], iou_scores)
        fc = self.fcos
        locations = [fc.compute_locations_per_level(L, fc.fpn_strides[l], logits.device) for l, (_, L) in enumerate(geo)]
        # main_model.py:74 casts the ground truth with .float(): the loss kernel does that on load (fp64 or fp32 in)
        targets = gt_start_end if gt_start_end.dtype in (torch.float32, torch.float64) else gt_start_end.float()
        # the BatchNorm step counters owed by this forward pass are applied by the loss's own launch (DF.take_bn_counters)
        if self.training:
            res = fc._forward_train(locations, box_cls, box_reg, targets, iou_scores)
        else:
            res = fc._forward_test(locations, box_cls, box_reg, targets, iou_scores)
        DF.flush_bn_counters()
        return res

    def forward(self, query_tokens, query_length, props_features, props_start_end, gt_start_end, props_num=None,
                num_frames=None):
        g0, gates = self.forward_front(query_tokens, query_length, props_features, props_start_end)
        return self.forward_trunk(g0, gates, gt_start_end)
