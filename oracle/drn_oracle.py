"""Plain-PyTorch fp32 restatement of the DRN forward/backward hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  This is the CPU oracle the
HIP path is checked against and the `cpu_baseline` that bench.py times.  It is
pinned against golden vectors produced by importing the reference itself
(tests/golden/gen_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).

Every class keeps the reference's attribute names so `state_dict()` keys are
identical and reference checkpoints load unchanged.
Citations are relative to /root/reference.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

INF = 100000000.0
# model/loss.py:47-51 -- regression ranges per pyramid level (T-independent)
SIZES_OF_INTEREST = ((-1.0, 6.0), (5.6, 11.0), (11.0, INF))
# model/loss.py:98,178 ; model/inference.py:45 -- hard-coded proposal count
TARGET_SCALE = 32.0


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def conv_bn_relu(cin, cout, k=3, stride=1):
    """model/basic_blocks.py:5-33 with use_bn=True,use_relu=True:
    Conv1d(no bias, pad (k-1)//2) -> BatchNorm1d -> ReLU, kaiming_uniform(a=1)."""
    conv = nn.Conv1d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=False)
    nn.init.kaiming_uniform_(conv.weight, a=1)
    return nn.Sequential(conv, nn.BatchNorm1d(cout), nn.ReLU(inplace=True))


class XavierLinear(nn.Linear):
    """model/ops.py:16-25 -- U(+-sqrt(3/fan_avg)), zero bias."""

    def __init__(self, fin, fout):
        super().__init__(fin, fout)
        bound = math.sqrt(3.0 / ((fin + fout) / 2.0))
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.constant_(self.bias, 0.0)


class TextualAttention(nn.Module):
    """model/language_module.py:65-74 -- constructed, never called (keys only)."""

    def __init__(self, hidden_dim=1024):
        super().__init__()
        self.W1 = nn.Linear(hidden_dim, 1)
        self.W2 = nn.Linear(2 * hidden_dim, hidden_dim)
        self.W3 = nn.Linear(2 * hidden_dim, 2 * hidden_dim)


class QueryEncoder(nn.Module):
    """model/language_module.py:9-62."""

    def __init__(self, vocab_size, hidden_dim=512, embed_dim=300, num_layers=1, bidirection=True):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.embedding = nn.Embedding(vocab_size + 1, embed_dim, padding_idx=0)
        self.biLSTM = nn.LSTM(embed_dim, hidden_dim, num_layers, dropout=0.0,
                              batch_first=True, bidirectional=bidirection)
        self.textualAttention = TextualAttention()
        self.qInput = XavierLinear(hidden_dim * 4, hidden_dim)
        for t in range(3):
            setattr(self, "qInput%d" % t, XavierLinear(hidden_dim, hidden_dim * 2))
        self.cmd_inter2logits = XavierLinear(hidden_dim * 2, 1)

    def forward(self, tokens, lengths):
        emb = self.embedding(tokens)
        packed = pack_padded_sequence(emb, lengths.cpu(), batch_first=True)
        out, _ = self.biLSTM(packed)
        out, _ = pad_packed_sequence(out, batch_first=True)  # (B, Lmax, 1024)
        B, Lmax, _ = out.shape
        # language_module.py:48-54: cat(first step, last valid step), both full 1024-d
        idx = (lengths.to(out.device) - 1).view(B, 1, 1).expand(B, 1, out.size(2))
        q_vec = torch.cat([out[:, 0], out.gather(1, idx).squeeze(1)], dim=-1)
        base = F.relu(self.qInput(q_vec))
        pos = torch.arange(Lmax, device=out.device).view(1, Lmax)
        pad = pos >= lengths.to(out.device).view(B, 1)
        cmds = []
        for t in range(3):
            # language_module.py:27-36
            q_cmd = getattr(self, "qInput%d" % t)(base)
            raw = self.cmd_inter2logits(q_cmd[:, None, :] * out).squeeze(-1)
            raw = raw.masked_fill(pad, -1e30)
            att = F.softmax(raw, dim=-1)
            cmds.append(torch.bmm(att[:, None, :], out).squeeze(1))
        return cmds


class Backbone(nn.Module):
    """model/backbone.py:4-36."""

    def __init__(self, channels_list, conv_block=conv_bn_relu):
        super().__init__()
        self.num_layers = len(channels_list)
        for i, (cin, cout, k, s) in enumerate(channels_list):
            self.add_module("forward_conv%d" % i, conv_block(cin, cout, k, s))

    def forward(self, x, query_fts, position_fts):
        outs = []
        for i in range(self.num_layers):
            x = query_fts[i][:, :, None] * x                 # backbone.py:28-30
            if i == 0:
                x = torch.cat([x, position_fts[i]], dim=1)   # backbone.py:31-32
            x = getattr(self, "forward_conv%d" % i)(x)
            outs.append(x)
        return tuple(outs)


class FPN(nn.Module):
    """model/FPN.py:7-78 (top_blocks=None)."""

    def __init__(self, in_channels_list, out_channels, conv_block=conv_bn_relu):
        super().__init__()
        self.levels = len(in_channels_list)
        for i, cin in enumerate(in_channels_list, 1):
            self.add_module("fpn_inner%d" % i, conv_block(cin, out_channels, 1))
            self.add_module("fpn_layer%d" % i, conv_block(out_channels, out_channels, 3, 1))

    def forward(self, feats):
        n = self.levels
        last = getattr(self, "fpn_inner%d" % n)(feats[-1])
        outs = [getattr(self, "fpn_layer%d" % n)(last)]
        for lvl in range(n - 1, 0, -1):
            lateral = getattr(self, "fpn_inner%d" % lvl)(feats[lvl - 1])
            last = lateral + last.repeat_interleave(2, dim=-1)   # nearest x2, FPN.py:63-68
            outs.insert(0, getattr(self, "fpn_layer%d" % lvl)(last))
        return tuple(outs)


class Scale(nn.Module):
    """model/fcos.py:10-16."""

    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([float(init_value)]))

    def forward(self, x):
        return x * self.scale


class FCOSHead(nn.Module):
    """model/fcos.py:19-105."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        C = in_channels
        ncls = cfg["fcos_num_class"] - 1

        def tower():
            mods = []
            for _ in range(cfg["fcos_conv_layers"]):
                mods += [nn.Conv1d(C, C, 3, padding=1), nn.BatchNorm1d(C), nn.ReLU()]
            return nn.Sequential(*mods)

        self.cls_tower = tower()
        self.bbox_tower = tower()
        self.cls_logits = nn.Conv1d(C, ncls, 3, padding=1)
        self.bbox_pred = nn.Conv1d(C, 2, 3, padding=1)
        self.centerness = nn.Conv1d(C, 1, 3, padding=1)          # declared, never called
        self.mix_fc = nn.Sequential(nn.Conv1d(2 * C, C, 1), nn.BatchNorm1d(C), nn.ReLU())
        self.iou_scores = nn.Sequential(nn.Conv1d(C, C // 2, 3, padding=1), nn.BatchNorm1d(C // 2),
                                        nn.ReLU(), nn.Conv1d(C // 2, 1, 1))
        for grp in (self.cls_tower, self.bbox_tower, self.cls_logits, self.bbox_pred,
                    self.centerness, self.iou_scores, self.mix_fc):
            for m in grp.modules():
                if isinstance(m, nn.Conv1d):
                    nn.init.normal_(m.weight, std=0.01)
                    nn.init.constant_(m.bias, 0.0)
        p = cfg["fcos_prior_prob"]
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - p) / p))
        self.scales = nn.ModuleList([Scale(1.0) for _ in range(3)])

    def forward(self, feats):
        logits, reg, iou = [], [], []
        for l, f in enumerate(feats):
            ct = self.cls_tower(f)
            bt = self.bbox_tower(f)
            logits.append(self.cls_logits(ct))
            reg.append(torch.exp(self.scales[l](self.bbox_pred(bt))))
            iou.append(self.iou_scores(self.mix_fc(torch.cat([ct, bt], dim=1))))
        return logits, reg, [], iou


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def sigmoid_focal_loss_sum(logits, targets, gamma, alpha):
    """model/layers/sigmoid_focal_loss.py:40-52,61-69 (the in-repo CPU formula), summed."""
    C = logits.shape[1]
    cls = torch.arange(1, C + 1, dtype=targets.dtype, device=targets.device)[None]
    t = targets[:, None]
    p = torch.sigmoid(logits)
    pos = (t == cls).float()
    neg = ((t != cls) & (t >= 0)).float()
    loss = -pos * alpha * (1 - p) ** gamma * torch.log(p) - neg * (1 - alpha) * p ** gamma * torch.log(1 - p)
    return loss.sum()


def iou_loss_mean(pred, target):
    """model/layers/iou_loss.py:6-24."""
    inter = torch.min(pred[:, 1], target[:, 1]) + torch.min(pred[:, 0], target[:, 0])
    union = (target[:, 0] + target[:, 1]) + (pred[:, 0] + pred[:, 1]) - inter
    return (-torch.log((inter + 1e-8) / (union + 1e-8))).mean()


def iou_loss(pred, target, weight=None):
    """model/layers/iou_loss.py:6-24, the `weight` branch included (the module model/layers/__init__.py exports)."""
    inter = torch.min(pred[:, 1], target[:, 1]) + torch.min(pred[:, 0], target[:, 0])
    union = (target[:, 0] + target[:, 1]) + (pred[:, 0] + pred[:, 1]) - inter
    losses = -torch.log((inter + 1e-8) / (union + 1e-8))
    if weight is not None and weight.sum() > 0:
        return (losses * weight).sum() / weight.sum()
    assert losses.numel() != 0
    return losses.mean()


def segment_tiou(a, b):
    """model/loss.py:241-256."""
    inter = torch.clamp(torch.min(a[..., 1], b[..., 1]) - torch.max(a[..., 0], b[..., 0]), min=0)
    union = torch.clamp(torch.max(a[..., 1], b[..., 1]) - torch.min(a[..., 0], b[..., 0]), min=0)
    return inter / (union + 1e-6)


def fcos_targets(locations, gt):
    """model/loss.py:40-127.  locations: list of (L_l,) ; gt: (B,2) normalised.
    Returns labels (sumB L,) float {0,1} and reg_targets (sumB L, 2), ordered
    level-first / clip-major exactly like the reference's flattened tensors."""
    labels, regs = [], []
    for lvl, loc in enumerate(locations):
        lo, hi = SIZES_OF_INTEREST[lvl]
        bb = gt * TARGET_SCALE                                  # (B,2)
        l = loc[None, :] - bb[:, 0:1]                           # (B,L)
        r = bb[:, 1:2] - loc[None, :]
        mn, mx = torch.min(l, r), torch.max(l, r)
        lab = ((mn > 0) & (mx >= lo) & (mx <= hi)).to(l.dtype)
        labels.append(lab.reshape(-1))
        regs.append(torch.stack([l, r], dim=-1).reshape(-1, 2))
    return torch.cat(labels), torch.cat(regs)


class FCOSLoss(object):
    """model/loss.py:22-239."""

    def __init__(self, cfg):
        self.gamma = cfg["fcos_loss_gamma"]
        self.alpha = cfg["fcos_loss_alpha"]

    def __call__(self, locations, box_cls, box_regression, targets, iou_scores, is_first_stage=True):
        N = box_cls[0].size(0)
        ncls = box_cls[0].size(1)
        labels, reg_targets = fcos_targets(locations, targets)
        cls_flat = torch.cat([c.permute(0, 2, 1).reshape(-1, ncls) for c in box_cls])
        reg_flat = torch.cat([r.permute(0, 2, 1).reshape(-1, 2) for r in box_regression])

        iou_loss = None
        if not is_first_stage:
            merged = torch.cat(box_regression, dim=-1).transpose(2, 1)      # (B, sumL, 2)
            loc = torch.cat(locations)[None, :]
            pred = torch.stack([loc - merged[:, :, 0], loc + merged[:, :, 1]], dim=-1) / TARGET_SCALE
            # loss.py:180-181: clamps LOCATION index 0 of every clip
            head = pred[:, :1].clamp(min=0, max=1)
            pred = torch.cat([head, pred[:, 1:]], dim=1)
            iou_t = segment_tiou(pred, targets[:, None, :])
            iou_p = torch.cat(iou_scores, dim=-1).squeeze().sigmoid()       # A.3 #4
            mask = iou_t > 0.9
            # loss.py:189-192 index BEFORE the empty check: with B == 1 `squeeze()` has dropped the batch dimension of iou_p and the
            # (1, sumL) mask raises IndexError whether or not a positive exists (tests/golden/errors.json records the reference doing so)
            pos_t, pos_p = iou_t[mask], iou_p[mask]
            if int(mask.sum()) == 0:
                iou_loss = torch.tensor([0])                               # A.3 #3
            else:
                # target keeps its graph (A.3 #2)
                iou_loss = F.smooth_l1_loss(pos_p, pos_t)

        pos = torch.nonzero(labels > 0).squeeze(1)
        cls_loss = sigmoid_focal_loss_sum(cls_flat, labels.int(), self.gamma, self.alpha) / (pos.numel() + N)
        if pos.numel() > 0:
            reg_loss = iou_loss_mean(reg_flat[pos], reg_targets[pos])
        else:
            reg_loss = reg_flat[pos].sum()
        if is_first_stage:
            iou_loss = torch.zeros(1, device=cls_loss.device)
        return cls_loss, reg_loss, iou_loss


# --------------------------------------------------------------------------
# eval post-processing
# --------------------------------------------------------------------------
class FCOSPostProcessor(object):
    """model/inference.py:11-215."""

    def __init__(self, cfg):
        self.thr = cfg["fcos_inference_thr"]
        self.top_n = cfg["fcos_pre_nms_top_n"]
        self.is_first_stage = cfg["is_first_stage"]

    def _level(self, loc, cls, reg, level, iou):
        N = cls.shape[0]
        cls = cls.permute(0, 2, 1).sigmoid()             # (N, L, C)
        iou = iou.permute(0, 2, 1).sigmoid()
        reg = reg.permute(0, 2, 1)
        cand = cls > self.thr                            # before x iou (A.3 #9)
        topn = cand.reshape(N, -1).sum(1).clamp(max=self.top_n)
        if not self.is_first_stage:
            cls = cls * iou
        out = []
        for i in range(N):
            sc = cls[i][cand[i]]
            nz = cand[i].nonzero()
            where = nz[:, 0]
            breg = reg[i][where]
            bloc = loc[where]
            if int(cand[i].sum()) > int(topn[i]):
                sc, keep = sc.topk(int(topn[i]), sorted=False)
                breg, bloc = breg[keep], bloc[keep]
            det = torch.stack([bloc - breg[:, 0], bloc + breg[:, 1]], dim=1) / TARGET_SCALE
            det = det.clamp(min=0, max=1)
            out.append({"detections": det, "labels": nz[:, 1] + 1, "scores": torch.sqrt(sc),
                        "level": [level], "locations": bloc / 32})
        return out

    def __call__(self, locations, box_cls, box_regression, iou_scores):
        per_level = [self._level(l, c, r, i, s)
                     for i, (l, c, r, s) in enumerate(zip(locations, box_cls, box_regression, iou_scores))]
        results = []
        for dicts in zip(*per_level):
            det = [d["detections"] for d in dicts if len(d["detections"])]
            dev = dicts[0]["detections"].device
            if not det:                                   # inference.py:192-197
                res = {"detections": torch.tensor([[0.0, 1.0]], device=dev), "labels": [],
                       "scores": torch.tensor([1.0], device=dev), "level": [[-1]],
                       "locations": torch.tensor([0.5], device=dev)}
            else:
                res = {"detections": torch.cat(det), "labels": [],
                       "scores": torch.cat([d["scores"] for d in dicts if len(d["scores"])]),
                       "level": [d["level"] * len(d["detections"]) for d in dicts if len(d["level"])],
                       "locations": torch.cat([d["locations"] for d in dicts if len(d["locations"])])}
            results.append(res)
        return results


class FCOSModule(nn.Module):
    """model/fcos.py:108-211 (without the pickle side effect)."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        self.head = FCOSHead(cfg, in_channels)
        self.is_first_stage = cfg["is_first_stage"]
        self.box_selector_test = FCOSPostProcessor(cfg)
        self.loss_evaluator = FCOSLoss(cfg)
        self.fpn_strides = cfg["fpn_stride"]

    @staticmethod
    def locations_for(t, stride, device):
        # fcos.py:204-211
        return torch.arange(0, t * stride, step=stride, dtype=torch.float32, device=device) + stride / 2

    def forward(self, features, targets=None):
        cls, reg, _, iou = self.head(features)
        locs = [self.locations_for(f.size(-1), self.fpn_strides[l], f.device) for l, f in enumerate(features)]
        boxes = None
        if not self.training:
            boxes = self.box_selector_test(locs, cls, reg, iou)
        lc, lr, li = self.loss_evaluator(locs, cls, reg, targets, iou, self.is_first_stage)
        return boxes, {"loss_cls": lc, "loss_reg": lr, "loss_iou": li}


def build_fcos(cfg, in_channels):
    return FCOSModule(cfg, in_channels)


class LGP(nn.Module):
    """model/LGP.py:3-51 (dead in the reference; standalone op)."""

    def __init__(self, input_dim=1024, query_dim=1024, use_bn=True):
        super().__init__()
        conv = nn.Conv1d(query_dim, input_dim, 1, bias=not use_bn)       # LGP.py:18 (the BatchNorm is appended either way, :25)
        nn.init.kaiming_uniform_(conv.weight, a=1)
        self.query_fc = nn.Sequential(conv, nn.BatchNorm1d(input_dim))

    def forward(self, inputs, query):
        B, C, t = inputs.shape
        q = self.query_fc(query[:, :, None].repeat(1, 1, t))
        att = (inputs * q).view(B, C, t // 2, 2).sum(1)
        att = F.softmax(att, dim=-1)[:, None]
        return (inputs.view(B, C, t // 2, 2) * att).sum(-1)


class mainModel(nn.Module):
    """model/main_model.py:13-81."""

    def __init__(self, vocab_size, dataset_configs, hidden_dim=512, embed_dim=300, bidirection=True,
                 graph_node_features=1024):
        super().__init__()
        cfg = vars(dataset_configs) if not isinstance(dataset_configs, dict) else dataset_configs
        self.first_output_dim = cfg["first_output_dim"]
        self.fpn_feature_dim = cfg["fpn_feature_dim"]
        self.feature_dim = cfg[cfg["feature_type"]]["feature_dim"]
        self.query_encoder = QueryEncoder(vocab_size, hidden_dim, embed_dim, cfg["lstm_layers"], bidirection)
        d = self.first_output_dim
        chans = [(self.feature_dim + 256, d, 3, 1), (d, 2 * d, 3, 2), (2 * d, 4 * d, 3, 2)]
        self.backbone_net = Backbone(chans)
        self.fpn = FPN([256, 512, 1024], 512)
        self.fcos = build_fcos(cfg, self.fpn_feature_dim)
        self.prop_fc = nn.Linear(self.feature_dim, self.feature_dim)
        self.position_transform = nn.Linear(3, 256)
        for t in range(3):
            setattr(self, "qInput%d" % t, nn.Linear(1024, self.feature_dim if t == 0 else chans[t - 1][1]))

    def forward(self, query_tokens, query_length, props_features, props_start_end, gt_start_end,
                props_num=None, num_frames=None):
        q = self.query_encoder(query_tokens, query_length)
        q = [getattr(self, "qInput%d" % i)(q[i]) for i in range(3)]
        # main_model.py:51-55; only level 0 is consumed (backbone.py:31)
        dur = (props_start_end[:, :, 1] - props_start_end[:, :, 0]).unsqueeze(-1)
        pos = self.position_transform(torch.cat([props_start_end, dur], dim=-1).float()).permute(0, 2, 1)
        x = self.prop_fc(props_features).permute(0, 2, 1)
        feats = self.fpn(self.backbone_net(x, q, [pos, None, None]))
        return self.fcos(feats, gt_start_end.float())
