#!/usr/bin/env python
"""Generate golden vectors by importing the REFERENCE (read-only, /root/reference).

Runs only in the build container (the reference never travels to the GPU box).
Outputs small .npz fixtures into tests/golden/.  Nothing from the reference is
copied: it is imported, fed seeded weights/inputs (drn_amd.utils.synthetic) and
its outputs are recorded.

Harness-side shims, none of which touch reference files:
  1. stub `fcos_core` (absent third-party CUDA extension) in sys.modules;
  2. `torch.Tensor.cuda` -> identity (model/loss.py:239, model/inference.py:193-196);
  3. wrap the focal-loss gamma/alpha in lists so the in-repo CPU formula
     (model/layers/sigmoid_focal_loss.py:40-52) indexes them.

usage: python tests/golden/gen_golden.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from drn_amd.utils.synthetic import (VOCAB_SIZE, as_namespace, default_cfg, seeded_state_dict,  # noqa: E402
                                     synthetic_batch)


def install_shims():
    names = ["fcos_core", "fcos_core.modeling", "fcos_core.modeling.box_coder", "fcos_core.modeling.utils",
             "fcos_core.structures", "fcos_core.structures.bounding_box", "fcos_core.structures.boxlist_ops"]
    for n in names:
        sys.modules[n] = types.ModuleType(n)
    sys.modules["fcos_core"]._C = types.SimpleNamespace(nms=None)
    sys.modules["fcos_core.modeling.box_coder"].BoxCoder = object
    sys.modules["fcos_core.modeling.utils"].cat = torch.cat
    sys.modules["fcos_core.structures.bounding_box"].BoxList = object
    for f in ("cat_boxlist", "boxlist_nms", "remove_small_boxes"):
        setattr(sys.modules["fcos_core.structures.boxlist_ops"], f, None)
    torch.Tensor.cuda = lambda self, *a, **k: self


def build_reference(cfg, seed=0):
    from model.main_model import mainModel
    m = mainModel(VOCAB_SIZE, as_namespace(cfg))
    f = m.fcos.loss_evaluator.cls_loss_func
    f.gamma, f.alpha = [f.gamma], [f.alpha]
    m.load_state_dict(seeded_state_dict(m, seed))
    return m


def checksum(t):
    t = t.detach().double()
    flat = t.reshape(-1)
    step = max(1, flat.numel() // 64)
    return np.array([flat.sum().item(), flat.abs().sum().item()]), flat[::step][:64].float().numpy()


TAPS = ["prop_fc", "backbone_net.forward_conv0", "backbone_net.forward_conv1", "backbone_net.forward_conv2",
        "fpn.fpn_layer1", "fpn.fpn_layer2", "fpn.fpn_layer3"]


def matched_gt(m, batch, loc0=False):
    """Pick GT = one of the model's own train-mode predictions so tIoU>0.9 positives exist.  loc0: match every clip at location
    index 0, the one location model/loss.py:180-181 clamps -- its raw start (0.5 - reg) / 32 is negative, so the clamp is ACTIVE
    on a tIoU > 0.9 positive (it decides the target and zeroes the start's gradient)."""
    state = {k: v.clone() for k, v in m.state_dict().items()}
    caught = {}
    def grab(mod, i, o):
        caught["reg"] = o[1]
    h = m.fcos.head.register_forward_hook(grab)
    m.train()
    with torch.no_grad():
        m(*batch)
    h.remove()
    m.load_state_dict(state)
    m.fcos.loss_evaluator.total_points = []
    reg0 = caught["reg"][0]                                # (B,2,T) level 0
    B, _, T = reg0.shape
    gt = []
    for b in range(B):
        t = 0 if loc0 else (5 + 7 * b) % T
        loc = t + 0.5
        if loc0:
            assert loc - reg0[b, 0, t].item() < 0.0, "loc0 case: the raw start must be negative for the clamp to act"
        s = max((loc - reg0[b, 0, t].item()) / 32.0, 0.0)
        e = min((loc + reg0[b, 1, t].item()) / 32.0, 1.0)
        # shrink by 1.5 % per side: tIoU stays ~0.97 (> 0.9) but pred != gt, so no exact min/max ties --
        # a tie would make the gradient depend on sub-ulp differences between implementations
        w = e - s
        gt.append([s + 0.015 * w, e - 0.015 * w])
    return torch.tensor(gt, dtype=torch.float64)


def run_case(name, B, T, D, stage, train=True, match=False, num_class=None, loc0=False):
    ftype = "C3D" if D == 4096 else "TINY"
    cfg = default_cfg(ftype, D, stage)
    if num_class is not None:                              # model/fcos.py:27,43: cls_logits gets fcos_num_class - 1 channels
        cfg["fcos_num_class"] = num_class
    m = build_reference(cfg, seed=0)
    batch = list(synthetic_batch(B, T, D, seed=1))
    if match:
        batch[4] = matched_gt(m, batch, loc0=loc0)
    out = {"B": B, "T": T, "D": D, "stage": stage, "train": int(train), "gt": batch[4].numpy()}
    if num_class is not None:
        out["num_class"] = num_class
    taps = {}
    hooks = []
    mods = dict(m.named_modules())
    for tname in TAPS:
        hooks.append(mods[tname].register_forward_hook(lambda mod, i, o, n=tname: taps.__setitem__(n, o)))
    hooks.append(m.fcos.head.register_forward_hook(lambda mod, i, o: taps.__setitem__("head", o)))
    m.train(train)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)                                      # model/fcos.py:182 pickles into cwd in eval
        try:
            boxes, losses = m(*batch)
        finally:
            os.chdir(cwd)
    for h in hooks:
        h.remove()
    for k in ("loss_cls", "loss_reg", "loss_iou"):
        out[k] = losses[k].detach().double().numpy().reshape(-1)
    logits, reg, _, iou = taps["head"]
    for l in range(3):
        out["logits%d" % l] = logits[l].detach().numpy()
        out["reg%d" % l] = reg[l].detach().numpy()
        out["iou%d" % l] = iou[l].detach().numpy()
    for tname in TAPS:
        cs, smp = checksum(taps[tname])
        out["cs/" + tname] = cs
        out["smp/" + tname] = smp
    if train:
        if stage == 2:
            loss = losses["loss_iou"]
        else:
            loss = sum(l for l in losses.values())
        loss.backward()
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.detach().double().reshape(-1)
            step = max(1, g.numel() // 16)
            out["gn/" + k] = np.array([g.norm().item()])
            out["gs/" + k] = g[::step][:16].float().numpy()
        for k, v in m.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                out["bn/" + k] = v.numpy().copy()
    else:
        out["n_det"] = np.array([len(b["detections"]) for b in boxes])
        out["det"] = torch.cat([b["detections"] for b in boxes]).detach().numpy()
        out["score"] = torch.cat([b["scores"] for b in boxes]).detach().numpy()
        out["loc"] = torch.cat([b["locations"] for b in boxes]).detach().numpy()
        out["level"] = np.array([x for b in boxes for lv in b["level"] for x in lv])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: out[k] for k in ("loss_cls", "loss_reg", "loss_iou")},
          "npos_iou" if stage != 1 else "", flush=True)


def run_error_cases():
    """Inputs the reference REJECTS: with one clip per batch in stages 2 / 3, `squeeze()` (model/loss.py:186) drops the batch
    dimension of iou_pred and indexing it with the (1, sumL) mask raises (model/loss.py:192) -- whether or not a tIoU > 0.9
    positive exists.  Stage 1 and eval mode with B = 1 run.  Recorded: exception type + message, and that the stage-1 / eval runs pass."""
    import json
    out = []
    for stage, train, match in ((3, True, True), (2, True, False), (1, True, False), (3, False, False)):
        cfg = default_cfg("TINY", 64, stage)
        m = build_reference(cfg, seed=0)
        batch = list(synthetic_batch(1, 32, 64, seed=1))
        if match:                                            # (matched through the stage-1 twin: same seeded weights, and it runs)
            batch[4] = matched_gt(build_reference(default_cfg("TINY", 64, 1), seed=0), batch)
        m.train(train)
        rec = {"B": 1, "T": 32, "D": 64, "stage": stage, "train": int(train), "gt": batch[4].numpy().tolist()}
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                _, losses = m(*batch)
                rec["error"] = None
                rec["losses"] = {k: float(v.reshape(-1)[0]) for k, v in losses.items()} if train else None
            except Exception as e:                           # noqa: BLE001 -- the point is to record what the reference raises
                rec["error"], rec["message"] = type(e).__name__, str(e)
            finally:
                os.chdir(cwd)
        out.append(rec)
        print("error case", {k: rec[k] for k in ("stage", "train", "error")}, flush=True)
    with open(os.path.join(HERE, "errors.json"), "w") as f:
        json.dump(out, f, indent=1)


def run_lgp():
    from model.LGP import LGP
    torch.manual_seed(0)
    g = np.random.default_rng(7)
    B, C, t = 4, 64, 16
    net = LGP(input_dim=C, query_dim=C)
    net.load_state_dict(seeded_state_dict(net, seed=3))
    net.train()
    x = torch.from_numpy(g.standard_normal((B, C, t)).astype(np.float32)).requires_grad_()
    q = torch.from_numpy(g.standard_normal((B, C)).astype(np.float32)).requires_grad_()
    y = net(x, q)
    w = torch.from_numpy(g.standard_normal(tuple(y.shape)).astype(np.float32))
    (y * w).sum().backward()
    np.savez_compressed(os.path.join(HERE, "lgp.npz"), x=x.detach().numpy(), q=q.detach().numpy(), w=w.numpy(),
                        y=y.detach().numpy(), dx=x.grad.numpy(), dq=q.grad.numpy(),
                        dw=net.query_fc[0].weight.grad.numpy(), dgamma=net.query_fc[1].weight.grad.numpy(),
                        dbeta=net.query_fc[1].bias.grad.numpy(),
                        rm=net.query_fc[1].running_mean.numpy(), rv=net.query_fc[1].running_var.numpy())
    print("lgp", float(y.abs().sum()))


def run_layers():
    """model/layers/iou_loss.py:5-24 (IOULoss, all three branches of its return) and model/LGP.py with use_bn=False (biased 1x1
    conv, BatchNorm still appended) -> layers.npz."""
    from model.layers.iou_loss import IOULoss
    from model.LGP import LGP
    g = np.random.default_rng(21)
    out = {}
    N = 37
    pred = torch.from_numpy(g.uniform(0.2, 9.0, (N, 2)).astype(np.float32))
    target = torch.from_numpy(g.uniform(0.2, 9.0, (N, 2)).astype(np.float32))
    weights = {"none": None, "pos": torch.from_numpy(g.uniform(0.0, 1.0, (N,)).astype(np.float32)),
               "zero": torch.zeros(N)}
    out["iou/pred"], out["iou/target"], out["iou/weight_pos"] = pred.numpy(), target.numpy(), weights["pos"].numpy()
    for tag, w in weights.items():
        p, t = pred.clone().requires_grad_(), target.clone().requires_grad_()
        loss = IOULoss()(p, t, w)
        (loss * 1.7).backward()
        out["iou/%s/loss" % tag] = loss.detach().numpy().reshape(1)
        out["iou/%s/dpred" % tag], out["iou/%s/dtarget" % tag] = p.grad.numpy(), t.grad.numpy()
    B, C, t = 4, 64, 16
    for mode in ("train", "eval"):
        net = LGP(input_dim=C, query_dim=C, use_bn=False)
        net.load_state_dict(seeded_state_dict(net, seed=4))
        with torch.no_grad():                                  # a bias and running statistics that matter
            net.query_fc[0].bias.copy_(torch.from_numpy(g.standard_normal(C).astype(np.float32)))
            net.query_fc[1].running_mean.copy_(torch.from_numpy(g.standard_normal(C).astype(np.float32) * 0.3))
            net.query_fc[1].running_var.copy_(torch.from_numpy(g.uniform(0.5, 2.0, C).astype(np.float32)))
        out["lgp_nobn/%s/bias" % mode] = net.query_fc[0].bias.detach().numpy().copy()
        out["lgp_nobn/%s/rm0" % mode] = net.query_fc[1].running_mean.numpy().copy()
        out["lgp_nobn/%s/rv0" % mode] = net.query_fc[1].running_var.numpy().copy()
        net.train(mode == "train")
        x = torch.from_numpy(g.standard_normal((B, C, t)).astype(np.float32)).requires_grad_()
        q = torch.from_numpy(g.standard_normal((B, C)).astype(np.float32)).requires_grad_()
        y = net(x, q)
        w = torch.from_numpy(g.standard_normal(tuple(y.shape)).astype(np.float32))
        (y * w).sum().backward()
        pre = "lgp_nobn/%s/" % mode
        out[pre + "x"], out[pre + "q"], out[pre + "w"], out[pre + "y"] = x.detach().numpy(), q.detach().numpy(), w.numpy(), y.detach().numpy()
        out[pre + "dx"], out[pre + "dq"] = x.grad.numpy(), q.grad.numpy()
        out[pre + "dw"], out[pre + "dbias"] = net.query_fc[0].weight.grad.numpy(), net.query_fc[0].bias.grad.numpy()
        out[pre + "dgamma"], out[pre + "dbeta"] = net.query_fc[1].weight.grad.numpy(), net.query_fc[1].bias.grad.numpy()
        out[pre + "rm"], out[pre + "rv"] = net.query_fc[1].running_mean.numpy().copy(), net.query_fc[1].running_var.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "layers.npz"), **out)
    print("layers", {k: float(out[k][0]) for k in out if k.endswith("/loss")})


def run_metrics():
    """utils/evaluate_utils.py:13-16,91-215,328-354 (PostProcessRunner.run_evaluate, the R@k / IoU metric of main.py:362-364)
    on a synthetic raw-results dict in main.py:324-348's format.  cwd must be the reference root while the
    runner exists (evaluate_utils.py:16 opens ./data/dataset/Charades/Charades_word2id.json)."""
    import json
    g = np.random.default_rng(11)
    results = {}
    for v in range(9):
        items = []
        for q in range(int(g.integers(1, 4))):
            gs = float(g.uniform(0, 0.5))
            gt = [gs, float(min(1.0, gs + g.uniform(0.1, 0.5)))]
            per_level = [int(g.integers(0, 9)) for _ in range(3)]
            n = sum(per_level)
            if n == 0:                                         # the post-processor's fallback entry (inference.py:192-197)
                preds, level = [[0.0, 1.0, 1.0]], [[-1]]
            else:
                s = g.uniform(0, 0.7, size=n)
                e = np.minimum(s + g.uniform(0.02, 0.5, size=n), 1.0)
                if q == 0 and n > 2:                           # a few near-hits so R@1 / R@5 differ, and one exact score tie
                    s[0], e[0] = gt[0] + 0.01, gt[1] - 0.01
                sc = g.uniform(0, 1, size=n)
                if n > 3:
                    sc[2] = sc[3]
                preds = [[float(a), float(b), float(c)] for a, b, c in zip(s, e, sc)]
                level = [[l] * c for l, c in enumerate(per_level)]
            items.append({"query": "person opens the door %d" % q, "gt": gt, "node_predictions": preds,
                          "edge_predictions": preds, "level": level})
        results["VID%02d" % v] = items
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from utils.evaluate_utils import PostProcessRunner
        out = {"results": results, "cases": []}
        for nms in (True, False):
            runner = PostProcessRunner(json.loads(json.dumps(results)))
            cfg = {"iou": [0.3, 0.5, 0.7], "topk": [1, 5]}
            topks, acc = runner.run_evaluate(iou_topk_dict=cfg, temporal_nms=nms)
            picked = {vid: [{"node_predictions": it["node_predictions"], "level": it["level"]} for it in items]
                      for vid, items in runner.viz_processed_results.items()}
            out["cases"].append({"temporal_nms": nms, "iou_topk": cfg, "topks": list(topks), "accuracy": list(acc),
                                 "last_setting_picks": picked})
            print("metrics nms=%s" % nms, topks, acc)
        r = PostProcessRunner({})
        out["nms_cases"] = []
        for n in (1, 2, 7, 20):
            x1 = g.uniform(0, 0.6, size=n); x2 = x1 + g.uniform(0.05, 0.4, size=n); sc = g.uniform(0, 1, size=n)
            for ov in (0.25, 0.45, 0.65):
                out["nms_cases"].append({"x1": x1.tolist(), "x2": x2.tolist(), "s": sc.tolist(), "overlap": ov,
                                         "pick": r.nms_temporal(x1.tolist(), x2.tolist(), sc.tolist(), ov)})
        out["iou_cases"] = [{"a": [0.1, 0.5], "b": [0.3, 0.9], "iou": r.calculate_IoU((0.1, 0.5), (0.3, 0.9))},
                            {"a": [0.1, 0.2], "b": [0.6, 0.9], "iou": r.calculate_IoU((0.1, 0.2), (0.6, 0.9))}]
    finally:
        os.chdir(cwd)
    with open(os.path.join(HERE, "metrics.json"), "w") as f:
        json.dump(out, f)


def run_trajectory(stage, steps=4, B=2, T=32, D=64, base_lr=1e-3, tag=""):
    """main.py:124-138 (stage plan) + main.py:218-243 (one iteration), restated verbatim around the REFERENCE model
    (main.py itself needs tensorboardX / ruamel / nltk and cannot be imported): per-step losses and end-of-run parameter
    checksums for the trainer counterpart.  Matched GT so the IoU-score loss is live in stages 2 and 3."""
    from torch.nn.utils import clip_grad_norm_
    cfg = default_cfg("TINY", D, stage)
    model = build_reference(cfg, seed=0)
    batches = []
    for seed in (1, 2):
        b = list(synthetic_batch(B, T, D, seed=seed))
        b[4] = matched_gt(model, b)
        batches.append(b)
    lr = base_lr
    if stage == 1:
        for name, value in model.named_parameters():
            if 'iou_scores' in name or 'mix_fc' in name:
                value.requires_grad = False
        learned = filter(lambda p: p.requires_grad, model.parameters())
    elif stage == 2:
        learned = list(model.fcos.head.iou_scores.parameters()) + list(model.fcos.head.mix_fc.parameters())
        lr /= 100
    else:
        learned = model.parameters()
        lr /= 10000
    opt = torch.optim.Adam(learned, lr)
    model.train()
    opt.zero_grad()
    losses = []
    for it in range(steps):
        model.fcos.loss_evaluator.total_points = []
        _, loss_dict = model(*batches[it % 2])
        loss = loss_dict['loss_iou'] if stage == 2 else sum(l for l in loss_dict.values())
        losses.append([float(loss_dict[k].reshape(-1)[0]) for k in ("loss_cls", "loss_reg", "loss_iou")])
        if loss != 0:
            loss.backward()
        clip_grad_norm_(model.parameters(), 0.5)
        opt.step()
        opt.zero_grad()
    out = {"stage": stage, "B": B, "T": T, "D": D, "steps": steps, "lr": base_lr, "losses": np.array(losses),
           "gt0": batches[0][4].numpy(), "gt1": batches[1][4].numpy()}
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            cs, smp = checksum(v)
            out["cs/" + k] = cs
            out["smp/" + k] = smp
    np.savez_compressed(os.path.join(HERE, "traj_s%d%s.npz" % (stage, tag)), **out)
    print("trajectory stage", stage, np.array(losses).round(5).tolist())


MINI = os.path.join(HERE, "charades_mini")


def make_mini_dataset():
    """A 6-video synthetic dataset in the reference's on-disk formats (committed under tests/golden/charades_mini)."""
    import json
    g = np.random.default_rng(5)
    base = os.path.join(MINI, "data", "dataset", "Charades")
    os.makedirs(base, exist_ok=True)
    os.makedirs(os.path.join(MINI, "features"), exist_ok=True)
    words = ["a", "person", "is", "opens", "the", "door", "puts", "book", "on", "shelf", "sits", "down", "in", "chair", "and"]
    json.dump({w: i + 1 for i, w in enumerate(words)}, open(os.path.join(base, "Charades_word2id.json"), "w"))
    fps, dur, props_txt, lines = {}, {}, [], {"train": [], "test": []}
    for v in range(6):
        vid = "V%03dX" % v
        nf = int(g.integers(200, 900))
        f = [24.0, 25.0, 29.97, 30.0][v % 4]
        fps[vid], dur[vid] = str(f), round(nf / f, 2)
        props_txt += ["#", vid, str(nf)]
        for k in range(8):                                     # 8 proposals: a ladder of scales, ends past the frame count
            L = nf / (1 + k % 4)
            s = float(g.uniform(0, max(1.0, nf - L)))
            e = s + L * float(g.uniform(0.3, 1.2))
            props_txt.append("%d %d" % (int(s), int(e)) if k % 2 else "%.2f %.2f" % (s, e))
        nseg = max(1, (nf - 16) // 8 - (v % 2) * 3)             # some videos store fewer segments than frames imply
        torch.save(torch.from_numpy(g.standard_normal((nseg, 12)).astype(np.float32)), os.path.join(MINI, "features", vid + ".pt"))
        for q in range(2):
            n = int(g.integers(3, 9))
            sent = " ".join(words[int(i)] for i in g.integers(0, len(words), size=n)) + "."
            s = float(g.uniform(0, dur[vid] * 0.6))
            e = s + float(g.uniform(2.0, dur[vid]))             # may exceed the duration: capped by the reader
            lines["train" if (v + q) % 3 else "test"].append("%s %.1f %.1f##%s" % (vid, s, e, sent))
    json.dump(fps, open(os.path.join(base, "Charades_fps_dict.json"), "w"))
    json.dump(dur, open(os.path.join(base, "Charades_duration.json"), "w"))
    open(os.path.join(base, "mini_props.txt"), "w").write("\n".join(props_txt) + "\n")
    for sp, ls in lines.items():
        open(os.path.join(base, "Charades_sta_%s.txt" % sp), "w").write("\n".join(ls) + "\n")


def mini_config():
    return {"feature_type": "C3D", "C3D": {"feature_root": "./features", "feature_dim": 12, "ft_window_size": 16, "ft_overlap": 0.5},
            "props_file_path": "./data/dataset/Charades/mini_props.txt"}


def run_dataset():
    """dataset.py:61-224 (CharadesSTA + collate_data) on the mini dataset.  dataset.py imports PIL / torchvision / nltk at
    module level without needing them on this path: PIL and torchvision are stubbed, nltk.word_tokenize -> str.split (the
    mini sentences are plain space-separated words, where the two agree)."""
    import argparse
    for name in ["PIL", "PIL.Image", "torchvision", "torchvision.transforms"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    nl = types.ModuleType("nltk"); nl.word_tokenize = lambda s: s.split(); sys.modules["nltk"] = nl
    make_mini_dataset()
    cwd = os.getcwd()
    os.chdir(MINI)
    try:
        import dataset as refds
        out = {}
        for split in ("train", "test"):
            ds = refds.CharadesSTA(argparse.Namespace(**mini_config()), split=split)
            items = [ds[i] for i in range(len(ds))]
            names, pse, feats, gt, tok, qlen, nprops, nframes = refds.collate_data(items)
            out[split + "/names"] = np.array(names)
            out[split + "/props_s_e"] = pse.numpy(); out[split + "/feats"] = feats.numpy(); out[split + "/gt"] = gt.numpy()
            out[split + "/tokens"] = tok.numpy(); out[split + "/qlen"] = qlen.numpy(); out[split + "/nprops"] = nprops.numpy()
            out[split + "/nframes"] = nframes.numpy()
            print("dataset", split, len(ds), feats.shape, float(feats.abs().sum()))
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)


def run_keys():
    import json
    m = build_reference(default_cfg("C3D", 4096, 1))
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump({k: list(v.shape) for k, v in m.state_dict().items()}, f, indent=0)


if __name__ == "__main__":
    torch.set_num_threads(8)
    install_shims()
    if sys.argv[1:] == ["classes"]:                        # fcos_num_class = 4: three foreground channels (off every shipped config)
        run_case("tiny_k3_s1", 2, 32, 64, 1, num_class=4)
        run_case("tiny_k3_s3", 2, 32, 64, 3, match=True, num_class=4)
        run_case("tiny_k3_eval", 2, 32, 64, 3, train=False, num_class=4)
        run_case("tiny_k2_s3", 2, 32, 64, 3, match=True, num_class=3)      # two channels: one head call, no chunking
        sys.exit(0)
    if sys.argv[1:] == ["loc0"]:                           # round 6: active clamp (loss.py:180-181) and B = 1 (squeeze, loss.py:186)
        run_case("tiny_s3_loc0", 2, 32, 64, 3, match=True, loc0=True)
        run_case("tiny_s2_loc0", 2, 32, 64, 2, match=True, loc0=True)
        run_error_cases()
        sys.exit(0)
    if sys.argv[1:] == ["layers"]:
        run_layers()
        sys.exit(0)
    if sys.argv[1:] == ["metrics"]:
        run_metrics()
        sys.exit(0)
    if sys.argv[1:] == ["dataset"]:
        run_dataset()
        sys.exit(0)
    if sys.argv[1:] == ["trajectory"]:
        for st in (1, 2, 3):
            run_trajectory(st)
        run_trajectory(1, base_lr=1e-5, tag="_lowlr")
        sys.exit(0)
    run_keys()
    run_case("tiny_s1", 2, 32, 64, 1)
    run_case("tiny_s3", 2, 32, 64, 3, match=True)
    run_case("tiny_s2", 2, 32, 64, 2, match=True)
    run_case("tiny_eval", 2, 32, 64, 3, train=False)
    run_case("tiny_eval_s1", 3, 64, 64, 1, train=False)
    run_case("c3d_s1", 2, 64, 4096, 1)
    run_case("c3d_s3", 2, 64, 4096, 3, match=True)
    run_case("tiny_k3_s1", 2, 32, 64, 1, num_class=4)
    run_case("tiny_k3_s3", 2, 32, 64, 3, match=True, num_class=4)
    run_case("tiny_k3_eval", 2, 32, 64, 3, train=False, num_class=4)
    run_case("tiny_k2_s3", 2, 32, 64, 3, match=True, num_class=3)
    run_case("tiny_s3_loc0", 2, 32, 64, 3, match=True, loc0=True)
    run_case("tiny_s2_loc0", 2, 32, 64, 2, match=True, loc0=True)
    run_error_cases()
    run_lgp()
    run_layers()
    run_metrics()
    run_dataset()
    for st in (1, 2, 3):
        run_trajectory(st)
    run_trajectory(1, base_lr=1e-5, tag="_lowlr")
