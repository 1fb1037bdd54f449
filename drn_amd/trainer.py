"""Trainer / evaluator counterpart of the reference's main.py:44-456 (own code) -- SURVEY row 8f-1.

  * the 3-stage schedule (main.py:124-138): stage 1 freezes `iou_scores` / `mix_fc`, 10 epochs, lr; stage 2 trains only
    `fcos.head.iou_scores` + `fcos.head.mix_fc` on `loss_iou` with lr/100; stage 3 trains everything with lr/1e4;
  * one step (main.py:218-243): forward, `loss = loss_iou` (stage 2) or the sum of the three, backward,
    `clip_grad_norm_(model.parameters(), 0.5)`, Adam, zero_grad;
  * evaluation (main.py:270-366): post-processor dicts -> raw-results records -> R@1 / R@5 at IoU 0.5 with temporal NMS
    (drn_amd.metrics.PostProcessRunner);
  * checkpoints (main.py:160-166,369-373): {'epoch','state_dict','loss','top1','top5'} under the reference's file name,
    state_dict keys carrying the `module.` prefix nn.DataParallel gave them, key-filtered resume (main.py:104-111).

MI355X side: one process per GPU (drn_amd.dist.GradReducer: RCCL all-reduce of the flat gradient buckets) and the fused
clip+Adam kernels (drn_amd.optim.FusedAdam) in stages 1 and 3.  Stage 2 keeps torch's optimizer on purpose: the
reference clips over ALL parameters while zeroing only the optimizer's (main.py:238-243), so the gradients of the frozen
trunk accumulate step after step and keep shrinking the clip coefficient -- reproduced exactly by leaving those
`p.grad` to autograd; it trains 0.9 M parameters, so the optimizer is not the cost there."""
import os

import torch

from . import functional as DF
from .metrics import PostProcessRunner, results_entry


def stage_plan(model, stage, lr):
    """(learned parameters, lr, default n_epoch, loss selector) per main.py:124-138.  Mutates requires_grad in stage 1."""
    if stage == 1:
        for name, p in model.named_parameters():
            if "iou_scores" in name or "mix_fc" in name:
                p.requires_grad = False
        return [p for p in model.parameters() if p.requires_grad], lr, 10, "sum"
    if stage == 2:
        head = model.fcos.head
        return list(head.iou_scores.parameters()) + list(head.mix_fc.parameters()), lr / 100, None, "loss_iou"
    if stage == 3:
        return list(model.parameters()), lr / 10000, None, "sum"
    raise ValueError("stage must be 1, 2 or 3")


def select_loss(loss_dict, which):
    return loss_dict["loss_iou"] if which == "loss_iou" else DF.loss_total(loss_dict)                 # main.py:222-225


def to_device(batch, device):
    """collate_data's 8-tuple (names first) -> the model's 7 arguments on `device`."""
    names, pse, feats, gt, tok, qlen, nprops, nframes = batch
    mv = lambda t: t.to(device, non_blocking=True)
    return names, (mv(tok), mv(qlen), mv(feats), mv(pse), mv(gt), nprops, nframes)


class Trainer(object):
    def __init__(self, model, stage, lr=1e-3, clip_gradient=0.5, world_size=1, fused=True):
        self.model, self.stage, self.clip = model, stage, clip_gradient
        self.params, self.lr, self.default_epochs, self.which = stage_plan(model, stage, lr)
        self.device = next(model.parameters()).device
        self.fused = fused and stage != 2 and self.device.type == "cuda"
        self.world_size = world_size
        if self.fused:
            from .dist import GradReducer
            from .optim import FusedAdam
            self.reducer = GradReducer(self.params, world_size=world_size, adjacent=model.grad_stack_groups(),
                                       **({"bucket_bytes": 1 << 30} if world_size == 1 else {}))   # one process: one bucket
            self.opt = FusedAdam(self.reducer, lr=self.lr, max_norm=clip_gradient if clip_gradient is not None else 0.0)
        else:
            DF.unregister_grad_sinks(model.parameters())   # autograd owns this model's gradients (other models keep their sinks)
            self.reducer = None
            self.opt = torch.optim.Adam(self.params, self.lr)
            self.opt.zero_grad()

    def train_step(self, args):
        """One main.py:218-243 iteration on device-resident arguments.  Returns the loss dict (device tensors, no sync)."""
        self.model.train()
        if self.fused:
            self.reducer.zero()
        _, loss_dict = self.model(*args)
        DF.backward(select_loss(loss_dict, self.which))
        if self.fused:
            self.reducer.finish()
            self.opt.step()
        else:
            if self.world_size > 1:
                self._average_grads()
            if self.clip is not None:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)     # over ALL parameters (main.py:238-239)
            self.opt.step()
            self.opt.zero_grad()                                                          # only the optimizer's (main.py:243)
        return loss_dict

    def _average_grads(self):
        """Un-fused path under one-process-per-GPU: all-reduce(sum)/world of EVERY existing gradient -- the frozen trunk's
        too, because the clip below runs over all parameters (main.py:238-239).  Those accumulate from step to step and
        the averaging is linear and idempotent on the already-averaged part, so the stage-2 quirk is preserved."""
        import torch.distributed as td
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        if not grads:
            return
        flat = torch._utils._flatten_dense_tensors(grads)
        td.all_reduce(flat)
        flat.div_(self.world_size)
        for g, r in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(r)

    def train_epoch(self, loader, epoch=None):
        # one process per GPU: the DistributedSampler reshuffles (and re-shards) per epoch only when told the epoch
        sampler = getattr(loader, "sampler", None)
        if epoch is not None and hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        total, n = None, 0
        for batch in loader:
            _, args = to_device(batch, self.device)
            bs = args[2].size(0)
            loss = select_loss(self.train_step(args), self.which).detach().reshape(-1)[0] * bs
            total = loss if total is None else total + loss
            n += bs
        return float(total) / max(n, 1) if total is not None else 0.0

    @torch.no_grad()
    def evaluate(self, loader, id2word=None, iou_topk=None):
        """main.py:270-366: returns (mean loss, topks, accuracies, raw results dict)."""
        self.model.eval()
        results, total, n = {}, 0.0, 0
        for batch in loader:
            names, args = to_device(batch, self.device)
            boxes, loss_dict = self.model(*args)
            bs = args[2].size(0)
            total += float(select_loss(loss_dict, self.which).reshape(-1)[0]) * bs
            n += bs
            tokens, qlen, gts = args[0].cpu(), args[1].cpu(), args[4].cpu().numpy()
            for i in range(bs):
                words = [id2word[int(t)] if id2word else str(int(t)) for t in tokens[i, :int(qlen[i])]]
                results.setdefault(names[i], []).append(results_entry(" ".join(words), gts[i], boxes[i]))
        iou_topk = iou_topk or {"iou": [0.5], "topk": [1, 5]}                            # main.py:362
        topks, accs = PostProcessRunner(results).run_evaluate(iou_topk_dict=iou_topk, temporal_nms=True)
        return total / max(n, 1), topks, accs, results

    def fit(self, train_loader, test_loader, n_epoch=None, eval_freq=1, snapshot_pref=None, dataset="Charades", id2word=None,
            start_epoch=0):
        """main.py:142-190: train, validate every eval_freq epochs, keep the best-R@1 and best-R@5 checkpoints."""
        n_epoch = self.default_epochs if self.default_epochs is not None else n_epoch
        best1 = best5 = 0.0
        history = []
        for epoch in range(start_epoch, n_epoch):
            train_loss = self.train_epoch(train_loader, epoch)
            rec = {"epoch": epoch, "train_loss": train_loss}
            if (epoch + 1) % eval_freq == 0 or epoch == n_epoch - 1:
                val_loss, topks, accs, _ = self.evaluate(test_loader, id2word)
                top1, top5 = accs[0] * 100, accs[1] * 100
                rec.update(val_loss=val_loss, top1=top1, top5=top5)
                state = {"epoch": epoch + 1, "state_dict": checkpoint_state_dict(self.model), "loss": val_loss, "top1": top1,
                         "top5": top5}
                if snapshot_pref is not None:
                    if top1 > best1:
                        save_checkpoint(state, snapshot_pref, dataset, epoch, top1, top5)
                    if top5 > best5:
                        save_checkpoint(state, snapshot_pref, dataset, epoch, top1, top5)
                best1, best5 = max(best1, top1), max(best5, top5)
            history.append(rec)
        return history


    def fit_train_only(self, train_loader, n_epoch, start_epoch=0):
        """Ranks other than 0 of a multi-GPU run: the SAME epochs as rank 0's fit() (every step holds a collective), no
        evaluation / checkpoints."""
        n_epoch = self.default_epochs if self.default_epochs is not None else n_epoch
        return [{"epoch": e, "train_loss": self.train_epoch(train_loader, e)} for e in range(start_epoch, n_epoch)]


# ---------------------------------------------------------------------------------------------- checkpoints
def checkpoint_state_dict(model):
    """state_dict with the `module.` prefix the reference's nn.DataParallel wrapper puts on every key (main.py:99,161)."""
    DF.flush_bn_counters()
    return {"module." + k: v for k, v in model.state_dict().items()}


def checkpoint_name(snapshot_pref, dataset, epoch, top1, top5):
    return "{}/model_{}_epoch{}_top1_{:.3f}_top5_{:.3f}_model_best.pth.tar".format(snapshot_pref, dataset, epoch, top1, top5)


def save_checkpoint(state, snapshot_pref, dataset, epoch, top1, top5):
    os.makedirs(snapshot_pref, exist_ok=True)
    path = checkpoint_name(snapshot_pref, dataset, epoch, top1, top5)
    torch.save(state, path)
    return path


def load_checkpoint(model, path, map_location="cpu"):
    """Key-filtered resume (main.py:104-111): only keys the model has are taken; a `module.` prefix is accepted."""
    ckpt = torch.load(path, map_location=map_location)
    own = model.state_dict()
    picked = {}
    for k, v in ckpt["state_dict"].items():
        k = k[len("module."):] if k.startswith("module.") else k
        if k in own:
            picked[k] = v
    own.update(picked)
    model.load_state_dict(own)
    DF.bump_weights_epoch()
    return ckpt.get("epoch", 0), sorted(picked)


def init_glove(model, glove_weights):
    """main.py:92-94: copy the (vocab+1, 300) GloVe table into the embedding if the file exists."""
    if glove_weights and os.path.exists(glove_weights):
        model.query_encoder.embedding.weight.data.copy_(torch.load(glove_weights))
        DF.bump_weights_epoch()
        return True
    return False
