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
import contextlib
import functools
import os

import torch

from . import functional as DF
import numpy as np

from .metrics import results_entries, PostProcessRunner, results_entry, recall_from_first_hits


def stage_plan(model, stage, lr):
    """(learned parameters, lr, default n_epoch, loss selector) per main.py:124-138.  Mutates requires_grad in stage 1.
    Parameters the forward pass never reaches (mainModel.unused_parameters: the reference's dead TextualAttention) are left out:
    their gradient is None in the reference, where clip_grad_norm_ and Adam skip them -- same result, nothing to exchange/update."""
    dead = set(id(p) for p in model.unused_parameters()) if hasattr(model, "unused_parameters") else set()
    if stage == 1:
        for name, p in model.named_parameters():
            if "iou_scores" in name or "mix_fc" in name:
                p.requires_grad = False
        return [p for p in model.parameters() if p.requires_grad and id(p) not in dead], lr, 10, "sum"
    if stage == 2:
        head = model.fcos.head
        return list(head.iou_scores.parameters()) + list(head.mix_fc.parameters()), lr / 100, None, "loss_iou"
    if stage == 3:
        return [p for p in model.parameters() if id(p) not in dead], lr / 10000, None, "sum"
    raise ValueError("stage must be 1, 2 or 3")


def select_loss(loss_dict, which):
    return loss_dict["loss_iou"] if which == "loss_iou" else DF.loss_total(loss_dict)                 # main.py:222-225


def to_device(batch, device):
    """collate_data's 8-tuple (names first) -> the model's 7 arguments on `device`."""
    names, pse, feats, gt, tok, qlen, nprops, nframes = batch
    mv = lambda t: t.to(device, non_blocking=True)
    return names, (mv(tok), mv(qlen), mv(feats), mv(pse), mv(gt), nprops, nframes)


class _StepSlot(object):
    """Static device inputs + the captured hipGraph of one training-step geometry (clips, proposals, feature dim, padded query
    length)."""

    def __init__(self, key, device, pse_dtype, gt_dtype, feat_dtype=torch.float32):
        B, T, D, Lq = key
        self.key = key
        self.tok = torch.zeros((B, Lq), dtype=torch.int64, device=device)
        self.qlen = torch.zeros((B,), dtype=torch.int64, device=device)
        self.feats = torch.zeros((B, T, D), dtype=feat_dtype, device=device)
        self.pse = torch.zeros((B, T, 2), dtype=pse_dtype, device=device)
        self.gt = torch.zeros((B, 2), dtype=gt_dtype, device=device)
        self.args = (self.tok, self.qlen, self.feats, self.pse, self.gt, None, None)
        self.seen, self.graph, self.out, self.gen, self.alt = 0, None, None, None, 0
        self.forked, self.forked_warm = None, False      # drn_amd.graph.ForkedStep of this geometry (one process), and whether it ran eagerly once

    def load(self, tok, qlen, feats, pse, gt):
        """Copy one batch (host or device tensors) into the static buffers; the token matrix is zero-padded (padding_idx 0)
        out to the slot's query length -- the kernels read the true lengths from `qlen` on the device."""
        L = tok.shape[1]
        srcs, dsts = (tok, qlen, feats, pse, gt), (self.tok, self.qlen, self.feats, self.pse, self.gt)
        if all(t.is_cuda and t.device == d.device and t.dtype == d.dtype for t, d in zip(srcs, dsts)) and \
                all(t.is_contiguous() for t in srcs[1:]) and tok.stride(-1) == 1:
            # a device-resident batch: all five buffers (token padding included) in ONE launch instead of five or six framework
            # copies of 5-8 us each between two replays (drn_copy_multi)
            from . import ops
            ops.copy_multi(list(zip(dsts, srcs)))
            return
        self.tok[:, :L].copy_(tok, non_blocking=True)
        if L < self.tok.shape[1]:
            self.tok[:, L:].zero_()
        self.qlen.copy_(qlen, non_blocking=True)
        self.feats.copy_(feats, non_blocking=True)
        self.pse.copy_(pse, non_blocking=True)
        self.gt.copy_(gt, non_blocking=True)


class _Preloaded(object):
    """A batch whose tensors are already (being) copied into a step slot's input buffers on the copy stream."""
    __slots__ = ("slot", "event", "batch_size")

    def __init__(self, slot, event, batch_size):
        self.slot, self.event, self.batch_size = slot, event, batch_size


class Trainer(object):
    def __init__(self, model, stage, lr=1e-3, clip_gradient=0.5, world_size=1, fused=True, graph=False, lq_bucket=4,
                 graph_warmup=2, max_graphs=32, prefetch=None, forked=None):
        """graph=True (fused stages only): `train_step` replays the step as a hipGraph -- the launch path bench.py measures --
        one capture per input geometry (clips, proposals, feature dim, query length rounded up to a multiple of `lq_bucket`:
        the query kernels take the true lengths from the device, so padding changes no value) -- two, used in turn, when
        `train_epoch` is fed from the host: while one replays, the other one's input buffers are filled from the next batch on a
        copy stream (`prefetch`), so the H2D copy of the features overlaps the previous step.  The first `graph_warmup` steps of each capture, geometries beyond `max_graphs` captures and
        everything in stage 2 run eagerly.  With
        world_size > 1 forward + backward replay and the gradient exchange + optimizer run eagerly after them."""
        self.model, self.stage, self.clip = model, stage, clip_gradient
        self.params, self.lr, self.default_epochs, self.which = stage_plan(model, stage, lr)
        self.device = next(model.parameters()).device
        if world_size > 1:
            # one process per GPU: every rank must start from rank 0's replica (parameters and BatchNorm buffers) -- what
            # nn.DataParallel's per-forward broadcast guarantees in the reference (main.py:99).  Ranks that built / resumed their
            # model differently would otherwise train `world` different models on averaged gradients.
            from .dist import sync_model_state
            sync_model_state(model, src=0)
        self.fused = fused and stage != 2 and self.device.type == "cuda"
        self.world_size = world_size
        self.graph = bool(graph) and self.fused
        self.lq_bucket, self.graph_warmup, self.max_graphs = max(int(lq_bucket), 1), max(int(graph_warmup), 1), max_graphs
        self._slots = {}
        self._pools = {}                                   # 0 / 1 (the alternate slots) -> graph memory pool shared by its captures
        self._eager_geos = set()                           # geometries that fell back to eager launches (logged once each)
        self._turn = {}                                    # geometry -> uses so far (its two slots alternate)
        self._copy_stream = None
        self.prefetch = (os.environ.get("DRN_TRAINER_PREFETCH", "1") != "0") if prefetch is None else bool(prefetch)
        # every step of a graph-mode trainer -- eager ones included -- runs on ONE side stream: autograd's AccumulateGrad nodes
        # remember the stream they were created on, and warm-up, capture and replay must agree on it (drn_amd/graph.py)
        self.stream = torch.cuda.Stream(device=self.device) if self.graph else None
        # forked=True (one process): the captured step is ONE hipGraph with two branches (drn_amd.graph.ForkedStep: query side
        # beside input preparation / weight gradients; the same launches and bits as the linear capture).  Worth 3.5 % in bench.py's
        # back-to-back replays, NOTHING inside this loop (2.360 vs 2.369 ms/step at T = 256, scripts/experiments/trainer_forked_probe.py:
        # the per-step input copies and stream hand-offs around the replay sit where the overlap was) -- so it is opt-in
        if forked is None:
            forked = os.environ.get("DRN_TRAINER_FORKED", "0") == "1"
        self.forked = bool(forked) and self.graph and world_size == 1 and hasattr(model, "forward_trunk")
        self._side = torch.cuda.Stream(device=self.device) if self.forked else None
        # graph mode: the epoch's loss sum (train_epoch's return value) is accumulated ON THE DEVICE by one add that is part of
        # every step -- captured with it -- instead of two framework launches on the step's stream after every replay
        # (the eager modes add the same way after each step, so every mode returns the same bits)
        self._loss_acc = torch.zeros((), dtype=torch.float32, device=self.device)
        if self.fused:
            from .dist import GradReducer
            from .optim import FusedAdam
            # (hipGraph + several ranks: collectives stay outside the captured region -> no launches from backward hooks)
            self.reducer = GradReducer(self.params, world_size=world_size, adjacent=model.grad_stack_groups(),
                                       overlap=not (self.graph and world_size > 1),
                                       **({"bucket_bytes": 1 << 30} if world_size == 1 else {}))   # one process: one bucket
            self.opt = FusedAdam(self.reducer, lr=self.lr, max_norm=clip_gradient if clip_gradient is not None else 0.0)
        else:
            DF.unregister_grad_sinks(model.parameters())   # autograd owns this model's gradients (other models keep their sinks)
            self.reducer = None
            self.opt = torch.optim.Adam(self.params, self.lr)
            self.opt.zero_grad()

    # ------------------------------------------------------------------------------------------ hipGraph mode
    def _slot_for(self, tok, feats, pse, gt, alternate=False):
        """The slot the NEXT step of this geometry runs from, or None: eager step.  alternate (the look-ahead of train_epoch): the
        geometry's two slots take turns, so that one can be filled while the other replays."""
        B, T, D = feats.shape
        Lq = -(-int(tok.shape[1]) // self.lq_bucket) * self.lq_bucket
        geo = (int(B), int(T), int(D), Lq)
        if feats.dtype != torch.float32:
            geo = geo + (feats.dtype,)        # features handed over in the compute dtype: their own captures (no cast in the step)
        turn = self._turn.get(geo, 0)
        key = geo + ((turn % 2) if alternate else 0,)
        slot = self._slots.get(key)
        if slot is None:
            if len(self._slots) >= self.max_graphs:
                if geo not in self._eager_geos:
                    self._eager_geos.add(geo)
                    import warnings
                    warnings.warn("drn_amd.Trainer: %d hipGraph captures exist (max_graphs); steps of geometry B=%d T=%d D=%d Lq=%d "
                                  "run as eager launches" % ((len(self._slots),) + geo[:4]))
                return None
            slot = self._slots[key] = _StepSlot(geo[:4], self.device, pse.dtype, gt.dtype, feats.dtype)
            slot.alt = key[-1]
        if pse.dtype != slot.pse.dtype or gt.dtype != slot.gt.dtype:
            return None
        if alternate:
            self._turn[geo] = turn + 1
        return slot

    def _fwd_bwd(self, args):
        self.reducer.zero()
        _, loss_dict = self.model(*args)
        loss = select_loss(loss_dict, self.which)
        DF.backward(loss)
        self.reducer.collect()
        self._accumulate(loss, args[2].size(0))
        return loss_dict

    def _accumulate(self, loss, batch_size):
        self._loss_acc.add_(loss.detach().reshape(-1)[0].float(), alpha=float(batch_size))

    def _exchange_and_update(self):
        self.reducer.finish()
        self.opt.step()

    def _graph_step(self, args):
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)                       # inputs produced on the caller's stream
        whole = self.world_size == 1                       # one process: the optimizer replays with the rest
        pre = args if isinstance(args, _Preloaded) else None
        if pre is not None and pre.event is not None:
            self.stream.wait_event(pre.event)              # the copy stream has filled the slot's input buffers
        with torch.cuda.stream(self.stream):
            if pre is not None:
                slot = pre.slot
            else:
                tok, qlen, feats, pse, gt = args[:5]
                slot = self._slot_for(tok, feats, pse, gt)
            if slot is None:
                dev_args = tuple(a.to(self.device, non_blocking=True) if torch.is_tensor(a) else a for a in args)
                out = self._fwd_bwd(dev_args)
                self._exchange_and_update()
            else:
                if pre is None:
                    slot.load(tok, qlen, feats, pse, gt)
                stores = getattr(self.opt, "stores", None)
                if slot.graph is not None and whole and DF.cache_generation(stores) != slot.gen:
                    # A re-laid weight copy appeared (or went) after this capture -- an eval-only path, another perm / dtype: the
                    # captured optimizer nodes do not maintain it and no Python runs in a replay to invalidate it.  Invalidate
                    # every copy now and re-capture after one eager step (which rebuilds the optimizer's tables).
                    DF.bump_weights_epoch(stores)
                    for s_ in self._slots.values():
                        if s_.graph is not None:
                            s_.graph, s_.seen = None, self.graph_warmup - 1
                if slot.graph is None and slot.seen >= self.graph_warmup:
                    if self.world_size > 1:
                        self.reducer.rearm()
                    # thread_local: a hipHostMalloc from the DataLoader's pin_memory thread during this capture must not fail it
                    # (global mode turns ANY thread's unsafe call into hipErrorStreamCaptureUnsupported).  Captures share one
                    # memory pool (they replay one at a time); the alternate slots of the H2D look-ahead share a second one.
                    alt = slot.alt
                    kw = {"pool": self._pools[alt]} if alt in self._pools else {}
                    if self.forked and whole:
                        from .graph import ForkedStep
                        fs = slot.forked
                        if fs is None:
                            # (the loss selector must not close over `self`: trainer -> slot -> step -> closure -> trainer would be a
                            # reference cycle, and with it the trainer's graphs and streams would die whenever the GC gets to them)
                            fs = slot.forked = ForkedStep(self.model, slot.args[:5], functools.partial(select_loss, which=self.which),
                                                          self.reducer, self.opt, main=self.stream, side=self._side)
                        if not slot.forked_warm:
                            # this step runs the two-branch schedule EAGERLY once (workspaces / counters of the side stream come
                            # into being outside a capture); the next one captures it
                            slot.forked_warm = True
                            fs._fresh = True
                            fs._schedule(fs._phase, fs.main, fs.side)
                            fs.main.wait_stream(fs.side)
                            self._accumulate(select_loss(fs.out, self.which), slot.feats.size(0))
                            slot.seen += 1
                            cur.wait_stream(self.stream)
                            return fs.out
                        g = fs._capture_once(kw.get("pool"))
                        slot.out = fs.out
                        slot.acc_outside = True              # (the two-branch capture does not contain the epoch-loss add)
                    else:
                        g = torch.cuda.CUDAGraph()
                        from .graph import capture_graph          # (thread_local capture, cyclic GC off while it is open)
                        with capture_graph(g, self.stream, **kw):
                            slot.out = self._fwd_bwd(slot.args)
                            if whole:
                                self._exchange_and_update()
                    self._pools.setdefault(alt, g.pool())
                    slot.graph, slot.gen = g, DF.cache_generation(stores)
                if slot.graph is not None:
                    if self.world_size > 1:
                        self.reducer.rearm()               # hooks only ran at capture time
                    slot.graph.replay()
                    if getattr(slot, "acc_outside", False):
                        self._accumulate(select_loss(slot.out, self.which), slot.feats.size(0))
                    if not whole:
                        self._exchange_and_update()
                    out = slot.out
                else:
                    slot.seen += 1
                    out = self._fwd_bwd(slot.args)
                    self._exchange_and_update()
        cur.wait_stream(self.stream)
        return out

    def train_step(self, args):
        """One main.py:218-243 iteration.  Returns the loss dict (device tensors, no sync).  Arguments: the model's 7
        positional inputs, device-resident (graph mode also takes host tensors: they are copied straight into the static
        input buffers of the captured step; the returned losses are then views of replay-owned memory -- read them before the
        next step)."""
        self.model.train()
        if self.graph:
            return self._graph_step(args)
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
        n = 0
        if getattr(self, "_loss_acc", None) is None:       # (a trainer built without __init__: tests)
            self._loss_acc = torch.zeros((), dtype=torch.float32, device=getattr(self, "device", "cpu"))
        if getattr(self, "graph", False) and getattr(self, "prefetch", False):
            # One batch of look-ahead: batch i+1 is copied into the OTHER slot of its geometry on a copy stream before step i is
            # launched, so the fp32 features (B x T x D x 4 bytes: 134 MB at the benchmarked shape, 2.4 ms of PCIe -- longer than
            # the step) cross the bus while the previous step computes and the step's own stream carries only the replay.
            it = iter(loader)
            nxt = self._prefetch(next(it, None))
            with torch.cuda.stream(self.stream):
                self._loss_acc.zero_()                     # (every step adds its loss x batch size itself: _accumulate)
            while nxt is not None:
                cur_b = nxt
                nxt = self._prefetch(next(it, None))
                bs = cur_b.batch_size if isinstance(cur_b, _Preloaded) else cur_b[2].size(0)
                self.train_step(cur_b)
                n += bs
            torch.cuda.current_stream().wait_stream(self.stream)
            return self._epoch_result(n)
        with torch.cuda.stream(self.stream) if self.graph else contextlib.nullcontext():
            self._loss_acc.zero_()
        for batch in loader:
            if self.graph:                                 # host tensors go straight into the captured step's input buffers
                names, pse, feats, gt, tok, qlen, nprops, nframes = batch
                args = (tok, qlen, feats, pse, gt, nprops, nframes)
            else:
                _, args = to_device(batch, self.device)
            bs = args[2].size(0)
            ld = self.train_step(args)
            if not self.graph:
                self._accumulate(select_loss(ld, self.which), bs)
            n += bs
        if self.graph:
            torch.cuda.current_stream().wait_stream(self.stream)
        return self._epoch_result(n)

    def _epoch_result(self, n):
        """Mean loss of the epoch (the one host sync) -- and, on the device, the in-launch exchanges' watchdog counters: a workgroup
        that gave up waiting lets invalid values through and only counts it (drn_amd.ops.check_watchdogs raises)."""
        mean = float(self._loss_acc) / max(n, 1) if n else 0.0
        if getattr(getattr(self, "device", None), "type", "cpu") == "cuda":
            from . import ops
            ops.check_watchdogs()
        return mean

    def _prefetch(self, batch):
        """graph mode: start copying a batch (host or device tensors) into the input buffers of the slot its step will run from,
        on the copy stream.  -> _Preloaded, or the plain argument tuple when the step has to run eagerly (no slot left)."""
        if batch is None:
            return None
        names, pse, feats, gt, tok, qlen, nprops, nframes = batch
        if all(t.is_cuda for t in (tok, qlen, feats, pse, gt)):
            # already on the device: the step copies it into its input buffers itself (a second stream beside the replay only
            # slows the latency-bound kernels: 1.43 vs 1.40 ms/step at T = 32)
            return (tok, qlen, feats, pse, gt, nprops, nframes)
        slot = self._slot_for(tok, feats, pse, gt, alternate=True)
        if slot is None:
            return (tok, qlen, feats, pse, gt, nprops, nframes)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        cs = self._copy_stream
        # behind everything queued so far: the previous use of this slot (two steps back) and a batch produced on the caller's stream
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            slot.load(tok, qlen, feats, pse, gt)
            ev = torch.cuda.Event()
            ev.record(cs)
        for t in (tok, qlen, feats, pse, gt):
            if t.is_cuda:
                t.record_stream(cs)                        # (a device-resident part of a mixed batch must outlive the copy)
        return _Preloaded(slot, ev, int(feats.size(0)))

    @torch.no_grad()
    def evaluate(self, loader, id2word=None, iou_topk=None, with_results=True, group=None):
        """main.py:270-366: returns (mean loss, topks, accuracies, raw results dict).
        with_results=False (what fit() uses): the raw-results records are not built -- Recall@k with temporal NMS is computed
        on the device from the post-processor's buffers (drn_eval_recall), one small copy at the end of the pass and no host
        synchronisation per batch; the fourth return value is then None.
        One process per GPU: every rank evaluates ITS shard of the test set (train.py builds the loader over
        drn_amd.data.ShardSampler) and the ranks' counts (or records) are merged with all_gather_object, so all ranks return
        the same numbers (`group`: the process group, default world; a single process skips the exchange)."""
        self.model.eval()
        import torch.distributed as td
        multi = td.is_available() and td.is_initialized() and td.get_world_size(group) > 1
        if multi:
            # every rank scores its shard with ITS replica: the parameters are equal by construction (same start, same averaged
            # gradients), the BatchNorm running statistics are not (per-rank batches) -- take rank 0's, the replica whose
            # state_dict fit() saves under the merged metric (nn.DataParallel keeps replica 0's buffers, main.py:99)
            from .dist import sync_model_state
            sync_model_state(self.model, src=0, group=group, buffers_only=True)
        iou_topk = iou_topk or {"iou": [0.5], "topk": [1, 5]}                            # main.py:362
        results, total, n, hits = {}, None, 0, []
        selector = getattr(getattr(self.model, "fcos", None), "box_selector_test", None)
        head = getattr(getattr(self.model, "fcos", None), "head", None)
        one_class = head is None or head.cls_logits.weight.shape[0] == 1       # (more foreground channels: the records path)
        fast = (not with_results) and selector is not None and self.device.type == "cuda" and one_class
        # (decided ONCE, from static facts, so every rank takes the same branch of the merge below; drn_eval_recall serves every
        # level table drn_postprocess accepts, so the fast path has no per-batch fallback to disagree about)
        fast0 = fast
        ious_dev = torch.tensor([float(x) for x in iou_topk["iou"]], dtype=torch.float64, device=self.device) if fast else None
        if fast:
            selector.device_only = True
        try:
            for batch in loader:
                # the words, lengths and ground truth of the records come from the HOST batch (no copy back); the loss is summed on
                # the device (one sync at the end); detections / scores of all clips cross in one copy each (metrics.results_entries)
                host_tok, host_qlen, host_gt = batch[4], batch[5], batch[3]
                names, args = to_device(batch, self.device)
                boxes, loss_dict = self.model(*args)
                bs = args[2].size(0)
                lsum = select_loss(loss_dict, self.which).detach().reshape(-1)[0].float() * bs
                total = lsum if total is None else total + lsum
                n += bs
                if fast and not isinstance(boxes, list):
                    from . import ops
                    hits.append(ops.eval_recall(boxes.det, boxes.scores, boxes.counts, args[4].contiguous(), ious_dev,
                                                max(iou_topk["topk"])))
                    continue
                if fast:
                    raise RuntimeError("Trainer.evaluate: the post-processor returned host records in device-only mode")
                tokens, qlen, gts = host_tok.cpu(), host_qlen.cpu(), host_gt.cpu().numpy()
                queries = [" ".join(id2word[int(t)] if id2word else str(int(t)) for t in tokens[i, :int(qlen[i])]) for i in range(bs)]
                for name, entry in zip(names, results_entries(queries, gts, boxes)):
                    results.setdefault(name, []).append(entry)
        finally:
            if fast:
                selector.device_only = False
        total = float(total) if total is not None else 0.0
        if fast0:                                          # (the branch every rank decided on before its first batch)
            fh = torch.cat(hits).cpu().numpy() if hits else np.zeros((0, len(iou_topk["iou"])), dtype=np.int32)
            if multi:
                parts = [None] * td.get_world_size(group)
                td.all_gather_object(parts, ("hits", fh, total, n), group=group)
                assert all(p[0] == "hits" for p in parts), "ranks disagree on the evaluation path"
                fh = np.concatenate([p[1] for p in parts])
                total, n = sum(p[2] for p in parts), sum(p[3] for p in parts)
            return total / max(n, 1), iou_topk["topk"], recall_from_first_hits(fh, iou_topk["iou"], iou_topk["topk"]), None
        if multi:
            parts = [None] * td.get_world_size(group)
            td.all_gather_object(parts, ("records", results, total, n), group=group)
            assert all(p[0] == "records" for p in parts), "ranks disagree on the evaluation path"
            parts = [p[1:] for p in parts]
            results = {}
            for part, _, _ in parts:                                   # rank order: deterministic, and the metric does not depend on it
                for vid, items in part.items():
                    results.setdefault(vid, []).extend(items)
            total, n = sum(p[1] for p in parts), sum(p[2] for p in parts)
        topks, accs = PostProcessRunner(results).run_evaluate(iou_topk_dict=iou_topk, temporal_nms=True)
        return total / max(n, 1), topks, accs, results

    def fit(self, train_loader, test_loader, n_epoch=None, eval_freq=1, snapshot_pref=None, dataset="Charades", id2word=None,
            start_epoch=0, rank=0):
        """main.py:142-190: train, validate every eval_freq epochs, keep the best-R@1 and best-R@5 checkpoints.
        One process per GPU: EVERY rank calls fit() -- training steps and the sharded evaluation are collective -- and rank 0
        alone writes checkpoints (`rank`)."""
        n_epoch = self.default_epochs if self.default_epochs is not None else n_epoch
        best1 = best5 = 0.0
        history = []
        for epoch in range(start_epoch, n_epoch):
            train_loss = self.train_epoch(train_loader, epoch)
            rec = {"epoch": epoch, "train_loss": train_loss}
            if (epoch + 1) % eval_freq == 0 or epoch == n_epoch - 1:
                val_loss, topks, accs, _ = self.evaluate(test_loader, id2word, with_results=False)
                top1, top5 = accs[0] * 100, accs[1] * 100
                rec.update(val_loss=val_loss, top1=top1, top5=top5)
                state = {"epoch": epoch + 1, "state_dict": checkpoint_state_dict(self.model), "loss": val_loss, "top1": top1,
                         "top5": top5}
                if snapshot_pref is not None and rank == 0:
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
    DF.bump_weights_epoch([DF.store_of(next(model.parameters()))])
    return ckpt.get("epoch", 0), sorted(picked)


def init_glove(model, glove_weights):
    """main.py:92-94: copy the (vocab+1, 300) GloVe table into the embedding if the file exists."""
    if glove_weights and os.path.exists(glove_weights):
        model.query_encoder.embedding.weight.data.copy_(torch.load(glove_weights))
        DF.bump_weights_epoch([DF.store_of(model.query_encoder.embedding.weight)])
        return True
    return False
