"""Data parallelism for the DRN path: one process per GPU, gradients all-reduced over RCCL/xGMI.

The reference only wraps the model in single-process nn.DataParallel (main.py:99); this is the MI355X-native
replacement: per-rank BatchNorm statistics and loss normalisation (what DataParallel replicas do), then ONE
exchange step per iteration -- a bucketed all-reduce(sum)/world of the parameter gradients, launched from
post-accumulate-grad hooks so buckets filled early in backward overlap the rest of backward.  Gradients live
in flat per-bucket buffers (`p.grad` are views), so there is no pack/unpack copy around the collective.
Backend "nccl" is RCCL on ROCm; "gloo" works on CPU tensors (tests/test_dist_cpu.py, world_size 2).
"""
import os

import torch
import torch.distributed as dist

from . import functional as DF


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*). Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1"):
            # one node: keep gloo on the loopback interface instead of resolving the host name, which may not resolve
            # inside a container (a rendezvous that then waits for its 30-minute timeout).  RCCL picks its own interface.
            if backend == "gloo":
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif os.environ.get("DRN_FORCE_DEVICE") is not None and torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ["DRN_FORCE_DEVICE"]))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


WGRAD_DEFER = os.environ.get("DRN_WGRAD_DEFER", "1") != "0"      # (experiment switch: 0 = every weight gradient reduces in its own launch)


def _initialized(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def broadcast_tensors(tensors, src=0, group=None):
    """In-place broadcast of `tensors` from rank `src`: one flat message per dtype (few, large messages: xGMI links are
    point-to-point), copied back into the tensors' own storage."""
    if not _initialized(group):
        return 0
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    n = 0
    with torch.no_grad():
        for dt, ts in by_dtype.items():
            flat = torch._utils._flatten_dense_tensors([t.detach().contiguous() for t in ts])
            dist.broadcast(flat, src=src, group=group)
            for t, r in zip(ts, torch._utils._unflatten_dense_tensors(flat, ts)):
                t.detach().copy_(r)
            n += flat.numel()
    return n


def sync_model_state(model, src=0, group=None, buffers_only=False):
    """Make every rank hold rank `src`'s replica: parameters AND buffers (BatchNorm running statistics / step counters).  The
    reference's nn.DataParallel (main.py:99) broadcasts the one model's parameters and buffers to its replicas every forward;
    one process per GPU has to do it explicitly -- once at start-up (ranks may have initialised / resumed differently) and, for
    the buffers, before every sharded evaluation (per-rank BatchNorm running statistics drift apart during training: each rank
    normalises its own shard, exactly like DataParallel's replicas, whose statistics are thrown away except replica 0's).
    The re-laid GEMM copies of the parameters are invalidated when parameters were overwritten."""
    if not _initialized(group):
        return 0
    DF.flush_bn_counters()
    ts = [] if buffers_only else [p for p in model.parameters()]
    ts += [b for b in model.buffers()]
    n = broadcast_tensors(ts, src=src, group=group)
    if not buffers_only:
        stores = set()
        for p in model.parameters():
            stores.add(DF.store_of(p))
        DF.bump_weights_epoch(list(stores))
    return n


class _Bucket(object):
    __slots__ = ("flat", "params", "offsets", "views", "pending", "handle", "launched")


class GradReducer(object):
    """Bucketed, backward-overlapped gradient averaging.

    params: iterable of parameters (only those with requires_grad are reduced).
    bucket_bytes: xGMI is point-to-point (7 links/GPU), so fewer, larger messages win; 157 MB of DRN gradients
    in 32 MB buckets = 5 collectives per step.
    """

    def __init__(self, params, world_size=None, bucket_bytes=32 << 20, group=None, overlap=True, steal=True, groups=None,
                 adjacent=None):
        """overlap=False defers every collective to finish() (required when backward is replayed from a hipGraph:
        hooks only run at capture time and collectives must stay outside the captured region).
        groups: optional list of parameter lists, each bucketed on its own and in the given order (`group_buckets[i]`
        lists group i's buckets): lets a caller reduce one part of the model while another part is still in backward.
        adjacent: optional list of parameter lists that must lie back to back, in that order, inside one bucket (the
        sources of a stacked GEMM operand, `mainModel.grad_stack_groups()`): the stacked gradient is then one slice of the
        flat buffer and is written there directly (functional.grad_buffer)."""
        self.group = group
        self.overlap = overlap
        # True (set by drn_amd.optim.FusedAdam): wait() leaves the all-reduced SUM in the buckets -- `p.grad` then reads
        # world x the mean -- and the optimizer kernels apply 1/world themselves (`grad_scale`)
        self.defer_average = False
        # steal=True: p.grad is None at the start of backward; backward kernels that know the sink write straight into
        # the flat bucket and autograd adopts that view (no zero-fill, no accumulate-add); anything else is copied in.
        self.steal = steal
        self._dirty = {}
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.params = [p for p in params if p.requires_grad]
        # (RCCL's kernels run beside backward: the GEMM kernels' in-launch split-K exchanges confirm their stores -- they always do,
        # drn_amd.ops._ksplit_arg)
        self._pend = None                  # ops.WgradPending over this reducer's buckets: the deferred weight-gradient reduces
        self.buckets, self._of, self.group_buckets = [], {}, []
        wanted = set(id(p) for p in self.params)
        plan = [self.params] if groups is None else [[p for p in g if id(p) in wanted] for g in groups]
        assert sum(len(g) for g in plan) == len(self.params), "groups must partition the trainable parameters"
        adj_of = {}
        for grp in (adjacent or []):
            grp = [p for p in grp if id(p) in wanted]
            # back to back only when no 4-element alignment padding falls between them; one-element parameters are kept
            # together too (equal 4-element spacing: functional.grad_buffer hands out a strided view)
            if len(grp) > 1 and (all(self._packs_tight(p, q) for p, q in zip(grp, grp[1:])) or all(p.numel() == 1 for p in grp)):
                for p in grp:
                    adj_of[id(p)] = grp
            elif len(grp) > 1:
                import warnings
                warnings.warn("GradReducer: adjacent group of sizes %s cannot lie back to back (alignment padding between its "
                              "members); its stacked gradient takes the copy path" % [p.numel() for p in grp])
        for part in plan:
            first = len(self.buckets)
            cur, cur_bytes = [], 0
            in_part, placed = set(id(p) for p in part), set()
            for p in reversed(part):                          # roughly the order backward produces gradients
                if id(p) in placed:
                    continue
                grp = adj_of.get(id(p))
                run = grp if grp is not None and all(id(q) in in_part for q in grp) else [p]
                for q in run:
                    placed.add(id(q))
                    cur.append(q)
                    cur_bytes += q.numel() * q.element_size()
                if cur_bytes >= bucket_bytes:
                    self._make_bucket(cur)
                    cur, cur_bytes = [], 0
            if cur:
                self._make_bucket(cur)
            self.group_buckets.append(self.buckets[first:])
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    @staticmethod
    def _packs_tight(p, q):
        """No alignment padding falls between `p` and `q` placed right behind it (see _offsets): a successor of >= 1024 elements
        starts on a 32-element boundary, so `p` must start on one too (>= 1024 elements itself) and be a multiple of 32 long; a
        smaller successor only needs `p` to be a multiple of 4 long."""
        if q.numel() >= 1024:
            return p.numel() >= 1024 and p.numel() % 32 == 0
        return p.numel() % 4 == 0

    @staticmethod
    def _offsets(params):
        """Start of every tensor inside the flat buffer: 4-element (16-byte) aligned for vector access in the fused optimizer, and
        32-element (128-byte, one cache line) aligned from 1024 elements up -- the optimizer walks gradient, m and v in 256-byte row
        segments of 64-channel tiles, which a start in the middle of a line spreads over three lines instead of two (round 4:
        every conv / FPN weight sat at offset % 32 = 24 behind the three one-element Scale parameters).  Tensors that must lie back
        to back (`adjacent=`) still do when their sizes are multiples of 32, which DRN's stacked tower weights are."""
        offs, off = [], 0
        for p in params:
            if p.numel() >= 1024:
                off = (off + 31) // 32 * 32
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        return offs, off

    def _make_bucket(self, params):
        b = _Bucket()
        offs, n = self._offsets(params)
        b.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        b.views = []
        for p, off in zip(params, offs):
            v = b.flat[off:off + p.numel()]
            b.views.append(v)
            p.grad = v.view_as(p)
            self._of[p] = (b, len(b.views) - 1)
            DF.register_grad_sink(p, v)
        b.params, b.offsets, b.pending, b.handle, b.launched = params, offs, len(params), None, False
        self.buckets.append(b)

    def _launch(self, b):
        b.launched = True
        if self.world > 1:
            b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        b, i = self._of[p]
        sink = b.views[i]
        if p.grad is not None and p.grad.data_ptr() != sink.data_ptr() and self.world > 1 and self.overlap:
            with torch.no_grad():
                sink.copy_(p.grad.reshape(-1))          # gradient produced elsewhere (stock autograd ops): move it in
            p.grad = sink.view_as(p)                    # (single process / deferred mode: finish() batches these copies)
        if p.grad is not None:                          # (the engine also runs the hook of a parameter whose gradient came back undefined)
            self._dirty[p] = True
        b.pending -= 1
        if self.overlap and b.pending == 0 and not b.launched:
            self._launch(b)

    def _wgrad_defer(self, on):
        """Between zero() and collect() the conv weight-gradient launches that write into THIS reducer's buckets leave their reduce
        passes to ONE launch that collect() issues (drn_amd.ops.WgradPending: a list this reducer owns; a launch finds it by where
        its dW lives): nothing reads those slices before collect().  Only with steal=True -- otherwise AccumulateGrad adds a
        still un-reduced dW into the bucket the moment backward hands it over -- and not in the eager multi-GPU mode, whose hooks
        hand buckets to RCCL as backward fills them.  zero() drops whatever an earlier backward that raised left recorded."""
        if not self.params or not self.params[0].is_cuda or (self.world > 1 and self.overlap) or not WGRAD_DEFER or not self.steal:
            return
        from . import ops
        want = bool(getattr(self, "ext_sumsq", False)) and self.world == 1
        if self._pend is None:
            self._pend = ops.WgradPending([(b.flat.data_ptr(), b.flat.data_ptr() + b.flat.numel() * b.flat.element_size())
                                           for b in self.buckets])
        if on:
            self._pend.reset()
            ops.wgrad_arm(self._pend)
            DF.want_sumsq(want)
            self.sumsq_notes = []
        else:
            if not any(q is self._pend for q in ops._armed):
                return                                  # (collect() without a zero() before it, remove() after collect())
            ops.wgrad_disarm(self._pend)
            res = ops.wgrad_reduce_pending(self._pend, sumsq=want)
            if want:
                if res is not None:
                    DF.note_sumsq(*res)
                # (kept until the next zero(): the fused optimizer's norm pass leaves these ranges out and adds the partials)
                self.sumsq_notes = getattr(self, "sumsq_notes", []) + DF.take_sumsq_notes()
                DF.want_sumsq(False)

    def zero(self):
        """Call before each backward: re-arm the buckets.  steal mode drops p.grad (the flat slices get overwritten by
        backward); otherwise the flat buffers are zeroed and p.grad stay views that autograd accumulates into."""
        self._wgrad_defer(True)
        for b in self.buckets:
            b.pending, b.handle, b.launched = len(b.params), None, False
            if self.steal:
                for p in b.params:
                    p.grad = None
            else:
                b.flat.zero_()
                for p, v in zip(b.params, b.views):
                    if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                        p.grad = v.view_as(p)

    def adopt(self, params, grads):
        """Gradients computed outside the engine's accumulation (torch.autograd.grad on a second stream,
        drn_amd.graph.DualStreamStep): install them as p.grad the way AccumulateGrad + the hook would have; collect() then
        moves in whatever is not already its sink."""
        for p, g in zip(params, grads):
            if g is not None and p in self._of:
                p.grad = g if g.shape == p.shape else g.view_as(p)
                self._dirty[p] = True

    def collect(self, buckets=None):
        """After backward (all buckets, or the given ones): make every flat bucket hold this step's gradients -- slices of parameters that got no gradient
        read as zero, gradients produced outside the sinks (stock autograd ops, hand-set) are copied in with one
        multi-tensor launch per bucket.  Device work only, so it can sit inside a captured hipGraph: call it at the end
        of the captured forward+backward and `reduce()` after the replay."""
        self._wgrad_defer(False)
        for b in (self.buckets if buckets is None else buckets):
            if not self.steal:
                continue
            dsts, srcs = [], []
            for p, v in zip(b.params, b.views):
                if p.grad is None:                          # no gradient this step: the slice must read as zero
                    if self._dirty.get(p, False):
                        v.zero_()
                        self._dirty[p] = False
                    p.grad = v.view_as(p)
                elif p.grad.data_ptr() != v.data_ptr():     # produced by stock autograd ops / set by hand: move it in
                    dsts.append(v)
                    srcs.append(p.grad.detach().reshape(-1))
                    p.grad = v.view_as(p)
                    self._dirty[p] = True
            if dsts:
                with torch.no_grad():
                    torch._foreach_copy_(dsts, srcs)        # one multi-tensor launch per bucket

    def reduce(self, buckets=None):
        """Launch the collectives that are not in flight yet (all buckets, or the given ones) without waiting."""
        for b in (self.buckets if buckets is None else buckets):
            if not b.launched:
                self._launch(b)

    def wait(self, timings=None):
        """Wait for all collectives and turn sums into means (unless the optimizer does that: `defer_average`).
        timings: a list that receives, per bucket in bucket order, (bucket index, start event, end event) recorded on the
        current stream around that bucket's wait -- the time the step's stream is held up by each exchange (bench.py's
        per-bucket `exposed_ms`); CUDA tensors only."""
        if self.world > 1:
            for i, b in enumerate(self.buckets):
                if b.handle is not None:
                    ev = None
                    if timings is not None and b.flat.is_cuda:
                        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                        ev[0].record()
                    b.handle.wait()
                    b.handle = None
                    if ev is not None:
                        ev[1].record()
                        timings.append((i, ev[0], ev[1]))
                    if not self.defer_average:
                        b.flat.div_(self.world)

    def rearm(self):
        """Deferred (hipGraph) mode: hooks only run at capture time, so mark every bucket as not yet reduced."""
        for b in self.buckets:
            b.launched, b.handle = False, None

    def finish(self, timings=None):
        """Call after backward: collect(), reduce the buckets that are not in flight yet (those whose parameters got no
        gradient this step contribute zeros), wait for all collectives and turn sums into means.  timings: see wait()."""
        self.collect()
        self.reduce()
        self.wait(timings)

    def remove(self):
        self._wgrad_defer(False)
        for h in self._hooks:
            h.remove()
        DF.unregister_grad_sinks(self.params)          # this reducer's parameters only: other models keep their sinks
