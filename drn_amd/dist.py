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
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif os.environ.get("DRN_FORCE_DEVICE") is not None and torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ["DRN_FORCE_DEVICE"]))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class _Bucket(object):
    __slots__ = ("flat", "params", "pending", "handle", "launched")


class GradReducer(object):
    """Bucketed, backward-overlapped gradient averaging.

    params: iterable of parameters (only those with requires_grad are reduced).
    bucket_bytes: xGMI is point-to-point (7 links/GPU), so fewer, larger messages win; 157 MB of DRN gradients
    in 32 MB buckets = 5 collectives per step.
    """

    def __init__(self, params, world_size=None, bucket_bytes=32 << 20, group=None, overlap=True):
        """overlap=False defers every collective to finish() (required when backward is replayed from a hipGraph:
        hooks only run at capture time and collectives must stay outside the captured region)."""
        self.group = group
        self.overlap = overlap
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.params = [p for p in params if p.requires_grad]
        self.buckets, self._of = [], {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):                       # roughly the order backward produces gradients
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._make_bucket(cur)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _make_bucket(self, params):
        b = _Bucket()
        n = sum(p.numel() for p in params)
        b.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            p.grad = b.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self._of[p] = b
        b.params, b.pending, b.handle, b.launched = params, len(params), None, False
        self.buckets.append(b)

    def _launch(self, b):
        b.launched = True
        if self.world > 1:
            b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        b = self._of[p]
        b.pending -= 1
        if self.overlap and b.pending == 0 and not b.launched:
            self._launch(b)

    def zero(self):
        """Call before each backward: zero the flat buffers (p.grad stay views) and re-arm the buckets."""
        for b in self.buckets:
            b.flat.zero_()
            b.pending, b.handle, b.launched = len(b.params), None, False
            off = 0
            for p in b.params:                                # something may have replaced p.grad (e.g. set_to_none)
                if p.grad is None or p.grad.data_ptr() != b.flat.data_ptr() + off * b.flat.element_size():
                    p.grad = b.flat[off:off + p.numel()].view_as(p)
                off += p.numel()

    def finish(self):
        """Call after backward: reduce the buckets whose parameters got no gradient this step (unused / frozen
        branches contribute zeros), wait for all collectives and turn sums into means."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        if self.world > 1:
            for b in self.buckets:
                b.handle.wait()
                b.flat.div_(self.world)

    def remove(self):
        for h in self._hooks:
            h.remove()
