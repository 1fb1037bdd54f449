"""Fused clip_grad_norm_ + Adam on the flat gradient buckets of drn_amd.dist.GradReducer (HIP kernels in
drn_amd/csrc/optim.hip).  Same arithmetic as `clip_grad_norm_(params, max_norm); torch.optim.Adam.step()`
(main.py:140,238-243; weight_decay is configured but never used by the reference)."""
import ctypes

import torch

from . import functional as DF
from ._lib import check, lib


class FusedAdam(object):
    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=0.5):
        self.reducer, self.lr, self.betas, self.eps, self.max_norm = reducer, lr, betas, eps, max_norm
        L = lib()
        L.drn_opt_nblocks.restype = ctypes.c_int64
        dev = reducer.buckets[0].flat.device
        self.state = []
        nparts = 0
        for b in reducer.buckets:
            n = b.flat.numel()
            offs, ptrs = [], []
            for p, off in zip(b.params, b.offsets):
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous fp32 parameters")
                offs.append(off)
                ptrs.append(p.data_ptr())
                pad_start = off + p.numel()
                if pad_start % 4:                             # alignment padding: a segment with a null pointer
                    offs.append(pad_start)
                    ptrs.append(0)
            offs.append(n)
            nb = int(L.drn_opt_nblocks(ctypes.c_int64(n)))
            import numpy as np
            starts = np.asarray(offs[:-1], dtype=np.int64)
            blk_seg = (np.searchsorted(starts, np.arange(nb, dtype=np.int64) * 4096, side="right") - 1).astype(np.int32)
            self.state.append({"m": torch.zeros_like(b.flat), "v": torch.zeros_like(b.flat),
                               "seg": torch.tensor(offs, dtype=torch.int64, device=dev),
                               "ptr": torch.tensor(ptrs, dtype=torch.int64, device=dev),
                               "nseg": len(ptrs), "part_off": nparts, "nb": nb,
                               "blk_seg": torch.from_numpy(blk_seg).to(dev)})
            nparts += nb
        self._mirror_sig, self._mirror_keys = None, ()
        self.partials = torch.zeros(nparts, dtype=torch.float32, device=dev)
        self.total_sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)

        self._ptr_sig = tuple(p.data_ptr() for b in reducer.buckets for p in b.params)

    def _check_ptrs(self):
        """The kernels reach the parameters through raw pointers baked into device tables: refuse to step if a parameter's
        storage moved since construction (model.to(), .float(), load_state_dict(assign=True) ...).  Host-only comparison."""
        if tuple(p.data_ptr() for b in self.reducer.buckets for p in b.params) != self._ptr_sig:
            raise RuntimeError("parameter storage moved after FusedAdam was built; build a new GradReducer + FusedAdam")

    def step(self, repack=True):
        """repack=False leaves the refresh of the re-laid weight copies to the caller (`repack(codes)`), who may split it
        over two points of its schedule (drn_amd.graph.DualStreamStep)."""
        self._check_ptrs()
        L = lib()
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        for i, (b, st) in enumerate(zip(self.reducer.buckets, self.state)):
            part = self.partials[st["part_off"]:]
            check(L.drn_sumsq_partials(P(b.flat), ctypes.c_int64(b.flat.numel()), P(part),
                                       P(self.step_counter) if i == 0 else None, s), "drn_sumsq_partials")
        check(L.drn_sumsq_finalize(P(self.partials), self.partials.numel(), P(self.total_sumsq), s), "drn_sumsq_finalize")
        self._refresh_mirrors()
        for b, st in zip(self.reducer.buckets, self.state):
            check(L.drn_adam_bucket(P(b.flat), P(st["m"]), P(st["v"]), ctypes.c_int64(b.flat.numel()), P(st["seg"]), P(st["ptr"]),
                                    st["nseg"], P(st["blk_seg"]), P(st.get("mirror")), P(self.total_sumsq), P(self.step_counter),
                                    ctypes.c_float(self.lr), ctypes.c_float(self.betas[0]), ctypes.c_float(self.betas[1]),
                                    ctypes.c_float(self.eps), ctypes.c_float(self.max_norm), s), "drn_adam_bucket")
        DF.bump_weights_epoch()       # parameters changed behind autograd's version counters
        if repack:
            self.repack()

    def repack(self, codes=None):
        """Refresh the other GEMM-layout copies of the weights (one launch per dtype; `codes`: only these dtypes)."""
        DF.repack_all(skip=self._mirror_keys, codes=codes)

    def _refresh_mirrors(self):
        """Device tables of the bf16 GEMM operands that keep their parameter's element order (drn_adam_bucket rewrites them
        in place).  Rebuilt only when the set of cached copies changes, and never while a hipGraph is being captured (the
        table upload is a host->device copy): copies that appear later are simply left to repack_all()."""
        copies = DF.identity_bf16_copies()
        sig = tuple(sorted((ptr, buf.data_ptr()) for ptr, (key, buf) in copies.items()))
        if sig == self._mirror_sig or torch.cuda.is_current_stream_capturing():
            return
        keys = []
        for st in self.state:
            tab = []
            for ptr in st["ptr"].tolist():
                hit = copies.get(ptr) if ptr else None
                tab.append(hit[1].data_ptr() if hit else 0)
                if hit:
                    keys.append(hit[0])
            st["mirror_bufs"] = [copies[p][1] for p in st["ptr"].tolist() if p and p in copies]      # keep them alive
            st["mirror"] = torch.tensor(tab, dtype=torch.int64, device=st["ptr"].device) if any(tab) else None
        self._mirror_sig, self._mirror_keys = sig, frozenset(keys)

    def total_norm(self):
        return self.total_sumsq[0].sqrt()
