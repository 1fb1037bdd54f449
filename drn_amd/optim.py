"""Fused clip_grad_norm_ + Adam on the flat gradient buckets of drn_amd.dist.GradReducer (HIP kernels in
drn_amd/csrc/optim.hip).  Same arithmetic as `clip_grad_norm_(params, max_norm); torch.optim.Adam.step()`
(main.py:140,238-243; weight_decay is configured but never used by the reference)."""
import ctypes

import torch

import os

from . import functional as DF
from ._lib import check, lib

# 1 = the norm pass leaves out the gradients whose producers already summed their squares (prop_fc's GEMM epilogue, the one-launch
# reduce of the conv weight gradients): the pass drops 28 -> 16 us, and the step does not move (2.051 vs 2.050-2.067 ms, in one box) --
# the plain pass was also pulling those 108 MB into the 256 MB infinity cache for the Adam kernels behind it (adam_tiled +9 us
# without it), and the producers pay 1-5 us for the sums.  Kept as a measured option, off.  (With it the two-branch step is deterministic
# but no longer bit-identical to the linear one: the one-launch reduce's partial sums follow the order the weight gradients were launched in.)
EXT_SUMSQ = os.environ.get("DRN_EXT_SUMSQ", "0") != "0"


class FusedAdam(object):
    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=0.5, tiled=True):
        """tiled=False keeps every parameter on the linear kernel and leaves the re-laid weight copies to repack_all()."""
        self.reducer, self.lr, self.betas, self.eps, self.max_norm, self.tiled = reducer, lr, betas, eps, max_norm, tiled
        # the kernels read every gradient as grad_scale * g: the reducer leaves the all-reduced SUM in its buckets and the
        # division by the world size rides here instead of one elementwise launch per bucket and step
        reducer.defer_average = True
        self.grad_scale = 1.0 / float(reducer.world)
        # one process: the kernels that write the large gradients leave their squared sums behind and the norm pass skips those
        # ranges (drn_sumsq_partials_skip / drn_sumsq_finalize2); with several ranks the norm is of the all-reduced gradients
        reducer.ext_sumsq = reducer.world == 1 and EXT_SUMSQ
        L = lib()
        L.drn_opt_nblocks.restype = ctypes.c_int64
        dev = reducer.buckets[0].flat.device
        self.state = []
        nparts = 0
        for b in reducer.buckets:
            n = b.flat.numel()
            offs, ptrs = [], []
            ends = list(b.offsets[1:]) + [n]
            for p, off, nxt in zip(b.params, b.offsets, ends):
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous fp32 parameters")
                offs.append(off)
                ptrs.append(p.data_ptr())
                pad_start = off + p.numel()
                if pad_start != nxt:                          # alignment padding (16 B / cache line): a segment with a null pointer
                    offs.append(pad_start)
                    ptrs.append(0)
            offs.append(n)
            nb = int(L.drn_opt_nblocks(ctypes.c_int64(n)))
            import numpy as np
            starts = np.asarray(offs[:-1], dtype=np.int64)
            blk_seg = (np.searchsorted(starts, np.arange(nb, dtype=np.int64) * 4096, side="right") - 1).astype(np.int32)
            self.state.append({"m": torch.zeros_like(b.flat), "v": torch.zeros_like(b.flat),
                               "seg": torch.tensor(offs, dtype=torch.int64, device=dev),
                               "ptr": torch.tensor(ptrs, dtype=torch.int64, device=dev), "ptr_host": list(ptrs),
                               "ptr_index": dict((q, i) for i, q in enumerate(ptrs) if q),
                               "nseg": len(ptrs), "part_off": nparts, "nb": nb,
                               "blk_seg": torch.from_numpy(blk_seg).to(dev)})
            nparts += nb
        self._mirror_sig, self._mirror_keys = None, ()
        self.partials = torch.zeros(nparts, dtype=torch.float32, device=dev)
        self.total_sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=dev)

        self._ptr_sig = tuple(p.data_ptr() for b in reducer.buckets for p in b.params)
        self._updated = frozenset(self._ptr_sig)
        # the weight-copy stores of the parameters this optimizer updates (one per model: drn_amd.functional.WeightCopies)
        self.stores = []
        for b in reducer.buckets:
            for p in b.params:
                st = DF.store_of(p)
                if not any(st is x for x in self.stores):
                    self.stores.append(st)

    def _check_ptrs(self):
        """The kernels reach the parameters through raw pointers baked into device tables: refuse to step if a parameter's
        storage moved since construction (model.to(), .float(), load_state_dict(assign=True) ...).  Host-only comparison."""
        if tuple(p.data_ptr() for b in self.reducer.buckets for p in b.params) != self._ptr_sig:
            raise RuntimeError("parameter storage moved after FusedAdam was built; build a new GradReducer + FusedAdam")

    def step(self, repack=True):
        """repack=False leaves the refresh of the re-laid weight copies to the caller (`repack(codes)`), who may split it
        over two points of its schedule (drn_amd.graph.DualStreamStep)."""
        self.norm()
        self.update()
        if repack:
            self.repack()

    def norm(self):
        """First half of a step: the squared global gradient norm over ALL buckets (+ the device-side step counter).  `update()`
        calls may follow in any split: every one of them clips with this norm and uses this step number."""
        self._check_ptrs()
        L = lib()
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        notes = list(getattr(self.reducer, "sumsq_notes", None) or []) if getattr(self.reducer, "ext_sumsq", False) else []
        spans = [(b.flat.data_ptr(), b.flat.data_ptr() + 4 * b.flat.numel()) for b in self.reducer.buckets]
        # a note is used whole or not at all: every range it covers must be bucket memory (a gradient that was produced elsewhere
        # and copied in is read by the pass like everything else) -- and at most DRN_SUMSQ_MAX_EXT arrays
        notes = [nt for nt in notes if all(any(lo <= ptr and ptr + 4 * ne <= hi for lo, hi in spans) for ptr, ne in nt[0])][:8]
        used = []
        for i, (b, st) in enumerate(zip(self.reducer.buckets, self.state)):
            part = self.partials[st["part_off"]:]
            base, n = b.flat.data_ptr(), b.flat.numel()
            rng = sorted((ptr - base) // 4 for nt in notes for ptr, ne in nt[0] if base <= ptr < base + 4 * n)
            ends = {(ptr - base) // 4: (ptr - base) // 4 + ne for nt in notes for ptr, ne in nt[0] if base <= ptr < base + 4 * n}
            merged = []
            for lo in rng:                                  # adjacent ranges (stacked parameters) become one
                if merged and merged[-1][1] >= lo:
                    merged[-1][1] = max(merged[-1][1], ends[lo])
                else:
                    merged.append([lo, ends[lo]])
            if len(merged) > 16:                            # DRN_SUMSQ_MAX_SKIP: not with DRN's parameter count; take the plain pass
                notes, merged = [], []
            if merged:
                lo_a = (ctypes.c_int64 * len(merged))(*[m[0] for m in merged])
                hi_a = (ctypes.c_int64 * len(merged))(*[min(m[1], n) for m in merged])
                key = (i, tuple((m[0], m[1]) for m in merged))
                cls = self.__dict__.setdefault("_skip_cls", {}).get(key)
                if cls is None and not torch.cuda.is_current_stream_capturing():     # (an upload: never inside a capture)
                    host = (ctypes.c_ubyte * st["nb"])()
                    check(L.drn_sumsq_block_classes(ctypes.c_int64(n), lo_a, hi_a, len(merged), host), "drn_sumsq_block_classes")
                    cls = self._skip_cls[key] = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(b.flat.device)
                check(L.drn_sumsq_partials_skip(P(b.flat), ctypes.c_int64(n), P(part), P(self.step_counter) if i == 0 else None, lo_a, hi_a,
                                                len(merged), P(cls), s), "drn_sumsq_partials_skip")
                used.append(True)
            else:
                check(L.drn_sumsq_partials(P(b.flat), ctypes.c_int64(n), P(part), P(self.step_counter) if i == 0 else None, s),
                      "drn_sumsq_partials")
        if used and notes:
            # (every range of a note must have been left out: they all lie in this reducer's buckets by construction)
            ext = (ctypes.c_void_p * len(notes))(*[nt[1].data_ptr() for nt in notes])
            ext_n = (ctypes.c_int32 * len(notes))(*[nt[1].numel() for nt in notes])
            check(L.drn_sumsq_finalize2(P(self.partials), self.partials.numel(), ext, ext_n, len(notes), P(self.total_sumsq),
                                        ctypes.c_float(self.grad_scale), s), "drn_sumsq_finalize2")
        else:
            check(L.drn_sumsq_finalize(P(self.partials), self.partials.numel(), P(self.total_sumsq), ctypes.c_float(self.grad_scale), s),
                  "drn_sumsq_finalize")
        self._refresh_mirrors()

    def update(self, buckets=None):
        """Second half: clip + Adam on the given buckets of the reducer (all by default), the re-laid copies the kernels maintain
        included.  A caller may update one part of the model early -- the query side, whose next forward pass can then start
        while the rest is still being updated (drn_amd.graph.ForkedStep, optimizer-first order)."""
        L = lib()
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        todo = [(b, st) for b, st in zip(self.reducer.buckets, self.state) if buckets is None or any(b is x for x in buckets)]
        for b, st in todo:
            check(L.drn_adam_bucket(P(b.flat), P(st["m"]), P(st["v"]), ctypes.c_int64(b.flat.numel()), P(st["seg"]), P(st["ptr"]),
                                    st["nseg"], P(st["blk_seg"]), P(st.get("mirror")), P(self.total_sumsq), P(self.step_counter),
                                    ctypes.c_float(self.lr), ctypes.c_float(self.betas[0]), ctypes.c_float(self.betas[1]),
                                    ctypes.c_float(self.eps), ctypes.c_float(self.max_norm), ctypes.c_float(self.grad_scale), s),
                      "drn_adam_bucket")
        for b, st in todo:
            if st.get("tiled") is not None:
                raw, bi, bt, nb = st["tiled"]
                check(L.drn_adam_tiled(P(b.flat), P(st["m"]), P(st["v"]), P(raw), P(bi), P(bt), nb, P(self.total_sumsq),
                                       P(self.step_counter), ctypes.c_float(self.lr), ctypes.c_float(self.betas[0]),
                                       ctypes.c_float(self.betas[1]), ctypes.c_float(self.eps), ctypes.c_float(self.max_norm),
                                       ctypes.c_float(self.grad_scale), s), "drn_adam_tiled")
        DF.bump_weights_epoch(self.stores)       # parameters changed behind autograd's version counters

    def repack(self, codes=None, buckets=None):
        """Refresh the other GEMM-layout copies of the weights (one launch per dtype; `codes`: only these dtypes; `buckets`: only
        the copies of the parameters in these buckets changed -- the rest are marked current)."""
        updated = self._updated if buckets is None else frozenset(p.data_ptr() for b in buckets for p in b.params)
        DF.repack_all(skip=self._mirror_keys, codes=codes, updated=updated, stores=self.stores)

    def _refresh_mirrors(self):
        """Device tables of the cached GEMM operands the optimizer kernels rewrite themselves while they hold the new value:
        (a) bf16 copies in the parameter's own element order (Linear / 1x1-conv forward operands without any other copy):
            drn_adam_bucket's `mirror` table;
        (b) every parameter with a RE-LAID copy (conv weights as [Cout][k][Cin] / [Cin][k][Cout], transposed Linear weights,
            fp32 stacks): taken out of drn_adam_bucket (NULL pointer in its table) and updated tile by tile by drn_adam_tiled,
            which writes up to one copy of each orientation per parameter.
        Rebuilt only when the set of cached copies changes, and never while a hipGraph is being captured (the table upload is
        a host->device copy): copies that appear later are simply left to repack_all()."""
        if DF.cache_generation(self.stores) == getattr(self, "_seen_gen", None) or torch.cuda.is_current_stream_capturing():
            return                                   # no cache entry came or went since the tables were built
        gen = DF.cache_generation(self.stores)
        copies = DF.identity_bf16_copies(self.stores)
        relaid = DF.relaid_copies(self.stores) if self.tiled else {}
        sig = (tuple(sorted((ptr, buf.data_ptr()) for ptr, (key, buf) in copies.items())),
               tuple(sorted((ptr, c["kind"], c["base"].data_ptr()) for ptr, lst in relaid.items() for c in lst)))
        if sig == self._mirror_sig:
            self._seen_gen = gen
            return
        from ._lib import AdamTiledItem
        # A hipGraph captured earlier (GraphedStep, DualStreamStep's "opt" phase) has the ADDRESSES of the current device
        # tables and of the copies they point to baked into its drn_adam_bucket / drn_adam_tiled nodes: the superseded tables
        # stay allocated (never freed, like ops._ws_retired) so that such a replay keeps updating what it updated at capture
        # time instead of reading pointers out of recycled memory.
        retired = self.__dict__.setdefault("_retired", [])
        retired.append((getattr(self, "_keep", None),
                        [(st.get("ptr"), st.get("tiled"), st.get("mirror"), st.get("mirror_bufs")) for st in self.state]))
        keys, keep = [], []
        for b, st in zip(self.reducer.buckets, self.state):
            dev = st["seg"].device
            ptrs = list(st["ptr_host"])
            items, blk_item, blk_tile, used_keys = [], [], [], []
            for p, off in zip(b.params, b.offsets):
                lst = relaid.get(p.data_ptr(), [])
                k = p.shape[2] if p.dim() == 3 else 1
                # tiled when there is anything but a lone same-order bf16 copy (that one rides in drn_adam_bucket)
                c1 = [c for c in lst if c["kind"] == 1]
                c2 = [c for c in lst if c["kind"] == 2]
                lone_identity = len(lst) == 1 and c1 and k == 1 and c1[0]["code"] == 1 and p.dim() in (2, 3)
                if not lst or lone_identity or p.dim() not in (1, 2, 3) or k > 3:
                    continue
                R, C = (p.shape[0], p.shape[1]) if p.dim() >= 2 else (1, p.shape[0])
                it = AdamTiledItem(p=p.data_ptr(), off=off, m1=None, m2=None, ld1=0, ld2=0, R=R, C=C, k=k, code1=0, code2=0,
                                   tiles_c=(C + 63) // 64)
                if c1:
                    it.m1, it.ld1, it.code1 = c1[0]["base"].data_ptr(), c1[0]["ld"], c1[0]["code"]
                    used_keys.append(c1[0]["key"]); keep.append(c1[0]["base"])
                if c2:
                    it.m2, it.ld2, it.code2 = c2[0]["base"].data_ptr(), c2[0]["ld"], c2[0]["code"]
                    used_keys.append(c2[0]["key"]); keep.append(c2[0]["base"])
                ntiles = ((R + 63) // 64) * it.tiles_c
                blk_item += [len(items)] * ntiles
                blk_tile += list(range(ntiles))
                items.append(it)
                ptrs[st["ptr_index"][p.data_ptr()]] = 0          # drn_adam_bucket skips it
            st["tiled"] = None
            if items:
                arr = (AdamTiledItem * len(items))(*items)
                raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
                st["tiled"] = (raw, torch.tensor(blk_item, dtype=torch.int32, device=dev),
                               torch.tensor(blk_tile, dtype=torch.int32, device=dev), len(blk_item))
            st["ptr"] = torch.tensor(ptrs, dtype=torch.int64, device=dev)
            tab = []
            for ptr in ptrs:
                hit = copies.get(ptr) if ptr else None
                tab.append(hit[1].data_ptr() if hit else 0)
                if hit:
                    keys.append(hit[0])
            st["mirror_bufs"] = [copies[q][1] for q in ptrs if q and q in copies]      # keep them alive
            st["mirror"] = torch.tensor(tab, dtype=torch.int64, device=dev) if any(tab) else None
            keys += used_keys
        # a copy made of several parameters is skipped by repack_all only if the tiled kernel writes ALL of its parts
        need = {}
        for ptr, lst in relaid.items():
            for c in lst:
                need[c["key"]] = need.get(c["key"], 0) + 1
        done = {}
        for kk in keys:
            done[kk] = done.get(kk, 0) + 1
        self._keep = keep
        self._seen_gen = gen
        self._mirror_sig = sig
        self._mirror_keys = frozenset(kk for kk in done if not isinstance(kk, tuple) or kk[0] not in ("pack", "pstack", "stack")
                                      or done[kk] >= need.get(kk, 1))

    def total_norm(self):
        return self.total_sumsq[0].sqrt()
