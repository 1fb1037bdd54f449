"""Autograd glue between the reference-shaped Python modules (drn_amd/model) and the HIP kernels.

Each `torch.autograd.Function` below owns one fused stage of the DRN hot path; forward and
backward are sequences of libdrn_hip.so launches (drn_amd/ops.py) on torch's current stream.
Activations are channels-last ("NLC", shape (B, L, C), C contiguous, row stride may exceed C for
column slices of a wider buffer) in the compute dtype (float32 = exact-f32 MFMA parity mode,
bfloat16 = storage with fp32 accumulation).  Parameters stay fp32; their GEMM-layout copies (`packed`, `stacked`,
`stacked_t`) live in the `WeightCopies` store of the model that owns the parameters (`mainModel.weight_copies`; a shared default
for stand-alone layers) and are refreshed by the optimizer kernels / in one launch after the optimizer step (`repack_all`).
"""
import os
import weakref

import torch

from . import ops
from ._lib import DrnError

class WeightCopies(object):
    """The re-laid GEMM copies of ONE model's parameters (what `packed`, `stacked`, `stacked_t` and the stacked conv operands
    hand out), their validity stamps and the two counters the fused optimizer watches.  drn_amd.model.mainModel owns one and
    tags its parameters with it (`p._drn_store`), so the copies live and die with the model; parameters that belong to no model
    (stand-alone layers, tests) share `_default_store`.  Entries are only ever dropped when their parameter is gone: a copy of a
    live parameter may be referenced by raw pointer from a launch descriptor being assembled or from a captured hipGraph."""

    _live = []                                   # weak references to every store (module-level helpers walk them)

    def __init__(self):
        self.pack, self.pstack, self.stack, self.shape_stack = {}, {}, {}, {}
        self.epoch = 0                           # bumped by optimizers that update parameters through raw pointers
        self.gen = 0                             # bumped whenever an entry comes or goes (drn_amd.optim re-reads the tables only then)
        WeightCopies._live.append(weakref.ref(self))

    def bump_gen(self):
        self.gen += 1

    def bump_epoch(self):
        self.epoch += 1

    def adopt(self, module):
        """Tag every parameter of `module` with this store (call again after parameters were replaced)."""
        for p in module.parameters():
            p._drn_store = self
        return self

    def purge_dead(self, cache, limit, refs_of):
        """Bound a cache by dropping the entries of parameters that no longer exist -- and ONLY those."""
        if len(cache) > limit:
            for k in [k for k, e in cache.items() if any(r() is None for r in refs_of(e))]:
                del cache[k]
            self.bump_gen()


_default_store = WeightCopies()


def store_of(params):
    """The WeightCopies of a parameter (or of the first of a list): its model's, or the shared default."""
    p = params[0] if isinstance(params, (list, tuple)) else params
    return getattr(p, "_drn_store", None) or _default_store


def all_stores():
    live = [r() for r in WeightCopies._live]
    WeightCopies._live[:] = [weakref.ref(st) for st in live if st is not None]
    return [st for st in live if st is not None]


def cache_generation(stores=None):
    """Sum of the entry counters of the given stores (all live ones by default): changes whenever a copy comes or goes."""
    return sum(st.gen for st in (all_stores() if stores is None else stores))


def bump_weights_epoch(stores=None):
    """Invalidate every cached re-laid weight of the given stores (all by default): called by optimizers that update parameters
    through raw pointers, behind autograd's version counters."""
    for st in (all_stores() if stores is None else stores):
        st.bump_epoch()


def _w3(w):
    """A Linear weight (N, K) packs like a k=1 conv weight (N, K, 1)."""
    return w.detach().unsqueeze(-1) if w.dim() == 2 else w.detach()


def packed(w, perm, code):
    """Weight `w` (fp32 conv (Cout,Cin,k) or linear (N,K) parameter) permuted + cast for the GEMM; cached on the
    parameter version.  Linear weights come back 2-D.  A tensor made by `stack_params` is packed straight from its source
    parameters (row blocks for the forward layout, column blocks for the data-gradient layout) and cached like them."""
    two_d = w.dim() == 2
    srcs = getattr(w, "_drn_stack_of", None)
    if srcs is not None and not two_d:
        return _packed_stack(srcs, perm, code)
    if not isinstance(w, torch.nn.Parameter):
        # temporaries may reuse an address with version 0: never cache them
        out = ops.pack_weight(_w3(w), perm, code)
        return out.view(out.shape[0], -1) if two_d else out
    S = store_of(w)
    key = (id(w), w.data_ptr(), perm, code)
    ver = (w._version, S.epoch)
    hit = S.pack.get(key)
    # the weakref guards against a new Parameter re-using a dead one's id / address / version
    if hit is not None and hit[0] == ver and hit[2]() is w and hit[1].device == w.device:
        out = hit[1]
    else:
        out = ops.pack_weight(_w3(w), perm, code)
        S.purge_dead(S.pack, 256, lambda e: (e[2],))
        S.pack[key] = (ver, out, weakref.ref(w))
        S.bump_gen()
    return out.view(out.shape[0], -1) if two_d else out


def _pstack_items(out, params, perm):
    """Pack items writing conv weights (Cout_i, Cin, k) into their block of the stacked operand `out`."""
    items, o = [], 0
    for p in params:
        n = p.shape[0]
        dst = out[o:o + n] if perm[0] == 0 else out[:, :, o:o + n]      # (Cout,k,Cin) rows / (Cin,k,Cout) columns
        items.append((p.detach(), perm, dst))
        o += n
    return items


def _packed_stack(params, perm, code):
    S = store_of(params)
    key = (tuple((id(p), p.data_ptr()) for p in params), perm, code)
    ver = (tuple(p._version for p in params), S.epoch)
    hit = S.pstack.get(key)
    if hit is not None and hit[0] == ver and hit[1].device == params[0].device and all(r() is p for r, p in zip(hit[2], params)):
        return hit[1]
    total = sum(p.shape[0] for p in params)
    _, Cin, k = params[0].shape
    shape = (total, k, Cin) if perm[0] == 0 else (Cin, k, total)
    out = hit[1] if hit is not None and hit[1].shape == shape and hit[1].device == params[0].device else \
        torch.empty(shape, dtype=ops.TORCH_DT[code], device=params[0].device)
    ops.pack_weights_into(_pstack_items(out, params, perm), code)
    S.purge_dead(S.pstack, 64, lambda e: e[2])
    S.bump_gen()
    S.pstack[key] = (ver, out, [weakref.ref(p) for p in params])
    return out


def _stack_items(out, params):
    items, r = [], 0
    flat = out.view(out.shape[0], -1)                   # rows of the stack, whatever the parameters' trailing dims
    for p in params:
        n = p.shape[0]
        src = p.detach().reshape(n, 1, -1) if p.dim() > 1 else p.detach().view(1, 1, n)
        items.append((src, (0, 1, 2), flat[r:r + n] if p.dim() > 1 else out[r:r + n]))
        r += n
    return items


def stacked(params):
    """fp32 concatenation along dim 0 of several parameters (e.g. the three qInput{t} weights, W_ih of both LSTM
    directions) so that one GEMM serves them all; built by one drn_pack_weights launch, cached on the parameter versions
    and refreshed in place by repack_all().  Not differentiable: callers compute the per-parameter gradients themselves."""
    S = store_of(params)
    key = tuple((id(p), p.data_ptr()) for p in params)
    ver = (tuple(p._version for p in params), S.epoch)
    hit = S.stack.get(key)
    if hit is not None and hit[0] == ver and hit[1].device == params[0].device and all(r() is p for r, p in zip(hit[2], params)):
        return hit[1]
    shape = (sum(p.shape[0] for p in params),) + tuple(params[0].shape[1:])
    out = hit[1] if hit is not None and hit[1].shape == shape and hit[1].device == params[0].device else \
        torch.empty(shape, dtype=torch.float32, device=params[0].device)
    ops.pack_weights_into(_stack_items(out, params), ops.F32)
    S.purge_dead(S.stack, 64, lambda e: e[2])
    S.bump_gen()
    S.stack[key] = (ver, out, [weakref.ref(p) for p in params])
    return out


def _shape_only_stack(params):
    """A buffer with the shape of cat(params, dim 0) whose CONTENTS are never read: the stacked conv weight of the two towers
    only carries its shape and its sources through autograd (the kernels read the re-laid copies `packed` builds from the
    sources), so it is neither filled nor refreshed after optimizer steps."""
    S = store_of(params)
    key = tuple((id(p), p.data_ptr()) for p in params)
    hit = S.shape_stack.get(key)
    if hit is not None and all(r() is p for r, p in zip(hit[1], params)) and hit[0].device == params[0].device:
        return hit[0]
    out = torch.empty((sum(p.shape[0] for p in params),) + tuple(params[0].shape[1:]), dtype=torch.float32, device=params[0].device)
    S.purge_dead(S.shape_stack, 64, lambda e: e[1])
    S.shape_stack[key] = (out, [weakref.ref(p) for p in params])
    return out


class _StackFn(torch.autograd.Function):
    """cat(params, dim 0) without a copy per step: forward hands out a view of the cached stack (`stacked`), backward
    splits the gradient into views.  The view remembers its sources so `packed` can build the GEMM operand from them."""

    @staticmethod
    def forward(ctx, *params):
        ctx.sizes = [p.shape[0] for p in params]
        ctx.set_materialize_grads(False)             # an absent gradient stays absent for every source
        buf = _shape_only_stack(list(params)) if params[0].dim() == 3 else stacked(list(params))
        out = buf.view(buf.shape)
        out._drn_stack_of = list(params)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * len(ctx.sizes)
        outs, o = [], 0
        for n in ctx.sizes:
            outs.append(g[o:o + n])
            o += n
        return tuple(outs)


def stack_params(params):
    return _StackFn.apply(*params)


def _stack_t_items(out, params):
    items, c = [], 0
    for p in params:                                    # p (N, K) -> out[:, c:c+N] = p^T
        n = p.shape[0]
        items.append((p.detach().unsqueeze(-1), (1, 2, 0), out[:, c:c + n]))
        c += n
    return items


def stacked_t(params):
    """[W_0^T | W_1^T | ...] for Linear weights W_i (N_i, K): a (K, sum N_i) fp32 matrix, so that the input gradient
    dX = [dY_0 | dY_1 | ...] [W_0; W_1; ...] is one NT product.  Cached / refreshed like `stacked`."""
    S = store_of(params)
    key = ("t",) + tuple((id(p), p.data_ptr()) for p in params)
    ver = (tuple(p._version for p in params), S.epoch)
    hit = S.stack.get(key)
    if hit is not None and hit[0] == ver and hit[1].device == params[0].device and all(r() is p for r, p in zip(hit[2], params)):
        return hit[1]
    shape = (params[0].shape[1], sum(p.shape[0] for p in params))
    out = hit[1] if hit is not None and hit[1].shape == shape and hit[1].device == params[0].device else \
        torch.empty(shape, dtype=torch.float32, device=params[0].device)
    ops.pack_weights_into(_stack_t_items(out, params), ops.F32)
    S.purge_dead(S.stack, 64, lambda e: e[2])
    S.bump_gen()
    S.stack[key] = (ver, out, [weakref.ref(p) for p in params])
    return out


def identity_bf16_copies(stores=None):
    """{parameter data_ptr: (cache key, bf16 copy)} for cached GEMM operands that keep the parameter's element order (Linear
    weights, 1x1 convs in the forward layout): the fused optimizer rewrites those itself while it has the value in registers.
    stores: the WeightCopies to look in (all live ones by default)."""
    out = {}
    for S in (all_stores() if stores is None else stores):
        for key, (ver, buf, ref) in S.pack.items():
            w = ref()
            if w is None or key[3] != ops.BF16 or key[2] != (0, 2, 1) or w.data_ptr() != key[1]:
                continue
            if w.dim() == 2 or w.shape[2] == 1:
                out[w.data_ptr()] = (key, buf)
    return out


def relaid_copies(stores=None):
    """Every cached re-laid copy of a live parameter, as the fused optimizer needs it to write the copy itself while it holds
    the updated value (drn_adam_tiled): {parameter data_ptr: [dict(kind 1|2, base tensor, ld, code, k, skip key)]}.
    kind 1 = [r][tap][c] order (element (r, c, tap) at base[(r*k + tap)*ld + c]): forward conv / Linear operands, fp32 stacks;
    kind 2 = [c][tap][r] order: data-gradient operands, transposed Linear weights.  A copy made of several parameters (stacked
    operands) is listed under each of them with the same skip key."""
    out = {}

    def add(p, kind, base, ld, code, key):
        out.setdefault(p.data_ptr(), []).append(dict(param=p, kind=kind, base=base, ld=ld, code=code, key=key,
                                                     k=p.shape[2] if p.dim() == 3 else 1))

    for S in (all_stores() if stores is None else stores):
        _relaid_of_store(S, add)
    return out


def _relaid_of_store(S, add):
    for key, (ver, buf, ref) in S.pack.items():
        w = ref()
        if w is None or w.data_ptr() != key[1] or buf.device != w.device or w.dim() not in (2, 3):
            continue
        Cout, Cin = w.shape[0], w.shape[1]
        if key[2] == (0, 2, 1):
            add(w, 1, buf, Cin, key[3], ("pack", key))
        elif key[2] == (1, 2, 0):
            add(w, 2, buf, Cout, key[3], ("pack", key))
    for key, (ver, buf, refs) in S.pstack.items():
        ps = [r() for r in refs]
        if any(p is None for p in ps) or tuple((id(p), p.data_ptr()) for p in ps) != key[0]:
            continue
        o, total = 0, sum(p.shape[0] for p in ps)
        for p in ps:
            n, Cin = p.shape[0], p.shape[1]
            if key[1] == (0, 2, 1):
                add(p, 1, buf[o:o + n], Cin, key[2], ("pstack", key))
            elif key[1] == (1, 2, 0):
                add(p, 2, buf[:, :, o:o + n], total, key[2], ("pstack", key))
            o += n
    for key, (ver, buf, refs) in S.stack.items():
        ps = [r() for r in refs]
        transposed = key[0] == "t"
        if any(p is None for p in ps) or tuple((id(p), p.data_ptr()) for p in ps) != (key[1:] if transposed else key):
            continue
        if transposed:
            c, total = 0, sum(p.shape[0] for p in ps)
            for p in ps:                                     # p (N, K) -> buf[:, c:c+N] = p^T
                add(p, 2, buf[:, c:c + p.shape[0]], total, ops.F32, ("stack", key))
                c += p.shape[0]
        elif all(p.dim() in (1, 2) for p in ps):
            r = 0
            for p in ps:                                     # rows [r, r+n) of the stack = p itself
                add(p, 1, buf[r:r + p.shape[0]], p.shape[1] if p.dim() == 2 else p.shape[0], ops.F32, ("stack", key))
                r += p.shape[0]


def repack_all(skip=(), codes=None, updated=None, stores=None):
    """Refresh every cached re-laid weight IN PLACE with one launch per dtype (drn_pack_weights) and mark it current:
    an optimizer that has just updated all parameters calls this instead of leaving ~20 small per-use launches to
    the next forward pass.  The copies keep their addresses, so captured hipGraphs stay valid.  `skip`: cache keys the
    caller has already refreshed itself (identity_bf16_copies).  `codes`: only the copies of these dtypes (ops.F32 holds
    the query side's stacks, ops.BF16 the conv / linear operands of a bf16 model) -- the rest stay stale until their call.
    `updated`: data_ptrs of the parameters the caller has changed; copies of all other parameters (frozen ones: mix_fc and
    iou_scores in stage 1) are still valid and only marked current.  `stores`: the WeightCopies to refresh (all live ones by default)."""
    for S in (all_stores() if stores is None else stores):
        _repack_store(S, skip, codes, updated)


def _repack_store(S, skip, codes, updated):
    want = lambda code: codes is None or code in codes
    same = lambda ps: updated is not None and not any(p.data_ptr() in updated for p in ps)
    by_code = {}
    for key, (ver, out, ref) in list(S.pack.items()):
        w = ref()
        if w is None or w.data_ptr() != key[1] or out.device != w.device:
            del S.pack[key]
            S.bump_gen()
            continue
        if not want(key[3]):
            continue
        # (a copy somebody else refreshed / an untouched parameter's copy is only marked current -- unless the parameter was
        # modified in place since the copy was built (load_state_dict, copy_): then it is repacked like the rest)
        if (key in skip or ("pack", key) in skip or same([w])) and ver[0] == w._version:
            S.pack[key] = ((w._version, S.epoch), out, ref)
            continue
        by_code.setdefault(key[3], []).append((key, w, out))
    for key, (ver, out, refs) in list(S.pstack.items()):
        ps = [r() for r in refs]
        if any(p is None for p in ps) or tuple((id(p), p.data_ptr()) for p in ps) != key[0] or out.device != ps[0].device:
            del S.pstack[key]
            S.bump_gen()
            continue
        if not want(key[2]):
            continue
        if (("pstack", key) not in skip and not same(ps)) or ver[0] != tuple(p._version for p in ps):
            by_code.setdefault(key[2], [])
            by_code.setdefault(("stack", key[2]), []).extend(_pstack_items(out, ps, key[1]))
        S.pstack[key] = ((tuple(p._version for p in ps), S.epoch), out, refs)
    stack_items = []
    for key, (ver, out, refs) in list(S.stack.items()):
        ps = [r() for r in refs]
        transposed = key[0] == "t"
        if any(p is None for p in ps) or tuple((id(p), p.data_ptr()) for p in ps) != (key[1:] if transposed else key) \
                or out.device != ps[0].device:
            del S.stack[key]
            S.bump_gen()
            continue
        if not want(ops.F32):
            continue
        if (("stack", key) not in skip and not same(ps)) or ver[0] != tuple(p._version for p in ps):
            stack_items += _stack_t_items(out, ps) if transposed else _stack_items(out, ps)
        S.stack[key] = ((tuple(p._version for p in ps), S.epoch), out, refs)
    for code, items in by_code.items():
        if isinstance(code, tuple):
            continue
        extra = (stack_items if code == ops.F32 else []) + by_code.get(("stack", code), [])
        if items or extra:
            ops.pack_weights_into([(_w3(w), key[2], out) for key, w, out in items] + extra, code)
        for key, w, out in items:
            S.pack[key] = ((w._version, S.epoch), out, weakref.ref(w))
    if stack_items and ops.F32 not in by_code:
        ops.pack_weights_into(stack_items, ops.F32)


# Gradient sinks: drn_amd.dist.GradReducer registers, per parameter storage, the slice of its flat bucket where that
# parameter's gradient must end up.  A backward that writes the weight gradient straight into a fresh VIEW of the
# slice lets autograd's AccumulateGrad adopt it (p.grad is None -> the incoming tensor is kept, no zero-fill, no add).
_grad_sinks = {}


def register_grad_sink(param, flat_slice):
    _grad_sinks[param.data_ptr()] = (weakref.ref(param), flat_slice)


def _sink_of(ptr):
    """The registered sink for the parameter storage at `ptr` -- None once the parameter that registered it is gone (a new
    tensor at a recycled address must not inherit a dead model's bucket)."""
    ent = _grad_sinks.get(ptr)
    if ent is None:
        return None
    owner = ent[0]()
    if owner is None or owner.data_ptr() != ptr:
        del _grad_sinks[ptr]
        _sink_handed.discard(ent[1].data_ptr())
        return None
    return ent[1]


def unregister_grad_sinks(params):
    """Forget the sinks of these parameters only (another model's reducer in the same process keeps its own)."""
    for p in params:
        ent = _grad_sinks.pop(p.data_ptr(), None)
        if ent is not None:
            _sink_handed.discard(ent[1].data_ptr())


def clear_grad_sinks():
    _grad_sinks.clear()
    _sink_handed.clear()


_sink_handed = set()     # data_ptr of every buffer grad_buffer() handed out that IS (part of) a reducer's flat bucket


def grad_buffer(param, dtype=torch.float32):
    """Uninitialised fp32 buffer shaped like `param` for its gradient: the registered sink when there is one.  For a
    `stack_params` view: the sinks of its sources when they lie back to back in that order (drn_amd.dist.GradReducer puts
    `adjacent=` groups that way), so the stacked gradient is written straight into the flat bucket and the per-source
    views autograd hands to the parameters ARE their sinks (no copy into the bucket afterwards)."""
    sink = _sink_of(param.data_ptr())
    if sink is not None and sink.numel() == param.numel() and sink.dtype == dtype and sink.device == param.device:
        _sink_handed.add(sink.data_ptr())
        return sink.view(param.shape)
    srcs = getattr(param, "_drn_stack_of", None)
    if srcs:
        sinks = [_sink_of(p.data_ptr()) for p in srcs]
        if all(s is not None and s.numel() == p.numel() and s.dtype == dtype and s.device == param.device
               for s, p in zip(sinks, srcs)) and \
                all(a.data_ptr() + a.numel() * a.element_size() == b.data_ptr() for a, b in zip(sinks, sinks[1:])) and \
                sum(s.numel() for s in sinks) == param.numel():
            strides, acc = [], 1
            for d in reversed(param.shape):
                strides.append(acc)
                acc *= d
            _sink_handed.add(sinks[0].data_ptr())
            return sinks[0].as_strided(tuple(param.shape), tuple(reversed(strides)), sinks[0].storage_offset())
        # one-element sources (the Scale parameters) at a fixed spacing inside the bucket (each slice starts on a 16-byte
        # boundary): a strided 1-D view, element l = source l's sink
        if all(s is not None and s.numel() == 1 and p.numel() == 1 and s.dtype == dtype and s.device == param.device
               for s, p in zip(sinks, srcs)) and param.dim() == 1 and len(sinks) > 1:
            step = (sinks[1].data_ptr() - sinks[0].data_ptr()) // sinks[0].element_size()
            if step > 0 and all(b.data_ptr() - a.data_ptr() == step * a.element_size() for a, b in zip(sinks, sinks[1:])):
                return sinks[0].as_strided((len(sinks),), (step,), sinks[0].storage_offset())
    return torch.empty(param.shape, dtype=dtype, device=param.device)


# Deferred weight gradients.  A weight gradient feeds nothing but the optimizer, so a backward node may hand its launch
# to the caller instead of issuing it in line: drn_amd.graph.DualStreamStep collects them (`begin_defer` / `take_deferred`)
# and runs them on the main stream WHILE the query side's backward -- a long chain of small, latency-bound launches --
# runs on a second stream.  Only gradients written straight into a reducer's flat bucket qualify (autograd then merely
# adopts the view; nothing reads the values before the optimizer); anything else is launched in line as before.
_deferred = None


def begin_defer():
    global _deferred
    _deferred = []


def take_deferred():
    """The jobs collected since begin_defer(), in issue order; deferral is switched off."""
    global _deferred
    jobs, _deferred = (_deferred or []), None
    return jobs


def _alias(t):
    """Another tensor object on the same memory.  A deferred job must not hold the very tensor its node returns to autograd:
    AccumulateGrad adopts an incoming gradient only while nobody else references it -- otherwise it CLONES it, here before
    the deferred kernel has written it."""
    return t.view(t.shape)


def _defer(job, *grad_bufs):
    if _deferred is not None and all(b.data_ptr() in _sink_handed for b in grad_bufs):
        _deferred.append(job)
    else:
        job()


# Squared-sum partials left behind by the kernels that write gradients (for the fused optimizer's norm pass, one process only):
# a reducer that wants them says so before backward (want_sumsq), producers note what they covered, the reducer collects the notes.
_sumsq_want = False
_sumsq_notes = []


def want_sumsq(on):
    global _sumsq_want
    _sumsq_want = bool(on)
    del _sumsq_notes[:]


def note_sumsq(ranges, partials):
    """ranges: [(data_ptr, elements)] of gradient memory whose squares are summed in `partials` (a device tensor, all of it)."""
    _sumsq_notes.append((list(ranges), partials))


def take_sumsq_notes():
    notes = list(_sumsq_notes)
    del _sumsq_notes[:]
    return notes


NT_WGRAD = True      # prop_fc weight gradient through the NT kernel on transposed operands (bf16): 250 vs 410 us for the TN kernel
# 1 = warm the prop_fc weight copy right before its GEMM when the general kernel runs it (round 2: 2.562 vs 2.577 ms without).  Off since
# round 5: with the Adam moments and fp32 masters accessed non-temporally the bf16 copy Adam writes is still in the Infinity Cache when the
# GEMM starts -- T = 32, three pairs in one box: 1.208 -> 1.201 ms linear, 1.149 -> 1.148 two-branch without the touch launch
TOUCH_W = os.environ.get("DRN_TOUCH_W", "0") != "0"

# BatchNorm `num_batches_tracked` increments are collected during a forward pass and applied by ONE multi-tensor add
# (flush_bn_counters) instead of one tiny launch per BN call.
_bn_pending = {}


def bump_bn_counter(t, n=1):
    key = id(t)
    if key in _bn_pending:
        _bn_pending[key][1] += n
    else:
        _bn_pending[key] = [t, n]


def take_bn_counters(device):
    """Hand the pending increments to a kernel that applies them itself (the loss's final kernel); [] when there are none or
    too many / foreign ones (those stay for flush_bn_counters)."""
    from ._lib import LOSS_MAX_BUMPS
    if not _bn_pending or len(_bn_pending) > LOSS_MAX_BUMPS:
        return []
    items = list(_bn_pending.values())
    if any(t.device != device or t.dtype != torch.int64 for t, _ in items):
        return []
    _bn_pending.clear()
    return items


def flush_bn_counters():
    if _bn_pending:
        items = list(_bn_pending.values())
        _bn_pending.clear()
        with torch.no_grad():
            torch._foreach_add_([t for t, _ in items], [n for _, n in items])


def code_of(dtype):
    return ops.BF16 if dtype == torch.bfloat16 else ops.F32


def geom(x):
    """(B, L, C, ld) of a channels-last activation; validates the layout."""
    if x.dim() != 3 or x.stride(2) != 1 or (x.shape[0] > 1 and x.stride(0) != x.shape[1] * x.stride(1)):
        raise DrnError("expected a channels-last (B, L, C) activation, got shape %s strides %s"
                       % (tuple(x.shape), x.stride()))
    return x.shape[0], x.shape[1], x.shape[2], x.stride(1)


def as_nlc(x_ncl, dtype):
    """Logical (B, C, L) tensor -> (B, L, C) channels-last view in the compute dtype (copy only if needed)."""
    x = x_ncl.permute(0, 2, 1)
    if x.dtype != dtype:
        x = x.to(dtype)
    if x.stride(2) != 1 or (x.shape[0] > 1 and x.stride(0) != x.shape[1] * x.stride(1)):
        x = x.contiguous()
    return x


def _grad_nlc(g, like_shape, dtype):
    if g is None:
        return None
    if g.dtype != dtype:
        g = g.to(dtype)
    if not g.is_contiguous():
        g = g.contiguous()
    return g


class _CastActFn(torch.autograd.Function):
    """Activation dtype change between two stages that run in different compute dtypes (see mainModel.forward_front: an
    unaligned feature dimension keeps the input stage and conv0 on the exact-f32 kernels inside a bf16 model)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return x.to(dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.src), None


def cast_act(x, dtype):
    return x if x.dtype == dtype else _CastActFn.apply(x, dtype)


# Debug tap (tests/test_parity_grad_gpu.py): a list that receives (weight tensor object, level, bool mask (B, L, C)) -- which
# ReLU outputs were passed -- for every conv->BN->ReLU call and for the query encoder's qInput ReLU, so that a reference run can
# be given this run's discrete decisions.  None (the default) costs nothing.
relu_tap = None


def _tap_relu(weight, level, out, up=None):
    if relu_tap is None:
        return
    if up is None:
        mask = out > 0
    else:             # out = relu(y) + nearest_x2(up): passed <=> the sum moved (a y below half an ulp of `up` reads as blocked)
        mask = out != up.repeat_interleave(2, dim=1)
    relu_tap.append((weight, level, mask))


class ConvMeta(object):
    """Static (non-tensor) description of one conv+BN(+ReLU) block call."""

    def __init__(self, stride, bn, training, dtype, relu=True, tail=None):
        self.stride, self.bn, self.training, self.dtype, self.relu = stride, bn, training, dtype, relu
        self.tail = tail          # EmbedTail: the last P input channels are a Linear of per-row features (see input_stage)


class _ConvBlockFn(torch.autograd.Function):
    """Conv1d(k, stride, pad=(k-1)//2) -> BatchNorm1d -> ReLU over 1..3 pyramid levels with shared weights
    (model/basic_blocks.py:5-33; model/fcos.py:29-42,58-69), with the consumer prologues fused into the
    BN-apply pass: `gate` (B,C) -> second output out*gate (model/backbone.py:28-30), `up` (B,L/2,C) ->
    out += nearest-x2 upsample (model/FPN.py:63-68).  Implicit GEMM on MFMA, batch statistics from the
    GEMM epilogue."""

    @staticmethod
    def forward(ctx, meta, weight, cbias, gamma, beta, gate, up, *xs):
        dt = meta.dtype
        code = code_of(dt)
        Cout, Cin, k = weight.shape
        pad = (k - 1) // 2
        nl = len(xs)
        assert nl == 1 or (gate is None and up is None)
        wp = packed(weight, (0, 2, 1), code)
        dev = weight.device
        geo, raws, stats, descs = [], [], [], []
        for x in xs:
            if x.dtype != dt:
                raise DrnError("activation dtype %s != compute dtype %s" % (x.dtype, dt))
            B, L, C, ld = geom(x)
            assert C == Cin, (C, Cin)
            Lo = (L + 2 * pad - k) // meta.stride + 1
            M = B * Lo
            raw = torch.empty((B, Lo, Cout), dtype=dt, device=dev)
            descs.append(ops.gemm_desc(x, wp, raw, M, Cout, Cin, taps=k, stride=meta.stride, pad=pad, Lout=Lo, Lsrc=L, lda=ld))
            geo.append((B, L, Lo, M, ld))
            raws.append(raw)
        bn = meta.bn
        sss, saves = [], []
        fused = meta.training and Cout % 64 == 0       # statistics merge + running statistics + apply in ONE launch

        def gemm_with_stats():
            """The conv as its own launch; in training it leaves the per-slab statistics for the BatchNorm pass that follows."""
            for l in range(nl):
                st = torch.empty(((geo[l][3] + 127) // 128, 2, Cout), dtype=torch.float32, device=dev) if meta.training else None
                stats.append(st)
                descs[l].stats = ops._p(st)
            ops.gemm_nt(descs, code)
        if not fused:
            gemm_with_stats()
        track = False
        if meta.training:
            if bn.momentum is None:
                raise DrnError("cumulative-average BatchNorm (momentum=None) is not supported")
            groups = []
            for l in range(nl):
                ss = torch.empty((2, Cout), dtype=torch.float32, device=dev)
                sv = torch.empty((2, Cout), dtype=torch.float32, device=dev)
                if not fused:
                    groups.append((stats[l], stats[l].shape[0], geo[l][3], ss, sv))
                sss.append(ss)
                saves.append(sv)
            track = bn.track_running_stats and bn.running_mean is not None
            if not fused:
                ops.bn_finalize(groups, Cout, gamma, beta, cbias, bn.running_mean if track else None,
                                bn.running_var if track else None, bn.momentum, bn.eps)
            if track and bn.num_batches_tracked is not None:
                bump_bn_counter(bn.num_batches_tracked, nl)
        else:
            ss = torch.empty((2, Cout), dtype=torch.float32, device=dev)
            ops.bn_eval_scale_shift(Cout, gamma, beta, cbias, bn.running_mean, bn.running_var, bn.eps, ss)
            sss = [ss] * nl
        outs, gated, levels = [], None, []
        for l in range(nl):
            B, L, Lo, M, ld = geo[l]
            out = torch.empty((B, Lo, Cout), dtype=dt, device=dev)
            if gate is not None:
                gated = torch.empty((B, Lo, Cout), dtype=dt, device=dev)
            upl = None
            if up is not None:
                ub, ul, uc, uld = geom(up)
                assert (ub, ul * 2, uc) == (B, Lo, Cout), "upsample source must be (B, L/2, C)"
                upl = up
            levels.append(dict(raw=raws[l], ld_raw=Cout, ss=sss[l], out=out, ld_out=Cout, M=M, L=Lo, up=upl,
                               ld_up=upl.stride(1) if upl is not None else 0, gate=gate, gated=gated, ld_gated=Cout))
            if fused:
                levels[-1].update(tiles=(M + 127) // 128, save=saves[l], gamma=gamma, beta=beta, conv_bias=cbias,
                                  running_mean=bn.running_mean if track else None, running_var=bn.running_var if track else None,
                                  momentum=bn.momentum, eps=bn.eps)
            outs.append(out)
        if fused:
            # conv -> BN -> ReLU of all pyramid levels in ONE launch (the workgroups normalise their own tiles after a per-column
            # arrival wait); where that launch does not apply (split-K, external upsample source, grid beyond what the chip holds
            # at once) the GEMM and the one-launch BatchNorm pass run separately -- same bits either way
            if not ops.conv_bn_train(descs, levels, code, relu=meta.relu):
                gemm_with_stats()
                for l in range(nl):
                    levels[l]["stats"] = stats[l]
                ops.bn_train_apply(levels, Cout, code, relu=meta.relu)
        else:
            ops.bn_apply_multi(levels, Cout, code, relu=meta.relu)        # all pyramid levels in one launch
        if relu_tap is not None and meta.relu:
            for l in range(nl):
                _tap_relu(weight, l, outs[l], up)
        ctx.meta, ctx.nl, ctx.geo, ctx.k = meta, nl, geo, k
        ctx.weight_obj = weight          # the Python object (a stack_params view carries its sources; saved tensors do not)
        ctx.has_gate, ctx.has_up, ctx.has_cbias = gate is not None, up is not None, cbias is not None
        ctx.beta_ref = beta
        ctx.save_for_backward(weight, gamma, gate if gate is not None else weight.new_empty(0), *xs, *raws, *sss, *saves, *outs)
        res = tuple(outs)
        if gate is not None:
            res = res + (gated,)
        return res

    @staticmethod
    def backward(ctx, *gouts):
        meta, nl, geo, k = ctx.meta, ctx.nl, ctx.geo, ctx.k
        if not meta.training:
            raise DrnError("backward through an eval-mode BatchNorm block is not supported")
        dt = meta.dtype
        code = code_of(dt)
        sv = ctx.saved_tensors
        weight, gamma, gate = sv[0], sv[1], sv[2]
        xs = sv[3:3 + nl]
        raws = sv[3 + nl:3 + 2 * nl]
        sss = sv[3 + 2 * nl:3 + 3 * nl]
        saves = sv[3 + 3 * nl:3 + 4 * nl]
        outs = sv[3 + 4 * nl:3 + 5 * nl]
        Cout, Cin, _ = weight.shape
        pad = (k - 1) // 2
        dev = weight.device
        dgamma = grad_buffer(gamma)
        dbeta = grad_buffer(ctx.beta_ref) if ctx.beta_ref is not None else torch.empty_like(gamma)
        dgate = dup = None
        draws, blevels = [], []
        for l in range(nl):
            B, L, Lo, M, ld = geo[l]
            d = _grad_nlc(gouts[l], None, dt)
            gb = None
            if ctx.has_gate:
                dG = _grad_nlc(gouts[nl], None, dt)
                if dG is not None:
                    dgate = torch.empty((B, Cout), dtype=torch.float32, device=dev)
                    if GATE_BN_FUSE and not ctx.has_up and nl == 1:
                        # the gate backward rides in the BatchNorm backward launch (DrnBnBwdDesc::gb_*; ops.bn_bwd_multi runs it as a
                        # launch of its own where that kernel cannot take it)
                        gb = dict(dg=dG, ld_dg=Cout, gate=gate, ldg=gate.stride(0), dgate=dgate, L=Lo, act=outs[l], ld_act=Cout)
                    else:
                        add, d = d, torch.empty((B, Lo, Cout), dtype=dt, device=dev)
                        ops.gate_bwd(dG, Cout, outs[l], Cout, gate, d, Cout, add, Cout, dgate, B, Lo, Cout, code)
                else:
                    dgate = torch.zeros((B, Cout), dtype=torch.float32, device=dev)
            if d is None and gb is None:
                d = torch.zeros((B, Lo, Cout), dtype=dt, device=dev)
            if ctx.has_up:
                dup = torch.empty((B, Lo // 2, Cout), dtype=dt, device=dev)
                ops.pairsum_add(dup, Cout, d, Cout, B * (Lo // 2), Cout, code, accumulate=False)
            draw = torch.empty((B, Lo, Cout), dtype=dt, device=dev)
            blevels.append(dict(dout=d, ld_dout=Cout, raw=raws[l], ld_raw=Cout, ss=sss[l], save=saves[l], gamma=gamma, draw=draw,
                                ld_draw=Cout, dgamma=dgamma, dbeta=dbeta, accumulate=l > 0, M=M, gb=gb))
            draws.append(draw)
        ops.bn_bwd_multi(blevels, Cout, code, relu=meta.relu)             # reduce / finalize / apply once for all levels
        dW = grad_buffer(ctx.weight_obj if getattr(ctx.weight_obj, "_drn_stack_of", None) else weight)
        wdescs = [ops.wgrad_desc(draws[l], xs[l], geo[l][3], Lout=geo[l][2], Lsrc=geo[l][1], ldy=Cout, ldx=geo[l][4])
                  for l in range(nl)]
        keep = (draws, xs)          # (the descriptors hold raw pointers: a deferred job keeps the operands alive)
        dWj = _alias(dW)
        _defer(lambda: (keep, ops.gemm_wgrad(wdescs, dWj, Cout, Cin, taps=k, stride=meta.stride, pad=pad, w_layout=1, dtype=code)), dWj)
        dxs = [None] * nl
        if any(ctx.needs_input_grad[7 + l] for l in range(nl)):
            wd = packed(ctx.weight_obj, (1, 2, 0), code)               # (Cin, k, Cout)
            tail = meta.tail if (meta.tail is not None and nl == 1 and k in (1, 3) and meta.tail.usable(Cin, Cout, dt)) else None
            ncols = Cin - tail.P if tail is not None else Cin
            descs = []
            for l in range(nl):
                B, L, Lo, M, ld = geo[l]
                dx = torch.empty((B, L, Cin), dtype=dt, device=dev)
                # (with a tail the product stops at the feature columns: the embedding columns of dx stay unwritten, and nobody
                # reads them -- the input stage takes its embedding gradients from the tail)
                descs.append(ops.gemm_desc(draws[l], wd, dx, B * L, ncols, Cout, taps=k, stride=meta.stride, pad=pad, mode=1,
                                           Lout=L, Lsrc=Lo, ldc=Cin))
                dxs[l] = dx
            gctx = tail.gate_ctx if tail is not None else None
            if (gctx is not None and k == 3 and meta.stride == 1 and gctx[2:] == (geo[0][0], geo[0][1], ncols)
                    and ops.gemm_nt_plan(descs, code) == ops.NT_KIND_W4C):
                # the input stage's gate backward in this launch's epilogue: dx is not written, its consumer gets these instead
                Z, gate0, B, T, D = gctx
                dZT = torch.empty((D, B * T), dtype=dt, device=dev)
                dgate0 = torch.empty((B, D), dtype=torch.float32, device=dev)
                dsum0 = torch.empty((B, D), dtype=torch.float32, device=dev)
                descs = [ops.gemm_desc(draws[0], wd, dxs[0], B * T, ncols, Cout, taps=k, stride=1, pad=pad, mode=1, Lout=T, Lsrc=T, ldc=Cin,
                                       gate=gate0, ldg=gate0.stride(0),
                                       gate_bwd=dict(act=Z, ld_act=D, dct=dZT, ldt=B * T, dgate=dgate0, dsum=dsum0))]
                tail.gate_out = (dZT, dgate0, dsum0)
            ops.gemm_nt(descs, code)
            if tail is not None:
                B, L, Lo, M, ld = geo[0]
                tail.backward(draws[0], wd, B, L, Lo, Cout, Cin, k, meta.stride, pad, code)
        # a conv bias in front of a train-mode BN cancels: its gradient is exactly zero -> None (no fill, no copy into the
        # bucket: the reducer's slice of a parameter without gradient reads as zero)
        return (None, dW, None, dgamma, dbeta, dgate, dup) + tuple(dxs)


def conv_block(xs, conv, bn, training, dtype, gate=None, up=None, relu=True, tail=None):
    """Apply conv->BN->ReLU (modules are parameter holders) to a list of channels-last level inputs.  `tail`: the EmbedTail of
    an input that came out of input_stage (single level only)."""
    meta = ConvMeta(conv.stride[0], bn, training, dtype, relu, tail=tail)
    res = _ConvBlockFn.apply(meta, conv.weight, conv.bias, bn.weight, bn.bias, gate, up, *xs)
    if gate is not None:
        return list(res[:-1]), res[-1]
    return list(res), None


class _PlainConvFn(torch.autograd.Function):
    """Conv1d(k, stride, pad=(k-1)//2) [+ bias] [-> ReLU] without BatchNorm: the factory variants of model/basic_blocks.py:5-33
    that DRN itself never instantiates (use_bn=False) and the FPN top blocks (model/FPN.py:86-103).  Same implicit-GEMM
    kernels; the bias rides in the GEMM epilogue, the ReLU in one elementwise pass (identity scale/shift through the BN-apply
    kernel)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, relu, dtype):
        code = code_of(dtype)
        Cout, Cin, k = weight.shape
        pad = (k - 1) // 2
        B, L, C, ld = geom(x)
        assert C == Cin and x.dtype == dtype
        Lo = (L + 2 * pad - k) // stride + 1
        M = B * Lo
        dev = x.device
        raw = torch.empty((B, Lo, Cout), dtype=dtype, device=dev)
        ops.gemm_nt([ops.gemm_desc(x, packed(weight, (0, 2, 1), code), raw, M, Cout, Cin, taps=k, stride=stride, pad=pad, Lout=Lo,
                                   Lsrc=L, lda=ld, bias=bias.detach() if bias is not None else None)], code)
        out = raw
        if relu:
            ss = torch.cat([torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)]).view(2, Cout)
            out = torch.empty_like(raw)
            ops.bn_apply(raw, Cout, ss, out, Cout, M, Cout, Lo, code, relu=True)
        ctx.meta = (stride, relu, dtype, (B, L, Lo, M, ld), k)
        ctx.has_bias = bias is not None
        ctx.weight_obj, ctx.bias_obj = weight, bias
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        stride, relu, dtype, (B, L, Lo, M, ld), k = ctx.meta
        code = code_of(dtype)
        x, weight, out = ctx.saved_tensors
        Cout, Cin, _ = weight.shape
        pad = (k - 1) // 2
        dev = x.device
        d = _grad_nlc(dout, None, dtype)
        if relu:
            d = d * (out > 0)                                   # (off the DRN hot path: one framework elementwise op)
        db = None
        if ctx.has_bias:
            db = grad_buffer(ctx.bias_obj)
            ops.colsum(d.view(M, Cout), Cout, M, Cout, db, code)
        dW = grad_buffer(ctx.weight_obj)
        ops.gemm_wgrad([ops.wgrad_desc(d, x, M, Lout=Lo, Lsrc=L, ldy=Cout, ldx=ld)], dW, Cout, Cin, taps=k, stride=stride, pad=pad,
                       w_layout=1, dtype=code)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, L, Cin), dtype=dtype, device=dev)
            ops.gemm_nt([ops.gemm_desc(d, packed(ctx.weight_obj, (1, 2, 0), code), dx, B * L, Cin, Cout, taps=k, stride=stride,
                                       pad=pad, mode=1, Lout=L, Lsrc=Lo)], code)
        return dx, dW, db, None, None, None


def plain_conv(x, conv, dtype, relu=False):
    """conv: nn.Conv1d parameter holder (dilation 1, padding (k-1)//2); x channels-last (B, L, Cin)."""
    if conv.dilation[0] != 1 or conv.padding[0] != (conv.kernel_size[0] - 1) // 2:
        raise DrnError("plain_conv: dilation 1 and 'same' padding only")
    return _PlainConvFn.apply(x, conv.weight, conv.bias, conv.stride[0], relu, dtype)


class _MultiConvFn(torch.autograd.Function):
    """n INDEPENDENT conv->BN->ReLU blocks (different weights, one input each) launched together: the implicit GEMMs of
    all blocks go through ONE grouped launch forward and ONE for the data gradients, so the coarse pyramid levels (64-128
    workgroups on their own) share the chip with the fine one.  With chain_up the blocks are the FPN laterals:
    out_l = relu(bn(conv_l(x_l))) + nearest_x2(out_{l+1}) (model/FPN.py:54-68), resolved coarse-to-fine in the BN-apply
    passes; without it they are the FPN output convs (FPN.py:56,69)."""

    @staticmethod
    def forward(ctx, meta, *args):
        n, dt, training, chain_up = meta["n"], meta["dtype"], meta["training"], meta["chain_up"]
        code = code_of(dt)
        weights, gammas, betas, xs = args[0:n], args[n:2 * n], args[2 * n:3 * n], args[3 * n:4 * n]
        bns, strides = meta["bns"], meta["strides"]
        dev = xs[0].device
        geo, raws, stats, descs = [], [], [], []
        for l in range(n):
            w, x = weights[l], xs[l]
            Cout, Cin, k = w.shape
            pad = (k - 1) // 2
            B, L, C, ld = geom(x)
            assert C == Cin and x.dtype == dt
            Lo = (L + 2 * pad - k) // strides[l] + 1
            M = B * Lo
            raw = torch.empty((B, Lo, Cout), dtype=dt, device=dev)
            descs.append(ops.gemm_desc(x, packed(w, (0, 2, 1), code), raw, M, Cout, Cin, taps=k, stride=strides[l], pad=pad,
                                       Lout=Lo, Lsrc=L, lda=ld))
            geo.append((B, L, Lo, M, ld, Cout, Cin, k, pad))
            raws.append(raw)
        sss, saves, fin = [], [], []
        same_c = all(g[5] == geo[0][5] for g in geo)
        for l in range(n):
            bn, Cout = bns[l], geo[l][5]
            ss = torch.empty((2, Cout), dtype=torch.float32, device=dev)
            if training:
                if bn.momentum is None:
                    raise DrnError("cumulative-average BatchNorm (momentum=None) is not supported")
                sv = torch.empty((2, Cout), dtype=torch.float32, device=dev)
                track = bn.track_running_stats and bn.running_mean is not None
                fin.append(dict(tiles=(geo[l][3] + 127) // 128, M=geo[l][3], ss=ss, save=sv, gamma=gammas[l], beta=betas[l],
                                running_mean=bn.running_mean if track else None, running_var=bn.running_var if track else None,
                                momentum=bn.momentum, eps=bn.eps))
                if track and bn.num_batches_tracked is not None:
                    bump_bn_counter(bn.num_batches_tracked, 1)
                saves.append(sv)
            else:
                ops.bn_eval_scale_shift(Cout, gammas[l], betas[l], None, bn.running_mean, bn.running_var, bn.eps, ss)
                saves.append(ss)
            sss.append(ss)
        # C % 64 == 0 (every FPN block): the apply launches merge the statistics themselves; otherwise a finalize launch first
        fused = bool(fin) and all(g[5] % 64 == 0 for g in geo)
        outs = [torch.empty((g[0], g[2], g[5]), dtype=dt, device=dev) for g in geo]
        lvs = []
        for l in range(n):
            B, L, Lo, M, ld, Cout = geo[l][:6]
            up = outs[l + 1] if (chain_up and l + 1 < n) else None
            lv = dict(raw=raws[l], ld_raw=Cout, ss=sss[l], out=outs[l], ld_out=Cout, M=M, L=Lo, up=up, ld_up=Cout if up is not None else 0)
            if fused:
                lv.update(fin[l])
            lvs.append(lv)
        # conv -> BN -> ReLU of all blocks in ONE launch; with chain_up a workgroup of level l recomputes the rows of the coarser
        # levels it adds from their raw outputs and statistics inside that launch (drn_conv_bn_train, up_group)
        one_launch = fused and same_c and ops.conv_bn_train(descs, lvs, code, up_group=[l + 1 if (chain_up and l + 1 < n) else -1
                                                                                             for l in range(n)])
        if not one_launch:
            for l in range(n):                                 # the convs as their own launch leave per-slab statistics behind
                st = torch.empty(((geo[l][3] + 127) // 128, 2, geo[l][5]), dtype=torch.float32, device=dev) if training else None
                stats.append(st)
                descs[l].stats = ops._p(st)
                if training:
                    fin[l]["stats"] = st
                    if fused:
                        lvs[l]["stats"] = st
            ops.gemm_nt(descs, code)
            if fin and not fused:                              # every level has its own BatchNorm module: one launch
                for grp in ([fin] if same_c else [[f] for f in fin]):
                    ops.bn_finalize_multi(grp, grp[0]["ss"].shape[1])
            apply = ops.bn_train_apply if fused else ops.bn_apply_multi
            halves = all(geo[l][2] == 2 * geo[l + 1][2] and geo[l][0] == geo[l + 1][0] for l in range(n - 1))
            if chain_up and fused and same_c and n <= 3 and halves and all(g[2] % 4 == 0 for g in geo[:max(n - 2, 0)]):
                # the whole top-down chain in ONE launch: a level whose `up` is another level's output of the same launch
                # recomputes the rows it adds from that level's raw rows and statistics (drn_bn_train_apply) -- same bits as
                # the coarse-to-fine order, two launches fewer
                apply(lvs, geo[0][5], code)
            elif chain_up or not same_c:
                for l in range(n - 1, -1, -1):                # coarse to fine: out_l reads out_{l+1}, one launch per level
                    apply([lvs[l]], geo[l][5], code)
            else:
                apply(lvs, geo[0][5], code)
        if relu_tap is not None:
            for l in range(n):
                _tap_relu(weights[l], 0, outs[l], outs[l + 1] if (chain_up and l + 1 < n) else None)
        ctx.meta, ctx.geo = meta, geo
        ctx.beta_refs = betas
        ctx.save_for_backward(*weights, *gammas, *xs, *raws, *sss, *saves)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        meta, geo = ctx.meta, ctx.geo
        n, dt, chain_up, strides = meta["n"], meta["dtype"], meta["chain_up"], meta["strides"]
        if not meta["training"]:
            raise DrnError("backward through an eval-mode BatchNorm block is not supported")
        code = code_of(dt)
        sv = ctx.saved_tensors
        weights, gammas, xs = sv[0:n], sv[n:2 * n], sv[2 * n:3 * n]
        raws, sss, saves = sv[3 * n:4 * n], sv[4 * n:5 * n], sv[5 * n:6 * n]
        dev = xs[0].device
        dtot = [None] * n
        g3 = [_grad_nlc(g, None, dt) for g in gouts] if (chain_up and n == 3) else None
        if (g3 is not None and all(g is not None for g in g3) and geo[0][5] == geo[1][5] == geo[2][5] and geo[1][2] * 2 == geo[0][2]
                and geo[2][2] * 2 == geo[1][2]):
            # the usual case -- three levels, every level with a gradient of its own: both upsample-add backward steps in ONE launch
            C3 = geo[0][5]
            dtot[0] = g3[0]
            dtot[1] = torch.empty((geo[1][0], geo[1][2], C3), dtype=dt, device=dev)
            dtot[2] = torch.empty((geo[2][0], geo[2][2], C3), dtype=dt, device=dev)
            ops.pairsum_chain3(g3[0], g3[1], dtot[1], g3[2], dtot[2], geo[1][3], C3, code)
        for l in range(n if dtot[0] is None else 0):          # fine to coarse: out_l also fed out_{l-1} through the upsample
            B, L, Lo, M, ld, Cout = geo[l][:6]
            d = _grad_nlc(gouts[l], None, dt)
            if chain_up and l > 0:
                if d is not None:                 # the incoming gradient stays untouched: out-of-place add, no clone
                    own, d = d, torch.empty((B, Lo, Cout), dtype=dt, device=dev)
                    ops.pairsum_add_to(d, Cout, own, Cout, dtot[l - 1], Cout, M, Cout, code)
                else:
                    d = torch.empty((B, Lo, Cout), dtype=dt, device=dev)
                    ops.pairsum_add(d, Cout, dtot[l - 1], Cout, M, Cout, code, accumulate=False)
            elif d is None:
                d = torch.zeros((B, Lo, Cout), dtype=dt, device=dev)
            dtot[l] = d
        draws, dgammas, dbetas, blevels = [], [], [], []
        for l in range(n):
            B, L, Lo, M, ld, Cout = geo[l][:6]
            draw = torch.empty((B, Lo, Cout), dtype=dt, device=dev)
            dg, db = grad_buffer(gammas[l]), grad_buffer(ctx.beta_refs[l])
            blevels.append(dict(dout=dtot[l], ld_dout=Cout, raw=raws[l], ld_raw=Cout, ss=sss[l], save=saves[l], gamma=gammas[l],
                                draw=draw, ld_draw=Cout, dgamma=dg, dbeta=db, accumulate=False, M=M))
            draws.append(draw)
            dgammas.append(dg)
            dbetas.append(db)
        if all(g[5] == geo[0][5] for g in geo):
            ops.bn_bwd_multi(blevels, geo[0][5], code)
        else:
            for l in range(n):
                ops.bn_bwd_multi([blevels[l]], geo[l][5], code)
        dxs = [None] * n
        need = [ctx.needs_input_grad[1 + 3 * n + l] for l in range(n)]
        descs = []
        for l in range(n):
            if not need[l]:
                continue
            B, L, Lo, M, ld, Cout, Cin, k, pad = geo[l]
            dx = torch.empty((B, L, Cin), dtype=dt, device=dev)
            descs.append(ops.gemm_desc(draws[l], packed(weights[l], (1, 2, 0), code), dx, B * L, Cin, Cout, taps=k,
                                       stride=strides[l], pad=pad, mode=1, Lout=L, Lsrc=Lo))
            dxs[l] = dx
        if descs:
            ops.gemm_nt(descs, code)
        dWs = [grad_buffer(weights[l]) for l in range(n)]
        wdescs = [ops.wgrad_desc(draws[l], xs[l], geo[l][3], Lout=geo[l][2], Lsrc=geo[l][1], ldy=geo[l][5], ldx=geo[l][4])
                  for l in range(n)]
        # same (Cout, k, stride) -- Cin may differ (the 1x1 laterals: 256 / 512 / 1024 input channels)
        same = all((geo[l][5], geo[l][7], geo[l][8]) == (geo[0][5], geo[0][7], geo[0][8]) and strides[l] == strides[0] for l in range(n))
        keep = (draws, xs)
        dWj = [_alias(w) for w in dWs]

        def wgrads():
            if same and n > 1 and n * max(g[3] for g in geo) <= 4 * sum(g[3] for g in geo):
                # one launch (+ one reduce) for all levels' weight gradients instead of one pair per level
                cins = [g[6] for g in geo]
                ops.gemm_wgrad_multi(wdescs, dWj, geo[0][5], cins[0] if len(set(cins)) == 1 else cins, taps=geo[0][7],
                                     stride=strides[0], pad=geo[0][8], w_layout=1, dtype=code)
            else:
                for l in range(n):
                    B, L, Lo, M, ld, Cout, Cin, k, pad = geo[l]
                    ops.gemm_wgrad([wdescs[l]], dWj[l], Cout, Cin, taps=k, stride=strides[l], pad=pad, w_layout=1, dtype=code)
            return keep
        _defer(wgrads, *dWj)
        return (None,) + tuple(dWs) + tuple(dgammas) + tuple(dbetas) + tuple(dxs)


def multi_conv_block(xs, blocks, training, dtype, chain_up=False):
    """blocks: list of (conv, bn) parameter holders (bias-free convs), one per input."""
    n = len(xs)
    meta = {"n": n, "dtype": dtype, "training": training, "chain_up": chain_up, "bns": [b for _, b in blocks],
            "strides": [c.stride[0] for c, _ in blocks]}
    args = [c.weight for c, _ in blocks] + [b.weight for _, b in blocks] + [b.bias for _, b in blocks] + list(xs)
    return list(_MultiConvFn.apply(meta, *args))


class InputPrep(object):
    """Everything the input stage needs that does not depend on the query: the feature tensor in the compute dtype (and, for
    the bf16 weight-gradient product, its transpose), the proposal position features, the GEMM copy of the prop_fc weight
    (warmed into the caches).  Non-differentiable."""
    __slots__ = ("xc", "xcT", "pf", "wfc", "dtype", "dims", "fc", "Z", "G0")


def input_prep(feats, props_start_end, prop_fc, dtype, want_wgrad=True, split_gate=False, position_transform=None):
    """props_start_end: (B, T, 2) proposal boundaries, or the (B, T, 3) position features themselves.
    split_gate: also run the prop_fc GEMM here, WITHOUT the query gate (-> pr.Z, the pre-gate value backward keeps anyway); the
    input stage then applies the gate in a pass of its own (ops.gate_fwd).  For schedules that run the query encoder beside this
    GEMM (drn_amd.graph.ForkedStep).  In bf16 the gated value is then bf16(bf16(z) * gate) instead of bf16(z * gate)."""
    code = code_of(dtype)
    B, T, D = feats.shape
    pr = InputPrep()
    pr.dtype, pr.dims = dtype, (B, T, D)
    xc = feats.contiguous()
    # bf16 training: the prop_fc weight gradient runs as an NT product of K-major operands (see backward); the
    # transposed copy of the features is written by the same pass that casts them
    nt_wgrad = code == ops.BF16 and NT_WGRAD and (B * T) % 8 == 0 and D % 8 == 0 and want_wgrad
    pr.xcT = None
    if nt_wgrad and xc.dtype == torch.float32:
        xc2, pr.xcT = ops.cast_transpose(xc.view(B * T, D), code)
        xc = xc2.view(B, T, D)
    elif nt_wgrad and xc.dtype == dtype:
        # features handed over in the compute dtype (drn_amd.data.collate_data(feature_dtype=...): rounded on the host by the
        # same rule): only the K-major copy is left to produce
        pr.xcT = ops.transpose2d(xc.view(B * T, D), code)
    elif xc.dtype != dtype:
        xc = ops.cast(xc.float(), code)
    pr.xc = xc
    Wfc = prop_fc.weight
    pr.wfc = Wfc.detach() if code == ops.F32 else packed(Wfc, (0, 2, 1), code)
    if TOUCH_W and code == ops.BF16 and _fc_kernel_kind(B, T, D, Wfc.shape[0], xc, pr.wfc, code, split_gate) != ops.NT_KIND_W4:
        ops.touch(pr.wfc)                                  # 2.9 ms old and evicted: ~8 us here saves the GEMM ~29 us
        # (not when the product runs on gemm_nt_w4_kernel, whose ring keeps 1.5 K-steps of loads in flight: 2.042 ms per step
        # without the touch against 2.051 with it, three rounds in one process)
        # (doing the same for the other forward weight copies -- 20 MB in a handful of launches -- measured 10 us SLOWER)
    # main_model.py:51-55: [start, end, end-start] in fp64, then float(); only level 0 is consumed (backbone.py:31)
    if props_start_end.shape[-1] == 3:                     # already [start, end, end-start]
        pf = props_start_end.float()
    elif props_start_end.dtype in (torch.float64, torch.float32):
        pf = ops.pos_feat(props_start_end)                 # one launch instead of sub + cat + cast
    else:
        duration = (props_start_end[:, :, 1] - props_start_end[:, :, 0]).unsqueeze(-1)
        pf = torch.cat((props_start_end, duration), dim=-1).float()
    pr.pf = pf.reshape(B * T, 3).contiguous()
    pr.Z = None
    pr.G0 = None
    if position_transform is not None and POS_EARLY:
        # conv0's input buffer, its position-embedding columns filled HERE: nothing of this depends on the query, so in the
        # two-branch step the launch sits beside the query encoder instead of between the prop_fc GEMM and conv0 (-7 us on that path)
        P = position_transform.weight.shape[0]
        pr.G0 = torch.empty((B, T, D + P), dtype=dtype, device=xc.device)
        ops.pos_embed_fwd(pr.pf, position_transform.weight.detach(), position_transform.bias.detach(), pr.G0.view(B * T, D + P)[:, D:], D + P,
                          B * T, P, code)
    if split_gate:
        pr.Z = torch.empty((B, T, D), dtype=dtype, device=xc.device)
        ops.gemm_nt([ops.gemm_desc(pr.xc, pr.wfc, pr.Z, B * T, D, D, Lout=T, ldc=D, bias=prop_fc.bias.detach())], code)
    return pr


def _fc_kernel_kind(B, T, D, N, xc, wfc, code, split_gate):
    """Which kernel the prop_fc product of the input stage will run on, asked of the library (drn_gemm_nt_plan) with a descriptor
    of the launch's shape and alignment class -- the rule (tile threshold, drn_tune switches, stride / alignment conditions)
    lives in gemm_nt.hip only.  No launch, one ctypes call."""
    P = 0 if split_gate else 256
    # (the output / pre-gate copy are allocated later: any 16-byte aligned address with the real row strides stands in)
    d = ops.gemm_desc(xc, wfc, xc, B * T, N, D, Lout=T, ldc=D + P, C2=None if split_gate else xc, ldc2=D)
    return ops.gemm_nt_plan([d], code)


POS_EARLY = os.environ.get("DRN_POS_EARLY", "1") != "0"              # (experiment switch: 0 = the position embedding after the prop_fc GEMM)
GATE_BN_FUSE = os.environ.get("DRN_GATE_BN_FUSE", "1") != "0"        # (experiment switch: 0 = drn_gate_bwd in front of the BatchNorm backward)
GATE_BWD_FUSE = os.environ.get("DRN_GATE_BWD_FUSE", "1") != "0"      # (experiment switch: 0 = drn_gate_bwd_t as a launch of its own)


class EmbedTail(object):
    """Link between the input stage and the conv block that consumes its output.  The position embedding occupies the last P
    channels of that conv's input and is a Linear(3, P) of per-row features, so its two gradients can be taken through the conv
    from the gradient at the conv's OUTPUT (ops.conv_tail_bwd) and the conv's input-gradient GEMM leaves the P columns out
    (T = 256: 544 -> 512 tiles of 256x256, two full rounds on 256 CUs instead of two and an eighth).  The conv block's backward
    fills dW / db; the input stage's backward, which runs after it, hands them to autograd instead of reducing the
    (unwritten) embedding columns of its incoming gradient."""
    __slots__ = ("pf", "Wpos", "bpos", "P", "dW", "db", "dtype", "gate_ctx", "gate_out")

    def __init__(self, pf, Wpos, bpos, dtype):
        self.pf, self.Wpos, self.bpos, self.P, self.dtype = pf, Wpos, bpos, Wpos.shape[0], dtype
        self.dW = self.db = None
        # the input stage's gate backward inside the conv block's data-gradient launch (DrnGemmDesc::gb_*): the input stage leaves
        # (Z, gate0, B, T, D) here in its forward; the conv block's backward, if its launch runs on gemm_nt_w4c_kernel, leaves
        # (dZT, dgate, dsum) -- and the input stage's backward does not read its incoming gradient at all
        self.gate_ctx = self.gate_out = None

    def usable(self, Cin, Cout, dt):
        vn = 8 if dt == torch.bfloat16 else 4
        return dt == self.dtype and Cin > self.P and Cout % vn == 0 and self.dW is None

    def backward(self, draw, wd, B, L, Lo, Cout, Cin, k, stride, pad, code):
        dW, db = grad_buffer(self.Wpos), grad_buffer(self.bpos)
        ops.conv_tail_bwd(draw, Cout, B, Lo, Cout, wd[Cin - self.P:], k * Cout, k, stride, pad, self.pf, L, self.P, dW, db, code)
        self.dW, self.db = dW, db

    def take(self):
        dW, db, self.dW, self.db = self.dW, self.db, None, None
        return dW, db

    def take_gate(self):
        out, self.gate_out = self.gate_out, None
        return out


class _InputStageFn(torch.autograd.Function):
    """prop_fc + level-0 query gating + position embedding, written into one (B, T, D+P) buffer that is conv0's
    input (model/main_model.py:51-59,67 + model/backbone.py:28-32: Linear, `q * x`, cat) -- one MFMA GEMM whose
    epilogue adds the bias, keeps the pre-gate value for backward and applies the gate, plus one tiny kernel."""

    @staticmethod
    def forward(ctx, prep, Wfc, bfc, gate0, Wpos, bpos, tail):
        dtype = prep.dtype
        code = code_of(dtype)
        B, T, D = prep.dims
        P = Wpos.shape[0]
        xc, xcT, pf, wfc = prep.xc, prep.xcT, prep.pf, prep.wfc
        dev = xc.device
        G0 = getattr(prep, "G0", None)                     # (input_prep may have made it, position columns filled: used once)
        pos_done = G0 is not None and tuple(G0.shape) == (B, T, D + P)
        if pos_done:
            prep.G0 = None
        else:
            G0 = torch.empty((B, T, D + P), dtype=dtype, device=dev)
        if getattr(prep, "Z", None) is not None:
            Z = prep.Z                                    # the GEMM already ran, un-gated (input_prep(split_gate=True))
            ops.gate_fwd(Z, D, gate0, G0, D + P, B, T, D, code)
        else:
            Z = torch.empty((B, T, D), dtype=dtype, device=dev)
            ops.gemm_nt([ops.gemm_desc(xc, wfc, G0, B * T, D, D, Lout=T, ldc=D + P, bias=bfc, gate=gate0, ldg=gate0.stride(0),
                                       C2=Z, ldc2=D)], code)
        if not pos_done:
            pos_slice = G0.view(B * T, D + P)[:, D:]
            ops.pos_embed_fwd(pf, Wpos, bpos, pos_slice, D + P, B * T, P, code)
        ctx.dtype, ctx.dims = dtype, (B, T, D, P)
        ctx.param_refs = (Wfc, bfc, Wpos, bpos)
        ctx.tail = tail
        if tail is not None:
            fuse = GATE_BWD_FUSE and dtype == torch.bfloat16 and xcT is not None and xcT.numel() and T in (32, 64, 128, 256) and D % 256 == 0
            tail.gate_ctx = (Z, gate0, B, T, D) if fuse else None
            tail.gate_out = None
        ctx.save_for_backward(xc, pf, gate0, Z, xcT if xcT is not None else xc.new_empty(0))
        return G0

    @staticmethod
    def backward(ctx, dG0):
        dtype = ctx.dtype
        code = code_of(dtype)
        B, T, D, P = ctx.dims
        xc, pf, gate0, Z, xcT = ctx.saved_tensors
        dev = xc.device
        fused = ctx.tail.take_gate() if ctx.tail is not None else None
        if fused is None:
            dG0 = _grad_nlc(dG0, None, dtype)
            dgate = torch.empty((B, D), dtype=torch.float32, device=dev)
            dsum = torch.empty((B, D), dtype=torch.float32, device=dev)      # per-clip column sums of dZ: prop_fc bias gradient
        Wfc, bfc, Wpos, bpos = ctx.param_refs
        dW, db = grad_buffer(Wfc), grad_buffer(bfc)
        dWp, dbp = ctx.tail.take() if ctx.tail is not None else (None, None)
        from_tail = dWp is not None            # the conv block that read G0 has already produced them (EmbedTail)
        if not from_tail:
            dWp, dbp = grad_buffer(Wpos), grad_buffer(bpos)
        dZ = dZT = None
        if fused is not None:
            dZT, dgate, dsum = fused         # conv0's data-gradient launch did the gate backward in its epilogue (dG0 was never written)
        elif xcT.numel() and T % 32 == 0:
            # dW[n][c] = sum_m dZ[m][n] * x[m][c] as an NT product of the K-major copies dZ^T (D, B*T) and x^T (D, B*T):
            # the NT kernel streams both operands with 16-byte LDS reads (1.1 PFLOP/s on this shape), while the TN
            # kernel's transposing ds_read_b64_tr_b16 fragments hold it to ~0.7.  dZ is only ever needed transposed.
            dZT = torch.empty((D, B * T), dtype=dtype, device=dev)
            ops.gate_bwd_t(dG0, D + P, Z, D, gate0, dZT, dgate, B, T, D, code, dsum=dsum)
        else:
            dZ = torch.empty((B, T, D), dtype=dtype, device=dev)
            ops.gate_bwd(dG0, D + P, Z, D, gate0, dZ, D, None, 0, dgate, B, T, D, code, dsum=dsum)

        # 2. the parameter gradients (deferrable: only the optimizer reads them)
        jW, jb, jWp, jbp = _alias(dW), _alias(db), _alias(dWp), _alias(dbp)

        def wgrads():
            if dZT is not None:
                d = ops.gemm_desc(dZT, xcT, jW, D, D, B * T, out_f32=True)
                if _sumsq_want and ops.gemm_nt_plan([d], code) == ops.NT_KIND_W4:
                    # the GEMM's epilogue leaves sum(dW^2) per output tile: the optimizer's norm pass does not read these 67 MB again
                    part = ops.persistent_buffer(("prop_fc_wgrad", jW.data_ptr()), (D // 256) * (D // 256), jW.device)
                    d.sumsq = ops._p(part)
                    note_sumsq([(jW.data_ptr(), jW.numel())], part)
                ops.gemm_nt([d], code)
            elif xcT.numel():
                dZt = ops.transpose2d(dZ.view(B * T, D), code)      # (a descriptor holds raw pointers: keep the operand alive)
                ops.gemm_nt([ops.gemm_desc(dZt, xcT, jW, D, D, B * T, out_f32=True)], code)
            else:
                ops.gemm_wgrad([ops.wgrad_desc(dZ, xc, B * T, ldy=D, ldx=D)], jW, D, D, taps=1, w_layout=0, dtype=code)
            ops.colsum(dsum, D, B, D, jb, ops.F32)
            if not from_tail:
                ops.pos_embed_bwd(dG0.view(B * T, D + P)[:, D:], D + P, pf, B * T, P, jWp, jbp, code)
        _defer(wgrads, jW, jb, jWp, jbp)
        return None, dW, db, dgate, dWp, dbp, None


def input_stage(prep, prop_fc, gate0, position_transform, with_tail=False):
    """with_tail: also return the EmbedTail to hand to the conv block that reads the result (conv_block(..., tail=))."""
    tail = EmbedTail(prep.pf, position_transform.weight, position_transform.bias, prep.dtype) if with_tail else None
    g0 = _InputStageFn.apply(prep, prop_fc.weight, prop_fc.bias, gate0, position_transform.weight, position_transform.bias, tail)
    return (g0, tail) if with_tail else g0


class _HeadOutFn(torch.autograd.Function):
    """Per-location output convs with 1-2 channels over all pyramid levels: `nheads` heads, head h reading the
    column slice [col[h], col[h]+C) of every level tensor (cls_logits / bbox_pred on the two tower halves,
    model/fcos.py:96-100; or the final iou_scores conv, fcos.py:68,102).  Outputs are fp32 (R, N_h), rows
    ordered level-first / clip-major (the order model/loss.py:150-166 flattens to)."""

    @staticmethod
    def forward(ctx, meta, *args):
        nheads, nl, dtype = meta["nheads"], meta["nl"], meta["dtype"]
        code = code_of(dtype)
        heads = [args[3 * h:3 * h + 3] for h in range(nheads)]           # (W, bias, scales|None)
        xs = args[3 * nheads:]
        dev = xs[0].device
        geo = [geom(x) for x in xs]
        R = sum(g[0] * g[1] for g in geo)
        outs, zs, calls = [], [], []
        for h, (W, bias, scales) in enumerate(heads):
            N, C, taps = W.shape
            c0 = meta["cols"][h]
            xsl = [(x[:, :, c0:c0 + C], g[3], g[0] * g[1], g[1]) for x, g in zip(xs, geo)]
            groups = ops.head_groups(xsl, scales=scales)
            out = torch.empty((R, N), dtype=torch.float32, device=dev)
            z = torch.empty((R, N), dtype=torch.float32, device=dev) if scales is not None else None
            calls.append(dict(groups=groups, W=packed(W, (0, 2, 1), ops.F32), bias=bias, N=N, C=C, taps=taps,
                              exp_mode=scales is not None, out=out, z=z))             # W as [N][taps][C] (cached copy)
            outs.append(out)
            zs.append(z if z is not None else out.new_empty(0))
        ops.heads_fwd(calls, code)                       # the heads of one call share launches (two per launch)
        ctx.meta, ctx.geo, ctx.R = meta, geo, R
        ctx.head_w = [h[0] for h in heads]           # the Parameter objects (`packed` caches per parameter, not per saved tensor)
        ctx.save_for_backward(*args, *outs, *zs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        meta, geo, R = ctx.meta, ctx.geo, ctx.R
        nheads, nl, dtype = meta["nheads"], meta["nl"], meta["dtype"]
        code = code_of(dtype)
        sv = ctx.saved_tensors
        nargs = 3 * nheads + nl
        args, outs, zs = sv[:nargs], sv[nargs:nargs + nheads], sv[nargs + nheads:]
        xs = args[3 * nheads:]
        dev = xs[0].device
        width = xs[0].shape[2]
        covered = sorted((meta["cols"][h], meta["cols"][h] + args[3 * h].shape[1]) for h in range(nheads))
        full = covered[0][0] == 0 and covered[-1][1] == width and all(covered[i][1] == covered[i + 1][0]
                                                                      for i in range(len(covered) - 1))
        alloc = torch.empty if full else torch.zeros
        dxs = [alloc((g[0], g[1], g[2]), dtype=dtype, device=dev) for g in geo]
        grads, calls = [], []
        for h in range(nheads):
            W, bias, scales = args[3 * h:3 * h + 3]
            N, C, taps = W.shape
            c0 = meta["cols"][h]
            dout = douts[h]
            if dout is not None:                         # the kernels write every element of the three gradients
                dW, db = grad_buffer(W), grad_buffer(bias)
                dsc = grad_buffer(scales) if scales is not None else None      # (nl,), possibly strided: the sinks themselves
            else:
                dW, db = torch.zeros_like(W), torch.zeros_like(bias)
                dsc = torch.zeros(nl, dtype=torch.float32, device=dev) if scales is not None else None
            if dout is not None:
                dout = dout.contiguous().float()
                xsl = [(x[:, :, c0:c0 + C], g[3], g[0] * g[1], g[1]) for x, g in zip(xs, geo)]
                dsl = [dx[:, :, c0:c0 + C] for dx in dxs]
                groups = ops.head_groups(xsl, dxs=dsl, scales=scales)
                # dX geometry must match X's: both are slices of equally wide buffers
                for x, dx in zip(xs, dxs):
                    assert x.stride(1) == dx.stride(1), "head input must be contiguous in backward"
                calls.append(dict(groups=groups, W=packed(ctx.head_w[h], (0, 2, 1), ops.F32), dout=dout, out=outs[h],
                                  z=zs[h] if scales is not None else None, N=N, C=C,
                                  taps=taps, exp_mode=scales is not None, accumulate_dx=False, dW=dW, dbias=db, dscale=dsc))
            elif full:
                for dx in dxs:
                    dx[:, :, c0:c0 + C].zero_()
            grads += [dW, db, dsc]
        if calls:
            ops.heads_bwd(calls, code)
        return (None,) + tuple(grads) + tuple(dxs)


def head_out(xs, heads, cols, dtype):
    """heads: list of (conv_module, scales_tensor_or_None); cols: first input channel of each head."""
    meta = {"nheads": len(heads), "nl": len(xs), "dtype": dtype, "cols": list(cols)}
    args = []
    for conv, scales in heads:
        args += [conv.weight, conv.bias, scales]
    return _HeadOutFn.apply(meta, *args, *xs)


class _FCOSLossFn(torch.autograd.Function):
    """Target assignment + focal / IoU / IoU-score losses (model/loss.py:40-239) in one kernel each way.
    Returns (loss_cls, loss_reg, loss_iou, counts2, total): each loss of shape (1,), counts2 = [n_pos, n_iou_pos], total =
    the sum of the three losses (main.py:225), shape (1,); all are views of one 6-float result buffer (no per-loss slicing or
    summing kernels in either direction).  The BatchNorm step counters the forward pass owes (bump_bn_counter) ride along."""

    @staticmethod
    def forward(ctx, meta, logits, reg, iou, gt):
        levels = ops.loss_levels(meta["levels"])
        B = meta["B"]
        out6 = torch.empty(6, dtype=torch.float32, device=logits.device)
        logits, reg = logits.contiguous(), reg.contiguous()
        iou = iou.contiguous() if iou is not None else None
        gt = gt.contiguous()
        if gt.dtype not in (torch.float32, torch.float64):
            gt = gt.float()
        ops.fcos_loss_fwd(levels, B, logits, reg, iou, gt, meta["gamma"], meta["alpha"], meta["target_scale"],
                          meta["iou_stage"], out6, bumps=take_bn_counters(logits.device))
        ctx.meta = meta
        ctx.save_for_backward(logits, reg, iou if iou is not None else logits.new_empty(0), gt, out6)
        counts = out6[3:5]
        ctx.mark_non_differentiable(counts)
        ctx.set_materialize_grads(False)           # unused outputs arrive as None in backward, not as zero-filled tensors
        return out6[0:1], out6[1:2], out6[2:3], counts, out6[5:6]

    @staticmethod
    def backward(ctx, g_cls, g_reg, g_iou, _gc, g_tot):
        if g_tot is not None:                      # gradient of the total: the same scalar for each of the three losses
            g_cls, g_reg, g_iou = (g_tot if g is None else g + g_tot for g in (g_cls, g_reg, g_iou))
        meta = ctx.meta
        logits, reg, iou, gt, out6 = ctx.saved_tensors
        levels = ops.loss_levels(meta["levels"])
        has_iou = bool(meta["iou_stage"])
        dlogits = torch.empty_like(logits)
        dreg = torch.empty_like(reg)
        diou = torch.empty_like(iou) if has_iou else None
        ops.fcos_loss_bwd(levels, meta["B"], logits, reg, iou if has_iou else None, gt, meta["gamma"], meta["alpha"],
                          meta["target_scale"], meta["iou_stage"], out6,
                          [None if g is None else g.contiguous().float() for g in (g_cls, g_reg, g_iou)], dlogits, dreg, diou)
        return None, dlogits, dreg, diou, None


class LossDict(dict):
    """The reference's loss dict (same keys) that also carries the sum of the three losses as computed by the loss kernel
    (main.py:222-225), so that the trainer's `sum(loss_dict.values())` costs no launch."""
    total = None


def loss_total(losses):
    """sum(loss_dict.values()) of the reference's training loop (main.py:225) without a launch when the dict came from this
    package; any other mapping is summed the reference's way."""
    t = getattr(losses, "total", None)
    return t.reshape(()) if t is not None else sum(l for l in losses.values())


_ones = {}


def backward(loss):
    """loss.backward() with a cached unit gradient (autograd otherwise fills a fresh ones_like(loss) per step)."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _ones.get(key)
    if one is None:
        one = _ones[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
    loss.backward(gradient=one)


def fcos_loss(logits, reg, iou, gt, levels, B, gamma, alpha, target_scale, iou_stage):
    meta = {"levels": levels, "B": B, "gamma": float(gamma), "alpha": float(alpha), "target_scale": float(target_scale),
            "iou_stage": int(bool(iou_stage))}
    return _FCOSLossFn.apply(meta, logits, reg, iou if iou_stage else None, gt)


# ---------------------------------------------------------------------------------------------------------------------
# Query side (model/language_module.py:38-63, model/main_model.py:36-50).  Every dense product here has a 32-clip batch
# dimension: forward / input gradients go through the grouped skinny kernel, all weight / bias gradients of one backward
# node through ONE grouped outer-product launch (drn_amd/csrc/qdense.hip) -- no library GEMM, no per-bias reduction launch.
# ---------------------------------------------------------------------------------------------------------------------
def _lstm_lowp(lowp, H):
    """bf16 recurrent weights (drn_lstm_step_*'s DRN_BF16 mode) need H % 128 == 0; smaller test models stay on the fp32 kernels."""
    return bool(lowp) and H % 128 == 0


def _lstm_forward(emb_tm, lens, lstm_params, B, L, qvec=None, lowp=False):
    """emb_tm (L*B, E) time-major fp32.  Returns out (B, L, 2H) and the tensors backward needs.  qvec (B, 4H) or None: filled
    with [out[b][0] ; out[b][len_b - 1]] by the step kernels (language_module.py:48-54)."""
    w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r = lstm_params
    H = w_hh_f.shape[1]
    dev = emb_tm.device
    wih_s = stacked([w_ih_f, w_ih_r])                                    # (8H, E): one input projection for both directions
    xproj = ops.skinny_rows(emb_tm, wih_s)                               # (L*B, 8H) = [L][B][2][4H]
    hseq = torch.empty((2, L + 1, B, H), dtype=torch.float32, device=dev)
    cseq = torch.empty((2, L + 1, B, H), dtype=torch.float32, device=dev)
    gates = torch.empty((L, B, 2, 4 * H), dtype=torch.float32, device=dev)
    hprev = torch.empty((L, B, 2, H), dtype=torch.float32, device=dev)
    out = torch.empty((B, L, 2 * H), dtype=torch.float32, device=dev)
    biases = (b_ih_f.detach(), b_hh_f.detach(), b_ih_r.detach(), b_hh_r.detach())
    whf, whr = w_hh_f.detach().contiguous(), w_hh_r.detach().contiguous()
    # lowp (the bf16 model): the steps keep an fp16 copy of the hidden state and multiply it on the fp16 MFMA (W_hh rounded to fp16 in
    # registers, fp32 accumulation, fp32 states / outputs): 64 KB of operands per workgroup instead of 96
    if _lstm_lowp(lowp, H) and ops.LSTM_SEQ and L > 1 and ops.lstm_seq_fwd(xproj, whf, whr, biases, hseq, cseq, gates, out, hprev, lens, B, L, H, qvec=qvec):
        return out, (cseq, gates, hprev)             # every step in one launch (same bits)
    hseq16 = torch.empty((2, L + 1, B, H), dtype=torch.float16, device=dev) if _lstm_lowp(lowp, H) else None
    for s in range(L):
        ops.lstm_step_fwd(xproj, whf, whr, biases, hseq, cseq, gates, out, hprev, lens, B, L, H, s, qvec=qvec, hseq16=hseq16)
    return out, (cseq, gates, hprev)


def _lstm_backward(dout, emb_tm, lens, lstm_params, saved, B, L, leaves, dqvec=None, lowp=False):
    """dout (B, L, 2H) contiguous fp32; dqvec (B, 4H) or None: gradient of the [first ; last] sentence vector, added to rows 0 and
    len_b - 1 of dout as the kernels read them.  Returns (demb_tm (L*B, E), the eight parameter gradients); the weight / bias
    gradient products are appended to `leaves` (the caller launches them with its own, ops.outer_wgrad); they land in the
    reducer's flat buckets when sinks are registered."""
    w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r = lstm_params
    cseq, gates, hprev = saved
    H = w_hh_f.shape[1]
    dev = dout.device
    wcode = ops.BF16 if _lstm_lowp(lowp, H) else ops.F32
    wtf, wtr = packed(w_hh_f, (1, 2, 0), wcode), packed(w_hh_r, (1, 2, 0), wcode)          # W_hh^T, cached
    dgates = torch.empty((L, B, 2, 4 * H), dtype=torch.float32, device=dev)
    scratch = torch.empty((2, 2, B, H), dtype=torch.float32, device=dev)
    # bf16 model: the cell backward also leaves a bf16 copy of the gate gradients for the next step's recurrent product
    dg16 = torch.empty((L, B, 2, 4 * H), dtype=torch.bfloat16, device=dev) if wcode == ops.BF16 else None
    ops.lstm_bwd_first(dout, gates, cseq, dgates, scratch[0], scratch[1], lens, B, L, H, dqvec=dqvec, dgates16=dg16)
    for s in range(L - 1, 0, -1):           # W_hh product of step s + cell backward of step s-1 in one launch
        ops.lstm_step_bwd(dout, gates, cseq, wtf, wtr, dgates, scratch[0], scratch[1], lens, B, L, H, s, dqvec=dqvec, dgates16=dg16)
    dg = dgates.view(L * B, 8 * H)
    hp = hprev.view(L * B, 2 * H)
    # dg [W_f; W_r]: (L*B, E); in the bf16 model from the gate gradients' bf16 copy (the rows are most of that product's bytes)
    demb_tm = ops.skinny_rows(dg16.view(L * B, 8 * H) if dg16 is not None else dg, stacked_t([w_ih_f, w_ih_r]))
    gr = [grad_buffer(p) for p in lstm_params]
    leaves += [dict(dY=dg[:, :4 * H], X=emb_tm, dW=gr[0], db=gr[2], db2=gr[3]), dict(dY=dg[:, :4 * H], X=hp[:, :H], dW=gr[1]),
               dict(dY=dg[:, 4 * H:], X=emb_tm, dW=gr[4], db=gr[6], db2=gr[7]), dict(dY=dg[:, 4 * H:], X=hp[:, H:], dW=gr[5])]
    return demb_tm, tuple(gr)


class _LinearFn(torch.autograd.Function):
    """nn.Linear on a batch-sized fp32 input: forward / input gradient on the skinny MFMA kernel, weight + bias gradient in
    one outer-product launch; parameter gradients land in the reducer's buckets."""

    @staticmethod
    def forward(ctx, x, W, b):
        x = x.float().contiguous()
        ctx.save_for_backward(x, W, b)
        return ops.skinny_linear(x, W.detach(), b.detach())

    @staticmethod
    def backward(ctx, dy):
        x, W, b = ctx.saved_tensors
        dy = dy.float().contiguous()
        dx = ops.skinny_linear(dy, packed(W, (1, 2, 0), ops.F32)) if ctx.needs_input_grad[0] else None
        dW, db = grad_buffer(W), grad_buffer(b)
        ops.outer_wgrad([dict(dY=dy, X=x, dW=dW, db=db)])
        return dx, dW, db


def linear(x, lin):
    """lin: nn.Linear parameter holder; x (M <= 64, K) fp32 on the GPU."""
    if x.shape[0] > 64 or lin.weight.shape[1] % 4:
        raise DrnError("drn_amd.functional.linear serves batch-sized inputs (<= 64 rows, K %% 4 == 0); got %s" % (tuple(x.shape),))
    return _LinearFn.apply(x, lin.weight, lin.bias)


def _dev_lengths(lengths, dev):
    if lengths.device != dev or lengths.dtype != torch.int64:
        lengths = lengths.to(device=dev, dtype=torch.int64)
    return lengths.contiguous()


class _BiLSTMFn(torch.autograd.Function):
    """Bidirectional 1-layer LSTM over padded sequences with device-side lengths (model/language_module.py:38-45:
    pack_padded_sequence -> nn.LSTM -> pad_packed_sequence).  The recurrence runs in drn_amd/csrc/lstm.hip, one launch per
    time step; the input projection and the weight gradients in drn_amd/csrc/qdense.hip."""

    @staticmethod
    def forward(ctx, lowp, emb, lengths, *lstm_params):
        B, L, E = emb.shape
        emb_tm = emb.detach().transpose(0, 1).reshape(L * B, E).float()
        lens = _dev_lengths(lengths, emb.device)
        out, saved = _lstm_forward(emb_tm, lens, lstm_params, B, L, lowp=lowp)
        ctx.lowp = lowp
        ctx.dims = (B, L, E)
        ctx.lstm_params = lstm_params
        ctx.save_for_backward(emb_tm, lens, *saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, L, E = ctx.dims
        emb_tm, lens = ctx.saved_tensors[:2]
        leaves = []
        demb_tm, grads = _lstm_backward(dout.contiguous().float(), emb_tm, lens, ctx.lstm_params, ctx.saved_tensors[2:], B, L, leaves,
                                        lowp=ctx.lowp)
        ops.outer_wgrad(leaves, lowp=ctx.lowp)
        return (None, demb_tm.view(L, B, E).transpose(0, 1), None) + grads


def bilstm(emb, lengths, lstm, lowp=False):
    """lstm: an nn.LSTM(num_layers=1, bidirectional=True, batch_first=True) used as a parameter holder.  lowp: recurrent
    products on the bf16 MFMA with bf16 copies of W_hh (needs hidden_size % 128 == 0, else the fp32 kernels run)."""
    return _BiLSTMFn.apply(bool(lowp), emb, lengths, *_lstm_param_list(lstm))


def _lstm_param_list(lstm):
    return (lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
            lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse, lstm.bias_hh_l0_reverse)


class _QueryEncoderFn(torch.autograd.Function):
    """The whole query side as one autograd node: embedding lookup, BiLSTM, [first ; last] sentence vector, qInput + ReLU,
    the three qInput{t} projections, the three attention commands (model/language_module.py:38-63 + 17-36) and -- when
    their parameters are passed -- mainModel's per-level gate projections qInput{t} (model/main_model.py:47-50), whose
    outputs then replace the commands as the node's outputs.  Forward: 8 + L launches; backward: 8 + L on the critical path
    plus ONE launch for all 13 (10) weight / bias gradient products.  Every parameter gradient is written by a kernel
    straight into its final buffer."""

    @staticmethod
    def forward(ctx, lowp, tokens, lengths, table, *params):
        lstm_params, (Wq, bq, W0, b0, W1, b1, W2, b2, wl, bl), gate_params = params[:8], params[8:18], params[18:]
        ctx.lowp = lowp
        B, L = tokens.shape
        E = table.shape[1]
        H = lstm_params[1].shape[1]
        C = 2 * H
        dev = table.device
        tokens = tokens.to(device=dev, dtype=torch.int64).contiguous()
        lens = _dev_lengths(lengths, dev)
        emb_tm = torch.empty((L * B, E), dtype=torch.float32, device=dev)
        ops.qe_embed_fwd(tokens, table.detach(), emb_tm, B, L, E)
        qvec = torch.empty((B, 2 * C), dtype=torch.float32, device=dev)
        out, saved = _lstm_forward(emb_tm, lens, lstm_params, B, L, qvec=qvec, lowp=lowp)     # qvec: language_module.py:48-54
        base = ops.skinny_linear(qvec, Wq.detach(), bq.detach(), relu=True)               # language_module.py:55-56
        _tap_relu(Wq, 0, base)
        qcmd = ops.skinny_linear(base, stacked([W0, W1, W2]), stacked([b0, b1, b2]))      # (B, 3*C): all three qInput{t}
        att = torch.empty((B, 3, L), dtype=torch.float32, device=dev)
        cmds = torch.empty((3, B, C), dtype=torch.float32, device=dev)
        ops.qe_attn_fwd(out, qcmd, wl.detach(), bl.detach(), lens, att, cmds, B, L, C)
        ctx.dims = (B, L, E, H)
        ctx.params = params
        ctx.table = table
        ctx.save_for_backward(tokens, lens, emb_tm, out, qvec, base, qcmd, att, cmds, *saved)
        if not gate_params:
            return cmds[0], cmds[1], cmds[2]
        # main_model.py:47-50: gate_t = qInput{t}(cmd_t), the three levels in one launch
        return tuple(ops.skinny_group([dict(X=cmds[t], W=gate_params[2 * t].detach(), bias=gate_params[2 * t + 1].detach())
                                       for t in range(3)]))

    @staticmethod
    def backward(ctx, d0, d1, d2):
        B, L, E, H = ctx.dims
        C = 2 * H
        tokens, lens, emb_tm, out, qvec, base, qcmd, att, cmds = ctx.saved_tensors[:9]
        params = ctx.params
        lstm_params, (Wq, bq, W0, b0, W1, b1, W2, b2, wl, bl), gate_params = params[:8], params[8:18], params[18:]
        dev = out.device
        f32 = dict(dtype=torch.float32, device=dev)
        douts = [None if d is None else d.contiguous().float() for d in (d0, d1, d2)]
        leaves, gate_grads = [], ()
        if gate_params:
            live = [t for t in range(3) if douts[t] is not None]
            dc = ops.skinny_group([dict(X=douts[t], W=packed(gate_params[2 * t], (1, 2, 0), ops.F32)) for t in live])
            dcmds = [None] * 3
            for t, d in zip(live, dc):
                dcmds[t] = d
            gg = []
            for t in range(3):
                Wg, bg = gate_params[2 * t], gate_params[2 * t + 1]
                if douts[t] is None:
                    gg += [torch.zeros_like(Wg), torch.zeros_like(bg)]
                else:
                    dWg, dbg = grad_buffer(Wg), grad_buffer(bg)
                    leaves.append(dict(dY=douts[t], X=cmds[t], dW=dWg, db=dbg))
                    gg += [dWg, dbg]
            gate_grads = tuple(gg)
        else:
            dcmds = douts
        dqcmd, dout = torch.empty((B, 3 * C), **f32), torch.empty((B, L, C), **f32)
        dw_part, db_part = torch.empty((B, C), **f32), torch.empty((B, 1), **f32)
        ops.qe_attn_bwd(dcmds, att, out, qcmd, wl.detach(), lens, dqcmd, dout, dw_part, db_part, B, L, C)
        dwl, dbl = grad_buffer(wl), grad_buffer(bl)
        leaves += [dict(dY=dw_part, db=dwl.view(-1)), dict(dY=db_part, db=dbl.view(-1))]
        # qInput{t}: q_cmd_t = base W_t^T + b_t ; then qInput's ReLU as the mask of the same launch
        dpre = ops.skinny_linear(dqcmd, stacked_t([W0, W1, W2]), mask=base)
        dW = [grad_buffer(W) for W in (W0, W1, W2)]
        db = [grad_buffer(b) for b in (b0, b1, b2)]
        leaves += [dict(dY=dqcmd[:, t * C:(t + 1) * C], X=base, dW=dW[t], db=db[t]) for t in range(3)]
        dqvec = ops.skinny_linear(dpre, packed(Wq, (1, 2, 0), ops.F32))
        dWq, dbq = grad_buffer(Wq), grad_buffer(bq)
        leaves.append(dict(dY=dpre, X=qvec, dW=dWq, db=dbq))
        demb_tm, lstm_grads = _lstm_backward(dout, emb_tm, lens, lstm_params, ctx.saved_tensors[9:], B, L, leaves, dqvec=dqvec,
                                             lowp=ctx.lowp)
        table = ctx.table
        dtable = grad_buffer(table)
        ops.qe_embed_bwd(tokens, demb_tm, dtable, B, L, E, table.shape[0], 0)      # nn.Embedding(padding_idx=0)
        ops.outer_wgrad(leaves, lowp=ctx.lowp)                                     # every weight / bias gradient of the node
        return (None, None, None, dtable) + lstm_grads + (dWq, dbq, dW[0], db[0], dW[1], db[1], dW[2], db[2], dwl, dbl) + gate_grads


def query_encoder(tokens, lengths, enc, gate_linears=None, lowp=False):
    """enc: drn_amd.model.language_module.QueryEncoder (parameter holder).  Returns the three (B, 2H) commands, or -- with
    gate_linears = mainModel's three qInput{t} nn.Linear holders -- the three per-level gate tensors (B, C_t).  lowp (the bf16
    model): the BiLSTM's forward recurrence multiplies an fp16 copy of the hidden state by W_hh rounded to fp16 on the fp16 MFMA
    (fp32 accumulation, fp32 cell state and outputs; `_lstm_forward`), the backward recurrence uses the bf16 copy of W_hh^T and
    bf16-rounded gate gradients on the bf16 MFMA (`_lstm_backward`); the dense forward products stay exact fp32."""
    if enc.embedding.padding_idx != 0:
        raise DrnError("query encoder kernels assume nn.Embedding(padding_idx=0) (model/language_module.py:13)")
    extra = []
    for lin in (gate_linears or []):
        extra += [lin.weight, lin.bias]
    return _QueryEncoderFn.apply(bool(lowp), tokens, lengths, enc.embedding.weight, *_lstm_param_list(enc.biLSTM),
                                 enc.qInput.weight, enc.qInput.bias, enc.qInput0.weight, enc.qInput0.bias,
                                 enc.qInput1.weight, enc.qInput1.bias, enc.qInput2.weight, enc.qInput2.bias,
                                 enc.cmd_inter2logits.weight, enc.cmd_inter2logits.bias, *extra)


class _GateProjFn(torch.autograd.Function):
    """mainModel's three per-level gate projections gate_t = qInput{t}(cmd_t) (model/main_model.py:47-50) as a node of their own:
    the same two launches `_QueryEncoderFn` spends on them when they ride inside it (one grouped skinny product each way, their
    weight / bias gradients in one outer-product launch), but with the commands as a cut point in the autograd graph -- the
    multi-GPU step (drn_amd.graph.TwoPhaseStep, four phases) sends the projections' 20 MB of gradients on their way while the
    query encoder's backward is still replaying."""

    @staticmethod
    def forward(ctx, lowp, c0, c1, c2, W0, b0, W1, b1, W2, b2):
        cmds = [c.contiguous().float() for c in (c0, c1, c2)]
        ctx.lowp = bool(lowp)                          # (the bf16 model: weight-gradient operands rounded to bf16, as inside the fused node)
        ctx.save_for_backward(*cmds, W0, b0, W1, b1, W2, b2)
        return tuple(ops.skinny_group([dict(X=cmds[t], W=W.detach(), bias=b.detach()) for t, (W, b) in enumerate(((W0, b0), (W1, b1), (W2, b2)))]))

    @staticmethod
    def backward(ctx, d0, d1, d2):
        sv = ctx.saved_tensors
        cmds, Ws, bs = sv[:3], sv[3::2], sv[4::2]
        douts = [None if d is None else d.contiguous().float() for d in (d0, d1, d2)]
        live = [t for t in range(3) if douts[t] is not None]
        dc = ops.skinny_group([dict(X=douts[t], W=packed(Ws[t], (1, 2, 0), ops.F32)) for t in live]) if live else []
        dcmds = [None] * 3
        for t, d in zip(live, dc):
            dcmds[t] = d
        leaves, grads = [], []
        for t in range(3):
            if douts[t] is None:
                grads += [torch.zeros_like(Ws[t]), torch.zeros_like(bs[t])]
            else:
                dW, db = grad_buffer(Ws[t]), grad_buffer(bs[t])
                leaves.append(dict(dY=douts[t], X=cmds[t], dW=dW, db=db))
                grads += [dW, db]
        if leaves:
            ops.outer_wgrad(leaves, lowp=ctx.lowp)
        return (None,) + tuple(dcmds) + tuple(grads)


def gate_projections(cmds, linears, lowp=False):
    """cmds: the query encoder's three (B, 2H) commands; linears: mainModel's three qInput{t} nn.Linear holders; lowp: as
    query_encoder's (the bf16 model)."""
    args = []
    for lin in linears:
        args += [lin.weight, lin.bias]
    return list(_GateProjFn.apply(bool(lowp), cmds[0], cmds[1], cmds[2], *args))


class _LGPFn(torch.autograd.Function):
    """Language-guided pooling (model/LGP.py:29-51).  The 1x1 conv on the tiled query is one (B, Cq)x(Cq, C) product and
    its train-mode BN sees B*t samples that repeat t times, i.e. batch statistics over B (unbiased factor from B*t):
    both run as the small exact-fp32 GEMM + BN kernels; the pooling itself is drn_lgp_fwd/bwd on the (B, t, C) tensor."""

    @staticmethod
    def forward(ctx, dtype, bn, training, x, query, weight, gamma, beta, cbias=None):
        code = code_of(dtype)
        B, t, C, ldx = geom(x)
        Cq = weight.shape[1]
        dev = x.device
        q = query.contiguous().float()
        w2 = weight.detach().reshape(C, Cq).contiguous()
        raw = torch.empty((B, C), dtype=torch.float32, device=dev)
        stats = torch.empty((1 if B <= 128 else (B + 127) // 128, 2, C), dtype=torch.float32, device=dev)
        ops.gemm_nt([ops.gemm_desc(q, w2, raw, B, C, Cq, Lout=1, Lsrc=1, stats=stats)], ops.F32)
        ss = torch.empty((2, C), dtype=torch.float32, device=dev)
        save = torch.empty((2, C), dtype=torch.float32, device=dev)
        if training:
            ops.bn_finalize([(stats, stats.shape[0], B, ss, save)], C, gamma, beta, None, None, None, 0.0, bn.eps)
            if bn.track_running_stats and bn.running_mean is not None:
                n = B * t
                with torch.no_grad():        # (C,)-sized buffer updates: n = B*t samples for the unbiased variance
                    var = 1.0 / (save[1] * save[1]) - bn.eps
                    # (use_bn=False: the biased conv's bias cancels in the normalised value and only shifts the batch mean)
                    bn.running_mean.mul_(1 - bn.momentum).add_(save[0] if cbias is None else save[0] + cbias.detach(), alpha=bn.momentum)
                    bn.running_var.mul_(1 - bn.momentum).add_(var * (n / max(n - 1, 1)), alpha=bn.momentum)
                    bump_bn_counter(bn.num_batches_tracked, 1)
                flush_bn_counters()
        else:
            ops.bn_eval_scale_shift(C, gamma, beta, cbias.detach() if cbias is not None else None, bn.running_mean, bn.running_var, bn.eps, ss)
        qn = torch.empty((B, C), dtype=torch.float32, device=dev)
        ops.bn_apply(raw, C, ss, qn, C, B, C, 1, ops.F32, relu=False)
        out = torch.empty((B, t // 2, C), dtype=dtype, device=dev)
        att = torch.empty((B, t // 2, 2), dtype=torch.float32, device=dev)
        ops.lgp_fwd(x, ldx, qn, out, att, B, t, C, code)
        ctx.dtype, ctx.dims, ctx.training = dtype, (B, t, C, Cq, ldx), training
        ctx.has_cbias = cbias is not None
        ctx.save_for_backward(x, q, weight, gamma, raw, ss, save, qn, att)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise DrnError("backward through eval-mode LGP is not supported")
        dtype = ctx.dtype
        code = code_of(dtype)
        B, t, C, Cq, ldx = ctx.dims
        x, q, weight, gamma, raw, ss, save, qn, att = ctx.saved_tensors
        dev = x.device
        dout = _grad_nlc(dout, None, dtype)
        dx = torch.empty((B, t, C), dtype=dtype, device=dev)
        dqn = torch.empty((B, C), dtype=torch.float32, device=dev)
        ops.lgp_bwd(x, ldx, qn, att, dout, dx, dqn, B, t, C, code)
        draw = torch.empty((B, C), dtype=torch.float32, device=dev)
        dgamma, dbeta = grad_buffer(gamma), torch.empty_like(gamma)
        ops.bn_bwd(dqn, C, raw, C, ss, save, gamma, draw, C, dgamma, dbeta, False, B, C, ops.F32, relu=False)
        dW = grad_buffer(weight)
        ops.gemm_wgrad([ops.wgrad_desc(draw, q, B, Lout=1, Lsrc=1, ldy=C, ldx=Cq)], dW.view(C, Cq, 1), C, Cq, taps=1, w_layout=1,
                       dtype=ops.F32)
        wt = ops.pack_weight(weight.detach().reshape(C, Cq, 1), (1, 2, 0), ops.F32).view(Cq, C)
        dq = torch.empty((B, Cq), dtype=torch.float32, device=dev)
        ops.gemm_nt([ops.gemm_desc(draw, wt, dq, B, Cq, C, Lout=1, Lsrc=1)], ops.F32)
        # a conv bias in front of a train-mode BatchNorm cancels: its gradient is exactly zero (the reference's is 1e-8 noise)
        dcb = torch.zeros(C, dtype=torch.float32, device=dev) if ctx.has_cbias else None
        return None, None, None, dx, dq, dW, dgamma, dbeta, dcb


def lgp(x, query, conv, bn, training, dtype):
    return _LGPFn.apply(dtype, bn, training, x, query, conv.weight, bn.weight, bn.bias, conv.bias)
