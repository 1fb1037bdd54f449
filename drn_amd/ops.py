"""Thin Python wrappers over the C-ABI (include/drn_hip.h): tensors in, kernel launches out.

Nothing here computes: every function marshals device pointers / shapes into a
libdrn_hip.so call on torch's current HIP stream.  Missing library or a non-zero
return code raises (no fallback).
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import GemmDesc, WgradDesc, check, lib

F32, BF16 = 0, 1


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise _lib.DrnError("unsupported dtype %s (float32 / bfloat16 only)" % t.dtype)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DrnError("drn_amd ops run on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)


def gemm_desc(A, B, C, M, N, Cin, taps=1, stride=1, pad=0, mode=0, Lout=None, Lsrc=None, lda=None, ldb=None,
              ldc=None, bias=None, gate=None, ldg=0, stats=None, C2=None, ldc2=None, accumulate=False, out_f32=False, sumsq=None,
              gate_bwd=None):
    """gate_bwd: dict(act, ld_act, dct, ldt, dgate, dsum) -- the data gradient feeds the input stage's gate backward, which then
    happens in the launch's epilogue (DrnGemmDesc::gb_*; gemm_nt_w4c_kernel only: ask gemm_nt_plan)."""
    _need_gpu(A, B, C, bias, gate, stats, C2)
    Lout = M if Lout is None else Lout
    Lsrc = Lout if Lsrc is None else Lsrc
    gb = gate_bwd or {}
    _need_gpu(gb.get("act"), gb.get("dct"), gb.get("dgate"), gb.get("dsum"))
    return GemmDesc(A=_p(A), B=_p(B), C=_p(C), C2=_p(C2), bias=_p(bias), gate=_p(gate), stats=_p(stats),
                    M=M, N=N, Cin=Cin, taps=taps, stride=stride, pad=pad, mode=mode, Lout=Lout, Lsrc=Lsrc,
                    lda=Cin if lda is None else lda, ldb=taps * Cin if ldb is None else ldb,
                    ldc=N if ldc is None else ldc, ldg=ldg, accumulate=int(accumulate),
                    ldc2=(N if ldc is None else ldc) if ldc2 is None else ldc2, out_f32=int(out_f32), sumsq=_p(sumsq),
                    gb_act=_p(gb.get("act")), gb_dct=_p(gb.get("dct")), gb_dgate=_p(gb.get("dgate")), gb_dsum=_p(gb.get("dsum")),
                    gb_ld_act=int(gb.get("ld_act", 0)), gb_ldt=int(gb.get("ldt", 0)))


# Optional per-launch timing of the MFMA kernels (bench.py): a list collecting (tag, flops, start_event, end_event),
# the events recorded on the same stream the kernels run on.
kernel_timer = None
# Split-K (drn_gemm_nt_splitk: one launch, the last-arriving split of a tile sums the partial tiles) for problems that cannot
# fill the chip (<= 160 tiles of 128x128 on 256 CUs): conv0 forward (128 tiles x 204 K-steps) 157 -> 69 us 4-way.  With warm
# operands the 12-48-step pyramid GEMMs gain nothing from it (the exchange costs 4-8 us: scripts/bench_splitk.py), but inside
# the step, where their operands are cold, 2-4 splits of >= 12 K-steps each put twice the loads in flight: -25 us per step.
SPLITK_MIN_KSTEPS = int(os.environ.get("DRN_SPLITK_MIN_KSTEPS", "24"))
SPLITK_STEPS_PER_SPLIT = int(os.environ.get("DRN_SPLITK_STEPS", "12"))     # a bf16 split owns at least that many K-steps
# Workgroups a split launch aims for.  256 = one 8-wave workgroup per CU on the 4-slot ring (three K-tiles of loads in flight
# against cold operands) rather than 512 = two per CU on 2-slot rings: conv0's forward (128 tiles x 204 K-steps) 4 -> 2 splits, the
# step 2.222 -> 2.209 ms at T = 256 in one process (384: 2.237, 192 = conv0 unsplit: 2.27), T = 32 unchanged
# (scripts/experiments/ab_env.sh DRN_KSPLIT_WGS 512 384 256 192).
KSPLIT_WGS = int(os.environ.get("DRN_KSPLIT_WGS", "256"))


def _timed(tag, flops, launch):
    if kernel_timer is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    kernel_timer.append((tag, flops, e0, e1))


def _ksplit(descs, dtype):
    """Split-K factor for a launch (one problem, or -- bf16 -- a group of them) that cannot fill 256 CUs with 128x128 tiles."""
    tiles = sum(((d.M + 127) // 128) * ((d.N + 127) // 128) for d in descs)
    nkt = min((d.taps * d.Cin) // (64 if dtype == BF16 else 32) for d in descs)
    # exact-f32 (parity) mode keeps round 1's rule -- split only the long-K single problems -- so its summation orders, and
    # with them the ReLU decisions the tolerance tests were calibrated on, stay what they were
    if dtype != BF16 and len(descs) > 1:
        return 1
    if tiles > 160 or nkt < (SPLITK_MIN_KSTEPS if dtype == BF16 else 96):
        return 1
    return max(1, min(8, (KSPLIT_WGS if dtype == BF16 else 512) // tiles, nkt // (SPLITK_STEPS_PER_SPLIT if dtype == BF16 else 6)))


# One long-K bf16 k = 3 convolution with 16-64 tiles of 256x256 (conv0's forward: 32 tiles x 204 K-steps): full-width tiles on the
# 4-wave kernel, the K loop split so that ~256 workgroups exist, fp32 partial planes + a second launch that adds them
# (drn_gemm_nt_splitk256) -- half the L2 -> LDS bytes of the 128x128 in-launch split below.  DRN_SPLITK256=0 switches it off.
SPLITK256 = os.environ.get("DRN_SPLITK256", "0") == "1"


def _ksplit256(descs, dtype):
    if not SPLITK256 or dtype != BF16 or len(descs) != 1:
        return 1
    d = descs[0]
    if d.gate or d.C2 or d.accumulate or d.out_f32:
        return 1
    if d.taps != 3 or d.stride != 1 or d.pad != 1 or d.mode not in (0, 1) or d.Lout != d.Lsrc or d.M % 256 or d.N % 256 or d.Cin % 64 or d.M % d.Lout:
        return 1
    if d.ldc % 8 or (d.C or 0) % 16:
        return 1
    tiles = (d.M // 256) * (d.N // 256)
    nkt = (d.taps * d.Cin) // 64
    if tiles > 64 or tiles < 16 or nkt < 96:
        return 1
    # a multiple of the 3 taps: the splits of a tile that sit a whole tap apart walk the same channel blocks of the same source
    # rows at the same time (all splits of a tile run on one XCD: grid x = tile), so the input is mostly read from HBM once
    per_tap = (256 // tiles) // 3
    per = nkt // 3
    while per_tap > 1 and per % per_tap:
        per_tap -= 1
    return 3 * per_tap if per_tap >= 1 and per // per_tap >= 8 else 1


NT_KIND_TILE128, NT_KIND_TILE256, NT_KIND_W4, NT_KIND_W4C, NT_KIND_W4H = 0, 1, 2, 3, 4


def gemm_nt_plan(descs, dtype):
    """The kernel drn_gemm_nt would pick for these problems (NT_KIND_*), asked of the library itself (drn_gemm_nt_plan)."""
    arr = (GemmDesc * len(descs))(*descs)
    kind = lib().drn_gemm_nt_plan(arr, len(descs), dtype)
    if kind < 0:
        check(kind, "drn_gemm_nt_plan")
    return kind


# One long-K bf16 problem with few output tiles (conv0's forward: 8192 x 256 x 13056 = 64 tiles of 256 x 128, 204 K-steps) on the
# 4-wave kernel's half-width tiles with the K loop split INSIDE the launch so that ~256 workgroups exist (gemm_nt_w4h_kernel:
# partial accumulators exchanged through the workspace, the last-arriving split of a tile sums them in split order).
# With the K loop in tap-major order this lost to 128x128 tiles split 2 ways (91 against 84 us: every 128-byte piece of conv0's
# 71 MB input sits in a DRAM page of its own and was fetched three times, once per tap, 68 K-steps apart; with the rows of A aliased
# -- scripts/experiments/conv0_alias_bench.py -- the same launch took 56 us).  Split k = 3 launches over >= 2048 input channels now
# walk K as (channel block, tap) (W4HT_LOOP_ASM, drn_tune "w4h_tapil"): the three taps of a channel block read the same lines one
# K-step after the other.  conv0's forward 85.8 -> 70.0 us alone and 81.9 -> 75.4 us inside the step (one box each, both paths
# traced); the step itself moves inside its noise (2.007 -> 2.002 ms).  DRN_KSPLIT_W4H=0 switches it off.
KSPLIT_W4H = os.environ.get("DRN_KSPLIT_W4H", "1") == "1"


def _ksplit_w4h(descs, dtype):
    if not KSPLIT_W4H or dtype != BF16 or len(descs) != 1:
        return 1
    d = descs[0]
    if d.M % 256 or d.N % 128 or d.Cin % 64:
        return 1
    tiles, nkt = (d.M // 256) * (d.N // 128), (d.taps * d.Cin) // 64
    if tiles > 128 or nkt < 96:
        return 1
    ks = min(8, KSPLIT_WGS // tiles, nkt // 24)
    while ks > 1 and -(-nkt // -(-nkt // ks)) != ks:           # every split non-empty
        ks -= 1
    if ks < 2:
        return 1
    arr = (GemmDesc * 1)(d)
    return ks if lib().drn_gemm_nt_splitk_plan(arr, 1, ks, dtype) == NT_KIND_W4H else 1


# The in-launch split-K exchanges (partial tiles published write-through, a ticket, the last arriver sums) must CONFIRM their stores
# before the ticket whenever a kernel of another queue may run beside the launch: with one, the ticket overtook a partial about once
# in 10^5 launches of skinny_group_kernel inside the replayed two-branch graph (qdense.hip has the measurements; that kernel and the
# loss kernel always confirm with a returning read-modify-write per stored address, it costs them nothing).  The GEMM kernels'
# exchanges have never shown the window, but nobody standing here can prove a single queue (the Trainer's prefetch copies run on their
# own stream beside the replay, RCCL's kernels beside backward, a second graph branch beside the first), so every launch confirms by
# READ-BACK -- an sc1 load of every stored request, +12 us per step at T = 256, the variant that closed the window on the proven
# kernel (0 events in 120 k replays against 7) -- which is what the library does when the call's `ksplit` carries no flag.  The mode
# travels WITH THE CALL (DRN_KSPLIT_CONFIRM_* bits, include/drn_hip.h): no process-wide switch.  DRN_XCHG_CONFIRM=0 / 1 select
# "nothing" / "returning atomics" for stress tests and A/Bs (scripts/experiments), 2 = the default.
KSPLIT_CONFIRM_ATOMIC, KSPLIT_CONFIRM_NONE = 0x20000, 0x80000
XCHG_CONFIRM = os.environ.get("DRN_XCHG_CONFIRM", "2")


def _ksplit_arg(ks):
    return ks | (KSPLIT_CONFIRM_NONE if XCHG_CONFIRM == "0" else KSPLIT_CONFIRM_ATOMIC if XCHG_CONFIRM == "1" else 0)


def xchg_need(delta):
    """Kept for callers of rounds 4-5 (declare / withdraw concurrency): the exchanges now confirm unconditionally, nothing to switch."""


def gemm_nt(descs, dtype):
    arr = (GemmDesc * len(descs))(*descs)
    flops = sum(2.0 * d.M * d.N * d.taps * d.Cin for d in descs)
    ks = _ksplit_w4h(descs, dtype)
    if ks > 1:
        d0 = descs[0]
        dev = torch.device("cuda", torch.cuda.current_device())
        ws = torch.empty(ks * d0.M * d0.N, dtype=torch.float32, device=dev)
        tag = "gemm_nt[bf16] g=1 M=%d N=%d K=%d mode=%d splitK=%d(w4h)" % (d0.M, d0.N, d0.taps * d0.Cin, d0.mode, ks)
        return _timed(tag, flops, lambda: check(lib().drn_gemm_nt_splitk_grouped(arr, 1, _ksplit_arg(ks), _p(ws), _p(_counters(dev)), dtype, _stream()),
                                                "drn_gemm_nt_splitk_grouped"))
    ks = _ksplit256(descs, dtype)
    if ks > 1:
        d0 = descs[0]
        dev = torch.device("cuda", torch.cuda.current_device())
        ws = workspace(ks * d0.M * d0.N, dev)
        tag = "gemm_nt[bf16] g=1 M=%d N=%d K=%d mode=%d splitK256=%d" % (d0.M, d0.N, d0.taps * d0.Cin, d0.mode, ks)
        return _timed(tag, flops, lambda: check(lib().drn_gemm_nt_splitk256(arr, ks, _p(ws), dtype, _stream()), "drn_gemm_nt_splitk256"))
    ks = _ksplit(descs, dtype)
    if ks > 1:
        d0 = descs[0]
        dev = torch.device("cuda", torch.cuda.current_device())
        tiles = sum(((d.M + 127) // 128) * ((d.N + 127) // 128) for d in descs)
        ws = torch.empty(ks * tiles * 128 * 128, dtype=torch.float32, device=dev)
        tag = "gemm_nt[%s] g=%d M=%d N=%d K=%d mode=%d splitK=%d" % ("bf16" if dtype == BF16 else "f32", len(descs),
                                                                     sum(d.M for d in descs), d0.N, d0.taps * d0.Cin, d0.mode, ks)
        return _timed(tag, flops, lambda: check(lib().drn_gemm_nt_splitk_grouped(arr, len(descs), _ksplit_arg(ks), _p(ws), _p(_counters(dev)),
                                                                                 dtype, _stream()), "drn_gemm_nt_splitk_grouped"))
    d0 = descs[0]
    tag = "gemm_nt[%s] g=%d M=%d N=%d K=%d mode=%d" % ("bf16" if dtype == BF16 else "f32", len(descs),
                                                      sum(d.M for d in descs), d0.N, d0.taps * d0.Cin, d0.mode)
    _timed(tag, flops, lambda: check(lib().drn_gemm_nt(arr, len(descs), dtype, _stream()), "drn_gemm_nt"))


def wgrad_desc(dY, X, M, Lout=None, Lsrc=None, ldy=None, ldx=None):
    _need_gpu(dY, X)
    Lout = M if Lout is None else Lout
    return WgradDesc(dY=_p(dY), X=_p(X), M=M, Lout=Lout, Lsrc=Lout if Lsrc is None else Lsrc,
                     ldy=dY.shape[-1] if ldy is None else ldy, ldx=X.shape[-1] if ldx is None else ldx)


class WgradPending(object):
    """A caller-owned list of deferred weight-gradient reduces (DrnWgradPending, include/drn_hip.h) + the workspaces it references.
    `ranges`: [(first byte, one past the last)] of the gradient memory whose reduces may be deferred into this list -- a launch finds
    its list by where its dW lives (pending_for), so nothing process-wide says "defer": two models in one process own two lists over
    disjoint buckets and never see each other's items, and a gradient written anywhere else is reduced at once."""

    def __init__(self, ranges=()):
        from ._lib import WgradPending as _Struct
        self.c = _Struct()                 # all-zero = empty
        self.ws = []
        self.ranges = [(int(lo), int(hi)) for lo, hi in ranges]

    def __len__(self):
        return int(self.c.n)

    def reset(self):
        """Drop whatever is recorded (a backward that raised leaves items behind: the next step must not run them)."""
        self.c.n = 0
        del self.ws[:]

    def covers(self, ptr):
        return any(lo <= ptr < hi for lo, hi in self.ranges)

    def outputs(self):
        return [(int(self.c.it[i].out), int(self.c.it[i].N) * int(self.c.it[i].taps) * int(self.c.it[i].Cin)) for i in range(len(self))]


_armed = []          # WgradPending lists currently accepting items (armed by their owners between zero() and collect())


def wgrad_arm(pend):
    if not any(q is pend for q in _armed):
        _armed.append(pend)


def wgrad_disarm(pend):
    _armed[:] = [q for q in _armed if q is not pend]


def pending_for(dWs):
    """The armed list whose ranges hold EVERY one of the given gradients, else None (reduce at once)."""
    ptrs = [dW.data_ptr() for dW in dWs]
    for q in _armed:
        if all(q.covers(x) for x in ptrs):
            return q
    return None


_persistent = {}


def persistent_buffer(key, n, device, dtype=torch.float32):
    """A buffer that keeps its address for the life of the process (per key and size): small per-step outputs that a LATER launch of
    the same or of the next step reads -- the squared-sum partials the gradient-writing kernels leave for the optimizer's norm pass --
    must not move between an eager step and a captured one."""
    k = (key, int(n), str(device), dtype)
    buf = _persistent.get(k)
    if buf is None:
        # never evicted: the addresses are baked into captured hipGraphs (BatchNorm-backward tags, the one-launch LSTM's workspace,
        # squared-sum partials); the buffers are a few KB each
        buf = _persistent[k] = torch.zeros(int(n), dtype=dtype, device=device)
    return buf


last_reduce_bytes = 0


def wgrad_reduce_pending(pend, sumsq=False):
    """Run every reduce recorded in `pend` in ONE launch on the current stream (drn_wgrad_reduce_pending).  sumsq=True: the launch also
    leaves the squared sums of what it wrote; returns (ranges [(data_ptr, elements)] of the reduced gradients, partials tensor) -- or
    None when nothing was pending."""
    global last_reduce_bytes
    L = lib()
    n = len(pend)
    res = None
    ref = ctypes.byref(pend.c)
    if n > 0:
        last_reduce_bytes = int(L.drn_wgrad_pending_bytes(ref))      # (for the profile tables: the launch's HBM-roofline denominator)
    if n > 0 and sumsq:
        ranges = pend.outputs()
        blocks = int(L.drn_wgrad_pending_blocks(ref))
        dev = torch.device("cuda", torch.cuda.current_device())
        part = persistent_buffer(("wgrad_reduce_all", ranges[0][0]), blocks, dev)
        check(L.drn_wgrad_reduce_pending(ref, _p(part), _stream()), "drn_wgrad_reduce_pending")
        res = (ranges, part)
    elif n > 0:
        check(L.drn_wgrad_reduce_pending(ref, None, _stream()), "drn_wgrad_reduce_pending")
    del pend.ws[:]
    return res


def gemm_wgrad(descs, dW, N, Cin, taps=1, stride=1, pad=0, w_layout=0, accumulate=False, dtype=F32):
    """dW (fp32) = sum over all groups / rows of dY^T * im2col(X); see include/drn_hip.h."""
    _need_gpu(dW)
    assert dW.dtype == torch.float32 and dW.is_contiguous()
    m_total = sum(d.M for d in descs)
    n_ws = lib().drn_wgrad_ws_elems(m_total, N, Cin, taps)
    ws = torch.empty(max(int(n_ws), 1), dtype=torch.float32, device=dW.device)
    arr = (WgradDesc * len(descs))(*descs)
    pend = pending_for([dW])
    tag = "gemm_wgrad[%s] g=%d M=%d N=%d K=%d" % ("bf16" if dtype == BF16 else "f32", len(descs), m_total, N, taps * Cin)
    _timed(tag, 2.0 * m_total * N * taps * Cin,
           lambda: check(lib().drn_gemm_wgrad(arr, len(descs), _p(dW), N, Cin, taps, stride, pad, w_layout, int(accumulate),
                                              _p(ws), dtype, ctypes.byref(pend.c) if pend is not None else None, _stream()),
                         "drn_gemm_wgrad"))
    if pend is not None:
        pend.ws.append(ws)                          # alive until the flush


def gemm_wgrad_multi(descs, dWs, N, Cin, taps=1, stride=1, pad=0, w_layout=0, accumulate=False, dtype=F32):
    """len(descs) independent weight gradients of equal N / taps (different weights / row counts) in one launch.  Cin: one int, or
    one per problem (the FPN laterals)."""
    for dW in dWs:
        _need_gpu(dW)
        assert dW.dtype == torch.float32 and dW.is_contiguous()
    cins = None
    if not isinstance(Cin, int):
        cins = (ctypes.c_int32 * len(descs))(*[int(c) for c in Cin])
        Cin = max(int(c) for c in Cin)
    n_ws = len(descs) * int(lib().drn_wgrad_ws_elems(max(d.M for d in descs), N, Cin, taps))
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=dWs[0].device)
    arr = (WgradDesc * len(descs))(*descs)
    ptrs = (ctypes.c_void_p * len(dWs))(*[dW.data_ptr() for dW in dWs])
    m_total = sum(d.M for d in descs)
    pend = pending_for(dWs)
    tag = "gemm_wgrad_multi[%s] n=%d M=%d N=%d K=%d" % ("bf16" if dtype == BF16 else "f32", len(descs), m_total, N, taps * Cin)
    flops = sum(2.0 * d.M * N * taps * (cins[i] if cins is not None else Cin) for i, d in enumerate(descs))
    _timed(tag, flops,
           lambda: check(lib().drn_gemm_wgrad_multi(arr, len(descs), ptrs, N, Cin, cins, taps, stride, pad, w_layout, int(accumulate),
                                                    _p(ws), dtype, ctypes.byref(pend.c) if pend is not None else None, _stream()),
                         "drn_gemm_wgrad_multi"))
    if pend is not None:
        pend.ws.append(ws)


# ---------------------------------------------------------------------------------------------
# HBM-bound helpers
# ---------------------------------------------------------------------------------------------
TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}
_ws_cache = {}
_ws_retired = []          # outgrown buffers stay allocated: a hipGraph captured earlier still writes to them on replay


def workspace(n_floats, device):
    """A grow-only fp32 scratch buffer per device AND stream (stream-ordered reuse on torch's current stream; two streams
    that run side by side -- drn_amd.graph.DualStreamStep -- never share one)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < n_floats:
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(int(n_floats * 1.25) + 1024, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


def cast(x, dtype):
    """fp32 -> compute dtype (contiguous)."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=TORCH_DT[dtype], device=x.device)
    check(lib().drn_cast(_p(x), _p(out), ctypes.c_int64(x.numel()), dtype, _stream()), "drn_cast")
    return out


def cast_transpose(x, dtype):
    """fp32 (M, K) -> (compute-dtype copy (M, K), its transpose (K, M)) in one pass."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    M, K = x.shape
    out = torch.empty((M, K), dtype=TORCH_DT[dtype], device=x.device)
    outT = torch.empty((K, M), dtype=TORCH_DT[dtype], device=x.device)
    if CAST_THROTTLE > 0:
        check(lib().drn_cast_transpose_throttled(_p(x), _p(out), _p(outT), M, K, dtype, CAST_THROTTLE, _stream()), "drn_cast_transpose_throttled")
    else:
        check(lib().drn_cast_transpose(_p(x), _p(out), _p(outT), M, K, dtype, _stream()), "drn_cast_transpose")
    return out, outT


# > 0: cast_transpose runs with at most that many workgroups resident (set by schedules that run the input preparation beside
# the query encoder: drn_amd.graph.ForkedStep)
CAST_THROTTLE = 0


def transpose2d(x, dtype):
    """(M, K) row-major -> (K, M) row-major, compute dtype, one LDS-tiled pass."""
    _need_gpu(x)
    M, K = x.shape
    assert x.stride(1) == 1
    out = torch.empty((K, M), dtype=x.dtype, device=x.device)
    check(lib().drn_transpose2d(_p(x), x.stride(0), _p(out), M, M, K, dtype, _stream()), "drn_transpose2d")
    return out


def pack_weight(w, perm, dtype):
    """out = w.permute(perm).contiguous() in compute dtype; w is a 3-D fp32 parameter (any strides)."""
    _need_gpu(w)
    assert w.dtype == torch.float32 and w.dim() == 3
    shape = [w.shape[i] for i in perm]
    st = [w.stride(i) for i in perm]
    out = torch.empty(shape, dtype=TORCH_DT[dtype], device=w.device)
    check(lib().drn_pack_weight(_p(w), _p(out), shape[0], shape[1], shape[2], ctypes.c_int64(st[0]), ctypes.c_int64(st[1]),
                                ctypes.c_int64(st[2]), dtype, _stream()), "drn_pack_weight")
    return out


def _row_stride(out):
    """Elements between consecutive rows of `out` viewed as (rows, C); size-1 dims carry meaningless strides."""
    for dim in range(out.dim() - 2, -1, -1):
        if out.shape[dim] > 1:
            return out.stride(dim)
    return 0


def pack_weights_into(items, dtype):
    """items: [(w, perm, out)] -- refresh existing re-laid copies `out` of fp32 weights `w` in one launch.  `out` may be a
    column block of a wider buffer (its row stride is honoured)."""
    from ._lib import PackDesc
    arr = (PackDesc * len(items))()
    for i, (w, perm, out) in enumerate(items):
        _need_gpu(w, out)
        d = arr[i]
        d.in_, d.out = _p(w), _p(out)
        d.A, d.B, d.C = (w.shape[j] for j in perm)
        d.sa, d.sb, d.sc = (w.stride(j) for j in perm)
        d.ldo = _row_stride(out)
    check(lib().drn_pack_weights(arr, len(items), dtype, _stream()), "drn_pack_weights")


_qd_counters = {}


def _counters(device):
    """DRN_QD_COUNTERS zeroed int32 arrival counters per device and stream for the K-split dense kernels (they re-arm
    themselves)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    c = _qd_counters.get(key)
    if c is None:
        c = _qd_counters[key] = torch.zeros(_lib.QD_COUNTERS, dtype=torch.int32, device=device)
    return c


def skinny_group(probs):
    """probs: list of dict(X (M<=64, K) fp32 [row stride % 4 == 0], W (N, K) contiguous, bias=None, mask=None, relu=False,
    Y=None): Y = X W^T (+bias)(ReLU)(zero where mask <= 0) for all problems in ONE launch (per DRN_QD_MAX problems).
    Returns the outputs.  Exact-fp32 MFMA, deterministic (drn_amd/csrc/qdense.hip)."""
    outs = []
    for c0 in range(0, len(probs), _lib.QD_MAX):
        chunk = probs[c0:c0 + _lib.QD_MAX]
        arr = (_lib.SkinnyDesc * len(chunk))()
        dev = chunk[0]["X"].device
        for d, q in zip(arr, chunk):
            X, W = q["X"], q["W"]
            _need_gpu(X, W)
            M, K = X.shape
            N = W.shape[0]
            assert W.shape[1] == K and W.is_contiguous() and X.stride(1) == 1 and W.dtype == torch.float32
            assert X.dtype in (torch.float32, torch.bfloat16)       # bf16 rows: the bf16 model's low-precision backward products
            Y = q.get("Y")
            if Y is None:
                Y = torch.empty((M, N), dtype=torch.float32, device=dev)
            mask = q.get("mask")
            d.X, d.W, d.bias, d.mask, d.Y = _p(X), _p(W), _p(q.get("bias")), _p(mask), _p(Y)
            d.ldx, d.ldy, d.ldm = X.stride(0), Y.stride(0), (mask.stride(0) if mask is not None else 0)
            d.M, d.N, d.K, d.relu = M, N, K, int(bool(q.get("relu")))
            d.x_dtype = BF16 if X.dtype == torch.bfloat16 else F32
            outs.append(Y)
        n_ws = int(lib().drn_skinny_group_ws_elems(arr, len(chunk)))
        ws = workspace(n_ws, dev) if n_ws else None
        check(lib().drn_skinny_group(arr, len(chunk), _p(ws), _p(_counters(dev)), _stream()), "drn_skinny_group")
    return outs


def skinny_linear(x, W, bias=None, relu=False, mask=None):
    """y = x W^T (+ bias)(ReLU) for batch-sized x (M <= 64 rows), fp32."""
    return skinny_group([dict(X=x, W=W, bias=bias, relu=relu, mask=mask)])[0]


def skinny_rows(X, W, bias=None):
    """Y = X W^T for a taller X (e.g. clips x words rows): 64-row blocks of X as the problems of grouped launches."""
    M = X.shape[0]
    Y = torch.empty((M, W.shape[0]), dtype=torch.float32, device=X.device)
    skinny_group([dict(X=X[r:r + 64], W=W, bias=bias, Y=Y[r:r + 64]) for r in range(0, M, 64)])
    return Y


def outer_wgrad(probs, lowp=False):
    """probs: list of dict(dY (M, N), X (M, K) or None, dW (N, K) or None, db=None, db2=None): dW = dY^T X, db = db2 = column
    sums of dY, all problems in one launch per DRN_QD_MAX (drn_amd/csrc/qdense.hip).  lowp (the bf16 model): operands rounded to
    bf16 on their way into the MFMA, fp32 accumulation."""
    for c0 in range(0, len(probs), _lib.QD_MAX):
        chunk = probs[c0:c0 + _lib.QD_MAX]
        arr = (_lib.OuterDesc * len(chunk))()
        for d, q in zip(arr, chunk):
            dY, X, dW = q["dY"], q.get("X"), q.get("dW")
            _need_gpu(dY, X, dW)
            assert dY.stride(1) == 1 and dY.dtype == torch.float32
            d.dY, d.X, d.dW, d.db, d.db2 = _p(dY), _p(X), _p(dW), _p(q.get("db")), _p(q.get("db2"))
            d.ldy, d.M, d.N = dY.stride(0), dY.shape[0], dY.shape[1]
            if dW is not None:
                assert X.stride(1) == 1 and X.shape[0] == dY.shape[0] and dW.stride(1) == 1 and dW.shape == (dY.shape[1], X.shape[1])
                d.ldx, d.ldw, d.K = X.stride(0), dW.stride(0), X.shape[1]
        check(lib().drn_outer_wgrad(arr, len(chunk), BF16 if lowp else F32, _stream()), "drn_outer_wgrad")


def skinny_ok(M, N, K):
    return M <= 64 and K % 4 == 0


def touch(t):
    """Stream a tensor through the caches (drn_touch)."""
    check(lib().drn_touch(_p(t), ctypes.c_int64(t.numel() * t.element_size()), _stream()), "drn_touch")


def copy_multi(pairs):
    """pairs: [(dst, src or None)] device tensors -- one launch (drn_copy_multi).  Contiguous pairs of equal byte size are flat
    copies (src None: zero-fill); a 2-D `src` with fewer columns than the contiguous 2-D `dst` (same rows, same dtype) is copied into
    the leading columns and the rest of every row is zeroed."""
    n = len(pairs)
    assert 1 <= n <= 8
    rows2d = (ctypes.c_int32 * (3 * n))()
    for i, (d, s_) in enumerate(pairs):
        _need_gpu(d, s_)
        assert d.is_contiguous()
        if s_ is None or (s_.is_contiguous() and s_.numel() == d.numel() and s_.dtype == d.dtype):
            continue
        assert s_.dim() == 2 and d.dim() == 2 and s_.dtype == d.dtype and s_.shape[0] == d.shape[0] and s_.shape[1] <= d.shape[1] \
            and s_.stride(1) == 1, "copy_multi: a padded pair needs (rows, <= cols) -> (rows, cols) of one dtype"
        es = d.element_size()
        rows2d[3 * i], rows2d[3 * i + 1], rows2d[3 * i + 2] = s_.shape[1] * es, s_.stride(0) * es, d.shape[1] * es
    srcs = (ctypes.c_void_p * n)(*[s_.data_ptr() if s_ is not None else None for _, s_ in pairs])
    dsts = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    nbytes = (ctypes.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in pairs])
    check(lib().drn_copy_multi(srcs, dsts, nbytes, rows2d, n, _stream()), "drn_copy_multi")


def pos_feat(start_end):
    """(B, T, 2) fp64 / fp32 proposal boundaries -> (B, T, 3) fp32 [start, end, end - start] (main_model.py:51-55)."""
    _need_gpu(start_end)
    se = start_end.contiguous()
    out = torch.empty(se.shape[:-1] + (3,), dtype=torch.float32, device=se.device)
    check(lib().drn_pos_feat(_p(se), int(se.dtype == torch.float64), _p(out), se.numel() // 2, _stream()), "drn_pos_feat")
    return out


def pos_embed_fwd(feat, W, b, out2d, ld_out, M, C, dtype):
    check(lib().drn_pos_embed_fwd(_p(feat), _p(W), _p(b), _p(out2d), ld_out, M, C, dtype, _stream()), "drn_pos_embed_fwd")


def pos_embed_bwd(dout, ld, feat, M, C, dW, db, dtype, accumulate=False):
    ws = workspace(1024 * C, dW.device)
    check(lib().drn_pos_embed_bwd(_p(dout), ld, _p(feat), M, C, _p(dW), _p(db), int(accumulate), _p(ws), dtype, _stream()),
          "drn_pos_embed_bwd")


def conv_tail_bwd(dY, ld_dy, B, Lo, Cout, wd_tail, ldw, k, stride, pad, feat, L, P, dW, db, dtype, accumulate=False):
    """Position-embedding gradients through the conv that reads the embedding (drn_conv_tail_bwd)."""
    L_ = lib()
    ws = workspace(int(L_.drn_conv_tail_bwd_ws_elems(B * Lo, k, Cout)), dW.device)
    check(L_.drn_conv_tail_bwd(_p(dY), ld_dy, B, Lo, Cout, _p(wd_tail), ctypes.c_int64(ldw), k, stride, pad, _p(feat), L, P, _p(dW),
                               _p(db), int(accumulate), _p(ws), dtype, _stream()), "drn_conv_tail_bwd")


def pairsum_add(dst, ld_dst, src, ld_src, Mdst, C, dtype, accumulate=True):
    check(lib().drn_pairsum_add(_p(dst), ld_dst, _p(src), ld_src, Mdst, C, int(accumulate), dtype, _stream()), "drn_pairsum_add")


def pairsum_chain3(d0, own1, d1, own2, d2, M1, C, dtype):
    """d1 = own1 + pairs(d0), d2 = own2 + pairs(d1) in one launch (all row strides C); the bits of two pairsum_add_to launches."""
    check(lib().drn_pairsum_chain3(_p(d0), C, _p(own1), C, _p(d1), C, _p(own2), C, _p(d2), C, M1, C, dtype, _stream()), "drn_pairsum_chain3")


def pairsum_add_to(dst, ld_dst, base, ld_base, src, ld_src, Mdst, C, dtype):
    check(lib().drn_pairsum_add_to(_p(dst), ld_dst, _p(base), ld_base, _p(src), ld_src, Mdst, C, dtype, _stream()), "drn_pairsum_add_to")


def gate_fwd(z, ld_z, gate, out, ld_out, nseq, L, C, dtype):
    """out[s, t, :C] = z[s, t, :C] * gate[s]  (the level-0 query gate as its own pass, see drn_amd.graph.ForkedStep)."""
    check(lib().drn_gate_fwd(_p(z), ld_z, _p(gate), gate.stride(0), _p(out), ld_out, nseq, L, C, dtype, _stream()), "drn_gate_fwd")


def gate_bwd(dG, ld_dg, act, ld_act, gate, dC, ld_dc, add, ld_add, dgate, nseq, L, C, dtype, dsum=None):
    """dC = (add or 0) + dG * gate; dgate = sum_t dG * act; dsum (nseq, C) fp32 = sum_t dG * gate."""
    check(lib().drn_gate_bwd(_p(dG), ld_dg, _p(act), ld_act, _p(gate), gate.stride(0), _p(add), ld_add, _p(dC), ld_dc, _p(dgate),
                             dgate.stride(0), _p(dsum), nseq, L, C, dtype, _stream()), "drn_gate_bwd")


def gate_bwd_t(dG, ld_dg, act, ld_act, gate, dCT, dgate, nseq, L, C, dtype, dsum=None):
    """dCT (C, nseq*L) = (dG * gate)^T; dgate = sum_t dG * act; dsum (nseq, C) = sum_t dG * gate."""
    check(lib().drn_gate_bwd_t(_p(dG), ld_dg, _p(act), ld_act, _p(gate), gate.stride(0), _p(dCT), ctypes.c_int64(dCT.stride(0)),
                               _p(dgate), dgate.stride(0), _p(dsum), nseq, L, C, dtype, _stream()), "drn_gate_bwd_t")


def colsum(X, ld, M, C, out, dtype, accumulate=False):
    ws = workspace(64 * C, out.device)
    check(lib().drn_colsum(_p(X), ld, M, C, _p(out), int(accumulate), _p(ws), dtype, _stream()), "drn_colsum")


# ---------------------------------------------------------------------------------------------
# BatchNorm
# ---------------------------------------------------------------------------------------------
def bn_finalize(groups, C, gamma, beta, conv_bias, running_mean, running_var, momentum, eps):
    """groups: list of (stats, tiles, M, scale_shift, save)."""
    arr = (_lib.BnGroup * len(groups))(*[_lib.BnGroup(stats=_p(s), tiles=t, M=m, scale_shift=_p(ss), save=_p(sv))
                                         for (s, t, m, ss, sv) in groups])
    check(lib().drn_bn_finalize(arr, len(groups), C, _p(gamma), _p(beta), _p(conv_bias), _p(running_mean), _p(running_var),
                                ctypes.c_float(momentum), ctypes.c_float(eps), _stream()), "drn_bn_finalize")


def bn_finalize_multi(groups, C):
    """groups: list of dicts(stats, tiles, M, ss, save, gamma, beta, conv_bias, running_mean, running_var, momentum, eps);
    one launch, groups processed in order (they may share a BatchNorm module or not)."""
    arr = (_lib.BnFinDesc * len(groups))(*[
        _lib.BnFinDesc(stats=_p(g["stats"]), tiles=g["tiles"], M=g["M"], scale_shift=_p(g["ss"]), save=_p(g["save"]),
                       gamma=_p(g["gamma"]), beta=_p(g["beta"]), conv_bias=_p(g.get("conv_bias")),
                       running_mean=_p(g.get("running_mean")), running_var=_p(g.get("running_var")),
                       momentum=g["momentum"], eps=g["eps"]) for g in groups])
    check(lib().drn_bn_finalize_multi(arr, len(groups), C, _stream()), "drn_bn_finalize_multi")


def bn_apply_multi(levels, C, dtype, relu=True):
    """levels: list of dicts(raw, ld_raw, ss, out, ld_out, M, L[, up, ld_up, gate, gated, ld_gated]): one launch."""
    arr = (_lib.BnApplyDesc * len(levels))()
    for d, v in zip(arr, levels):
        gate = v.get("gate")
        d.raw, d.scale_shift, d.out = _p(v["raw"]), _p(v["ss"]), _p(v["out"])
        d.up, d.gate, d.gated = _p(v.get("up")), _p(gate), _p(v.get("gated"))
        d.ld_raw, d.ld_out, d.ld_up = v["ld_raw"], v["ld_out"], v.get("ld_up", 0)
        d.ldg, d.ld_gated = (gate.stride(0) if gate is not None else 0), v.get("ld_gated", 0)
        d.M, d.L = v["M"], v["L"]
    check(lib().drn_bn_apply_multi(arr, len(levels), C, int(relu), dtype, _stream()), "drn_bn_apply_multi")


def bn_train_apply(levels, C, dtype, relu=True):
    """Train-mode BatchNorm (+ReLU) forward of up to DRN_MAX_GROUPS levels in ONE launch (C % 64 == 0): levels = list of dicts
    with the statistics side (stats, tiles, ss, save, gamma, beta[, conv_bias, running_mean, running_var], momentum, eps) and
    the apply side (raw, ld_raw, out, ld_out, M, L[, up, ld_up, gate, gated, ld_gated]) of drn_bn_finalize_multi +
    drn_bn_apply_multi; the groups' running statistics are updated in list order."""
    arr = _bn_train_descs(levels)
    check(lib().drn_bn_train_apply(arr, len(levels), C, int(relu), dtype, _stream()), "drn_bn_train_apply")


def _bn_train_descs(levels):
    arr = (_lib.BnTrainDesc * len(levels))()
    for d, v in zip(arr, levels):
        gate = v.get("gate")
        d.stats, d.scale_shift, d.save, d.gamma, d.beta = _p(v.get("stats")), _p(v["ss"]), _p(v["save"]), _p(v["gamma"]), _p(v["beta"])
        d.conv_bias, d.running_mean, d.running_var = _p(v.get("conv_bias")), _p(v.get("running_mean")), _p(v.get("running_var"))
        d.raw, d.out, d.up, d.gate, d.gated = _p(v["raw"]), _p(v["out"]), _p(v.get("up")), _p(gate), _p(v.get("gated"))
        d.momentum, d.eps, d.tiles = v["momentum"], v["eps"], v["tiles"]
        d.ld_raw, d.ld_out, d.ld_up = v["ld_raw"], v["ld_out"], v.get("ld_up", 0)
        d.ldg, d.ld_gated = (gate.stride(0) if gate is not None else 0), v.get("ld_gated", 0)
        d.M, d.L = v["M"], v["L"]
    return arr


# conv -> BN(train) -> ReLU in ONE launch (drn_conv_bn_train): built, bit-identical to the two-launch path and stress-tested
# (tests/test_conv_bn_gpu.py) -- and OFF by default, because it measures slower inside the step (DESIGN.md section 8, round 4:
# FPN + heads forward 274 vs 257 us).  After a workgroup's K loop the hand-off is publish -> wait -> merge, three memory-side round
# trips plus 45 MB of L2-bypassing statistics reads per launch (11-13 us), against ~3 us of epilogue + a 6-15 us BatchNorm launch
# that reads the same statistics out of L2.  DRN_BN_FUSE=1 turns it on (never together with DRN_FORCE_DEVICE, the test mode that
# puts several ranks on ONE GPU: its in-kernel wait needs the whole grid resident on a device this process owns).
BN_FUSE = os.environ.get("DRN_BN_FUSE", "0") == "1" and os.environ.get("DRN_FORCE_DEVICE") is None
DRN_ERR_UNSUPPORTED = -3


_bn_tagged = {}


def _bn_tagged_ws(device, nbytes):
    """(workspace, generation word) of the one-launch conv->BN kernel per device and stream: 64-bit {value, generation} pairs,
    zero at first and afterwards written by that kernel only (a pair is valid when it carries the current launch's generation,
    so the buffer is never cleared); grows in powers of two from 16 MB (a new, zeroed buffer; the generation keeps counting)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    ent = _bn_tagged.get(key)
    if ent is None or ent[0].numel() * 8 < nbytes:
        n = 1 << 21
        while n * 8 < nbytes:
            n *= 2
        gen = ent[1] if ent is not None else torch.zeros(1, dtype=torch.int32, device=device)
        retired = ent[2] + [ent[0]] if ent is not None else []      # (a captured hipGraph may still point into the old buffer)
        ent = _bn_tagged[key] = (torch.zeros(n, dtype=torch.int64, device=device), gen, retired)
    return ent[0], ent[1]


def conv_bn_train(descs, levels, dtype, relu=True, up_group=None):
    """descs: gemm_desc per group (no bias / gate / C2; `stats` unused); levels: the bn_train_apply dicts of the same groups
    (raw = the GEMM's C; `stats` unused).  up_group[i] = j: out_i += nearest_x2(out_j) inside the launch (the FPN top-down chain).
    -> True when launched; False when this launch cannot be fused (the caller runs gemm_nt + bn_train_apply)."""
    if not BN_FUSE or _ksplit(descs, dtype) > 1:
        return False
    arr = (GemmDesc * len(descs))(*descs)
    barr = _bn_train_descs(levels)
    ug = (ctypes.c_int32 * len(descs))(*[int(u) for u in up_group]) if up_group is not None else None
    dev = torch.device("cuda", torch.cuda.current_device())
    nbytes = int(lib().drn_conv_bn_train_ws_bytes(arr, len(descs)))
    ws, gen = _bn_tagged_ws(dev, nbytes)
    d0 = descs[0]
    flops = sum(2.0 * d.M * d.N * d.taps * d.Cin for d in descs)
    tag = "gemm_nt[%s] g=%d M=%d N=%d K=%d mode=0 +bn" % ("bf16" if dtype == BF16 else "f32", len(descs), sum(d.M for d in descs), d0.N,
                                                           d0.taps * d0.Cin)
    rc = []
    _timed(tag, flops, lambda: rc.append(lib().drn_conv_bn_train(arr, barr, len(descs), int(relu), ug, _p(ws), ctypes.c_int64(ws.numel() * 8),
                                                                 _p(gen), dtype, _stream())))
    if rc[0] == DRN_ERR_UNSUPPORTED:
        if kernel_timer is not None:
            kernel_timer.pop()
        return False
    check(rc[0], "drn_conv_bn_train")
    return True


def conv_bn_train_timeouts(reset=True):
    """Workgroups of fused conv->BN launches that gave up waiting for their tile column (0 in a healthy run).  Synchronises."""
    return int(lib().drn_conv_bn_train_timeouts(int(reset)))


BN_BWD_ONE = os.environ.get("DRN_BN_BWD_ONE", "1") != "0"      # (experiment switch: 0 = reduce + apply launches, drn_bn_bwd_multi)
_bn1_maxwg = os.environ.get("DRN_BN1_MAXWG")                  # (experiment switch: workgroup budget of the one-launch kernel)


def bn_bwd_multi(levels, C, dtype, relu=True):
    """levels: list of dicts(dout, ld_dout, raw, ld_raw, ss, save, gamma, draw, ld_draw, dgamma, dbeta, accumulate, M)."""
    def fill(with_gb):
        arr = (_lib.BnBwdDesc * len(levels))()
        for d, v in zip(arr, levels):
            d.dout, d.raw, d.scale_shift, d.save, d.gamma = _p(v["dout"]), _p(v["raw"]), _p(v["ss"]), _p(v["save"]), _p(v["gamma"])
            d.draw, d.dgamma, d.dbeta = _p(v["draw"]), _p(v["dgamma"]), _p(v["dbeta"])
            d.ld_dout, d.ld_raw, d.ld_draw, d.accumulate, d.M = v["ld_dout"], v["ld_raw"], v["ld_draw"], int(v["accumulate"]), v["M"]
            gb = v.get("gb") if with_gb else None
            if gb is not None:
                d.gb_dg, d.gb_gate, d.gb_dgate = _p(gb["dg"]), _p(gb["gate"]), _p(gb["dgate"])
                d.gb_ld_dg, d.gb_ldg, d.gb_L = gb["ld_dg"], gb["ldg"], gb["L"]
        return arr

    def ungate():
        # the query-gate backward as launches of its own (drn_gate_bwd) for the levels that asked for it inside the BatchNorm launch
        for v in levels:
            gb = v.get("gb")
            if gb is not None:
                C_ = C
                d_new = torch.empty_like(gb["dg"]) if v["dout"] is None else torch.empty_like(v["dout"])
                gate_bwd(gb["dg"], gb["ld_dg"], gb["act"], gb["ld_act"], gb["gate"], d_new, C_, v["dout"], v["ld_dout"], gb["dgate"],
                         v["M"] // gb["L"], gb["L"], C_, dtype)
                v["dout"], v["ld_dout"] = d_new, C_
                v["gb"] = None

    has_gb = any(v.get("gb") is not None for v in levels)
    arr = fill(has_gb)
    global _bn1_maxwg
    if _bn1_maxwg:
        lib().drn_tune(b"bn1_maxwg", int(_bn1_maxwg))
        _bn1_maxwg = None
    if BN_BWD_ONE:
        # one launch when the grid fits the chip at once (drn_bn_bwd_one): the tagged-pair workspace is zero at birth and keeps the
        # launch generation afterwards -- one buffer per size AND stream (launches on one stream are ordered; two streams running
        # same-sized passes at once must not share tags and the generation word)
        nbytes = int(lib().drn_bn_bwd_one_ws_bytes(arr, len(levels), C, dtype))
        if nbytes == 0 and has_gb:            # not with the gate backward inside: that one as its own launch, then ask again
            ungate()
            has_gb = False
            arr = fill(False)
            nbytes = int(lib().drn_bn_bwd_one_ws_bytes(arr, len(levels), C, dtype))
        if nbytes > 0:
            tws = persistent_buffer(("bn_bwd_one", torch.cuda.current_stream().cuda_stream), nbytes // 8, levels[0]["draw"].device,
                                    torch.int64)
            rc = lib().drn_bn_bwd_one(arr, len(levels), C, int(relu), _p(tws), ctypes.c_int64(nbytes), dtype, _stream())
            if rc != DRN_ERR_UNSUPPORTED:
                check(rc, "drn_bn_bwd_one")
                return
    if has_gb:
        ungate()
        arr = fill(False)
    ws = workspace(len(levels) * 515 * C, levels[0]["draw"].device)
    check(lib().drn_bn_bwd_multi(arr, len(levels), C, int(relu), _p(ws), dtype, _stream()), "drn_bn_bwd_multi")


def check_watchdogs():
    """Raise if any in-launch exchange gave up waiting since the last call (one-launch BatchNorm backward, fused conv -> BN, the
    one-launch BiLSTM forward): such a launch lets its outputs through INVALID and only bumps a device counter -- which somebody has to
    read.  Synchronises; drn_amd.trainer calls it at the end of every epoch and evaluation, bench.py after every timed region."""
    bad = {}
    for name, fn in (("drn_bn_bwd_one", "drn_bn_bwd_one_timeouts"), ("drn_conv_bn_train", "drn_conv_bn_train_timeouts"),
                     ("drn_lstm_seq_fwd", "drn_lstm_seq_fwd_timeouts")):
        n = int(getattr(lib(), fn)(1))
        if n:
            bad[name] = n
    if bad:
        raise _lib.DrnError("in-launch exchange watchdog fired (workgroups that gave up waiting; the step's results are invalid): %s" % bad)


def bn_bwd_one_timeouts(reset=True):
    """Workgroups of one-launch BatchNorm backward passes that gave up waiting for their channel tile (0 in a healthy run).  Synchronises."""
    return int(lib().drn_bn_bwd_one_timeouts(int(reset)))


def bn_eval_scale_shift(C, gamma, beta, conv_bias, running_mean, running_var, eps, ss):
    check(lib().drn_bn_eval_scale_shift(C, _p(gamma), _p(beta), _p(conv_bias), _p(running_mean), _p(running_var),
                                        ctypes.c_float(eps), _p(ss), _stream()), "drn_bn_eval_scale_shift")


def bn_apply(raw, ld_raw, ss, out, ld_out, M, C, L, dtype, up=None, ld_up=0, gate=None, gated=None, ld_gated=0, relu=True):
    check(lib().drn_bn_apply(_p(raw), ld_raw, _p(ss), _p(out), ld_out, M, C, L, _p(up), ld_up, _p(gate),
                             gate.stride(0) if gate is not None else 0, _p(gated), ld_gated, int(relu), dtype, _stream()),
          "drn_bn_apply")


def bn_bwd(dout, ld_dout, raw, ld_raw, ss, save, gamma, draw, ld_draw, dgamma, dbeta, accumulate, M, C, dtype, relu=True):
    ws = workspace(515 * C, draw.device)
    check(lib().drn_bn_bwd(_p(dout), ld_dout, _p(raw), ld_raw, _p(ss), _p(save), _p(gamma), _p(draw), ld_draw, _p(dgamma),
                           _p(dbeta), int(accumulate), M, C, int(relu), _p(ws), dtype, _stream()), "drn_bn_bwd")


# ---------------------------------------------------------------------------------------------
# 1-2 channel heads and losses
# ---------------------------------------------------------------------------------------------
def head_groups(xs, dxs=None, scales=None):
    """xs: list of (tensor2d_or_slice, ldx, M, L)."""
    gs = []
    for i, (x, ldx, M, L) in enumerate(xs):
        gs.append(_lib.HeadGroup(X=_p(x), dX=_p(dxs[i]) if dxs is not None else None, ldx=ldx, M=M, L=L,
                                 scale=ctypes.c_void_p(scales.data_ptr() + 4 * i) if scales is not None else None))
    return (_lib.HeadGroup * len(gs))(*gs)


def _head_calls(calls):
    """calls: list of dicts(groups (HeadGroup array), W, N, C, taps, exp_mode, + bias/out/z (forward) or dout/out/z/dW/dbias/
    dscale/ws (backward)).  Up to 2 heads share one launch (drn_amd/csrc/heads.hip)."""
    arr = (_lib.HeadCall * len(calls))()
    for d, c in zip(arr, calls):
        d.groups = ctypes.cast(c["groups"], ctypes.c_void_p)
        d.ngroups, d.N, d.C, d.taps, d.exp_mode = len(c["groups"]), c["N"], c["C"], c["taps"], int(c["exp_mode"])
        d.accumulate_dx, d.accumulate_dw = int(c.get("accumulate_dx", 0)), 0
        d.dscale_stride = c["dscale"].stride(0) if c.get("dscale") is not None and c["dscale"].numel() > 1 else 1
        d.W, d.bias, d.out, d.z = _p(c["W"]), _p(c.get("bias")), _p(c.get("out")), _p(c.get("z"))
        d.dout, d.dW, d.dbias, d.dscale, d.ws = _p(c.get("dout")), _p(c.get("dW")), _p(c.get("dbias")), _p(c.get("dscale")), _p(c.get("ws"))
    arr._groups = [c["groups"] for c in calls]        # the nested host arrays live as long as the descriptor array that points to them
    return arr


def heads_fwd(calls, dtype):
    for c0 in range(0, len(calls), 2):
        chunk = calls[c0:c0 + 2]
        check(lib().drn_heads_fwd(_head_calls(chunk), len(chunk), dtype, _stream()), "drn_heads_fwd")


def heads_bwd(calls, dtype):
    """Each call gets its own slice of the per-device workspace (drn_heads_ws_elems floats)."""
    for c0 in range(0, len(calls), 2):
        chunk = calls[c0:c0 + 2]
        sizes = [int(lib().drn_heads_ws_elems(sum(int(g.M) for g in c["groups"]), c["N"], c["C"], c["taps"])) for c in chunk]
        ws = workspace(sum(sizes), chunk[0]["dW"].device)
        off = 0
        for c, n in zip(chunk, sizes):
            c["ws"] = ws[off:off + n]
            off += n
        check(lib().drn_heads_bwd(_head_calls(chunk), len(chunk), dtype, _stream()), "drn_heads_bwd")


def head_out_fwd(groups, W, bias, N, C, taps, exp_mode, out, z, dtype):
    heads_fwd([dict(groups=groups, W=W, bias=bias, N=N, C=C, taps=taps, exp_mode=exp_mode, out=out, z=z)], dtype)


def head_out_bwd(groups, W, dout, out, z, N, C, taps, exp_mode, accumulate_dx, dW, dbias, dscale, R, dtype):
    heads_bwd([dict(groups=groups, W=W, dout=dout, out=out, z=z, N=N, C=C, taps=taps, exp_mode=exp_mode,
                    accumulate_dx=accumulate_dx, dW=dW, dbias=dbias, dscale=dscale)], dtype)


def loss_levels(levels):
    """levels: list of (L, stride, lo, hi)."""
    return (_lib.LossLevel * len(levels))(*[_lib.LossLevel(L=L, stride=s, lo=lo, hi=hi) for (L, s, lo, hi) in levels])


def fcos_loss_fwd(levels, B, logits, reg, iou, gt, gamma, alpha, target_scale, iou_stage, out6, labels=None, bumps=None):
    """out6 (6 floats) = loss_cls, loss_reg, loss_iou, n_pos, n_iou_pos, sum of the three losses.  gt (B, 2) fp32 or fp64.
    bumps: [(int64 device counter, increment)] applied by the same launch (BatchNorm num_batches_tracked)."""
    assert out6.numel() >= 6 and gt.dtype in (torch.float32, torch.float64) and gt.is_contiguous()
    ws = workspace(5 * ((logits.shape[0] + 255) // 256), logits.device)
    nb = len(bumps) if bumps else 0
    arr = (_lib.CounterBump * nb)(*[_lib.CounterBump(counter=_p(t), inc=int(n)) for t, n in bumps]) if nb else None
    check(lib().drn_fcos_loss_fwd(levels, len(levels), B, _p(logits), _p(reg), _p(iou), _p(gt), int(gt.dtype == torch.float64),
                                  ctypes.c_float(gamma), ctypes.c_float(alpha), ctypes.c_float(target_scale), int(iou_stage), _p(out6),
                                  _p(labels), _p(ws), _p(_counters(logits.device)[-1:]), arr, nb, _stream()), "drn_fcos_loss_fwd")


def fcos_loss_bwd(levels, B, logits, reg, iou, gt, gamma, alpha, target_scale, iou_stage, out6, g3, dlogits, dreg, diou):
    """g3: upstream gradients (1-element fp32 tensors or None) of loss_cls, loss_reg, loss_iou."""
    check(lib().drn_fcos_loss_bwd(levels, len(levels), B, _p(logits), _p(reg), _p(iou), _p(gt), int(gt.dtype == torch.float64),
                                  ctypes.c_float(gamma), ctypes.c_float(alpha), ctypes.c_float(target_scale), int(iou_stage), _p(out6),
                                  _p(g3[0]), _p(g3[1]), _p(g3[2]), _p(dlogits), _p(dreg), _p(diou), _stream()), "drn_fcos_loss_bwd")


def focal_fwd(logits, targets, gamma, alpha):
    """Per-element sigmoid focal losses (N, C); the reference's fcos_core._C.sigmoid_focalloss_forward."""
    _need_gpu(logits, targets)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 2
    assert targets.dtype == torch.int32 and targets.is_contiguous() and targets.numel() == logits.shape[0]
    out = torch.empty_like(logits)
    check(lib().drn_focal_fwd(_p(logits), _p(targets), ctypes.c_int64(logits.shape[0]), logits.shape[1], ctypes.c_float(gamma),
                              ctypes.c_float(alpha), _p(out), _stream()), "drn_focal_fwd")
    return out


def focal_bwd(logits, targets, d_losses, gamma, alpha):
    """d_logits (N, C) = d_losses * d loss / d logit; the reference's fcos_core._C.sigmoid_focalloss_backward."""
    _need_gpu(logits, targets, d_losses)
    assert d_losses.dtype == torch.float32 and d_losses.is_contiguous() and d_losses.shape == logits.shape
    out = torch.empty_like(logits)
    check(lib().drn_focal_bwd(_p(logits), _p(targets), _p(d_losses), ctypes.c_int64(logits.shape[0]), logits.shape[1],
                              ctypes.c_float(gamma), ctypes.c_float(alpha), _p(out), _stream()), "drn_focal_bwd")
    return out


def iou_loss_fwd(pred, target, weight=None):
    """out2 = (loss, signed divisor) of the stand-alone IOULoss (drn_iou_loss_fwd; model/layers/iou_loss.py:6-24)."""
    _need_gpu(pred, target, weight)
    assert pred.dtype == torch.float32 and pred.is_contiguous() and pred.dim() == 2 and pred.shape[1] == 2 and target.shape == pred.shape
    assert weight is None or (weight.dtype == torch.float32 and weight.is_contiguous() and weight.numel() == pred.shape[0])
    out2 = torch.empty(2, dtype=torch.float32, device=pred.device)
    check(lib().drn_iou_loss_fwd(_p(pred), _p(target), _p(weight), ctypes.c_int64(pred.shape[0]), _p(out2), _stream()), "drn_iou_loss_fwd")
    return out2


def iou_loss_bwd(pred, target, weight, out2, gout, want_pred=True, want_target=False):
    dpred = torch.empty_like(pred) if want_pred else None
    dtarget = torch.empty_like(target) if want_target else None
    check(lib().drn_iou_loss_bwd(_p(pred), _p(target), _p(weight), ctypes.c_int64(pred.shape[0]), _p(out2), _p(gout), _p(dpred),
                                 _p(dtarget), _stream()), "drn_iou_loss_bwd")
    return dpred, dtarget


# ---------------------------------------------------------------------------------------------
# query-encoder LSTM recurrence
# ---------------------------------------------------------------------------------------------
def lstm_step_fwd(xproj, whf, whr, biases, hseq, cseq, gates, out, hprev_t, lens, B, L, H, s, qvec=None, hseq16=None):
    """biases = (b_ih_f, b_hh_f, b_ih_r, b_hh_r); lens int64 on the device; layouts in include/drn_hip.h.  qvec (B, 4H) or None:
    the [first ; last] sentence vector, written by the steps that produce its rows.  whf / whr: W_hh (4H, H) fp32, or bf16 copies."""
    _need_gpu(xproj, out)
    assert whf.dtype == whr.dtype and whf.is_contiguous() and whr.is_contiguous()
    check(lib().drn_lstm_step_fwd(_p(xproj), _p(whf), _p(whr), BF16 if whf.dtype == torch.bfloat16 else F32, _p(biases[0]), _p(biases[1]), _p(biases[2]), _p(biases[3]),
                                  _p(hseq), _p(cseq), _p(gates), _p(out), _p(hprev_t), _p(qvec), _p(hseq16), _p(lens), B, L, H, s, _stream()),
          "drn_lstm_step_fwd")


# 1 = the fp16-state BiLSTM forward as ONE launch (drn_lstm_seq_fwd: hidden states handed over between resident workgroups as
# fp16 words whose spare exponent bit carries the launch parity, one wave polling one word per producer; bit-identical).  Third
# measurement of the idea, third null: 54.1 us for 8 steps against 8 x 7.3 = 58.4, the step 1.981 vs 1.982 ms -- a hand-off through
# memory costs ~5 us whatever replaces the kernel boundary (every thread polling its own words: 62.5 us).  Kept, tested, off.
LSTM_SEQ = os.environ.get("DRN_LSTM_SEQ", "0") != "0"


def lstm_seq_fwd(xproj, whf, whr, biases, hseq, cseq, gates, out, hprev_t, lens, B, L, H, qvec=None):
    """The fp16-state forward recurrence as ONE launch (drn_lstm_seq_fwd; the bits of L lstm_step_fwd launches with hseq16).  Returns
    False -- nothing launched -- when the grid does not fit the chip at once."""
    _need_gpu(xproj, out)
    assert whf.dtype == torch.float32 and whf.is_contiguous() and whr.is_contiguous()
    nbytes = int(lib().drn_lstm_seq_fwd_ws_bytes(B, L, H))
    ws = persistent_buffer(("lstm_seq", B, L, H), nbytes // 8, xproj.device, torch.int64)
    rc = lib().drn_lstm_seq_fwd(_p(xproj), _p(whf), _p(whr), _p(biases[0]), _p(biases[1]), _p(biases[2]), _p(biases[3]), _p(hseq), _p(cseq),
                                _p(gates), _p(out), _p(hprev_t), _p(qvec), _p(ws), ctypes.c_int64(nbytes), _p(lens), B, L, H, _stream())
    if rc == DRN_ERR_UNSUPPORTED:
        return False
    check(rc, "drn_lstm_seq_fwd")
    return True


def lstm_seq_fwd_timeouts(reset=True):
    return int(lib().drn_lstm_seq_fwd_timeouts(int(reset)))


def lstm_bwd_first(dout, gates, cseq, dgates, dc, dh_pass, lens, B, L, H, dqvec=None, dgates16=None):
    check(lib().drn_lstm_bwd_first(_p(dout), _p(gates), _p(cseq), _p(dgates), _p(dc), _p(dh_pass), _p(dqvec), _p(dgates16), _p(lens), B, L, H,
                                   _stream()), "drn_lstm_bwd_first")


def lstm_step_bwd(dout, gates, cseq, wtf, wtr, dgates, dc, dh_pass, lens, B, L, H, s, dqvec=None, dgates16=None):
    assert wtf.dtype == wtr.dtype and wtf.is_contiguous() and wtr.is_contiguous()
    check(lib().drn_lstm_step_bwd(_p(dout), _p(gates), _p(cseq), _p(wtf), _p(wtr), BF16 if wtf.dtype == torch.bfloat16 else F32, _p(dgates), _p(dc), _p(dh_pass), _p(dqvec),
                                  _p(dgates16), _p(lens), B, L, H, s, _stream()), "drn_lstm_step_bwd")


def postprocess(levels, B, logits, reg, iou, thr, top_n, downsample):
    """Eval post-processor on the loss-layout head outputs.  Returns (det (B,R,2), scores (B,R), locs (B,R), counts (B,nl))."""
    _need_gpu(logits, reg)
    R = sum(int(l.L) for l in levels)
    dev = logits.device
    det = torch.empty((B, R, 2), dtype=torch.float32, device=dev)
    scores = torch.empty((B, R), dtype=torch.float32, device=dev)
    locs = torch.empty((B, R), dtype=torch.float32, device=dev)
    counts = torch.empty((B, len(levels)), dtype=torch.int32, device=dev)
    check(lib().drn_postprocess(levels, len(levels), B, _p(logits), _p(reg), _p(iou), ctypes.c_float(thr), int(top_n),
                                ctypes.c_float(downsample), _p(det), _p(scores), _p(locs), _p(counts), _stream()), "drn_postprocess")
    return det, scores, locs, counts


def eval_recall(det, scores, counts, gt, ious, max_topk):
    """first_hit (B, len(ious)) int32 on the device: 0-based position, among the temporal-NMS survivors of clip b at IoU threshold
    ious[q], of the first one that overlaps gt[b] by >= ious[q]; max_topk when none of the first max_topk does
    (drn_eval_recall; utils/evaluate_utils.py:131-215).  ious: a device tensor of doubles."""
    _need_gpu(det, scores, counts, gt, ious)
    B, R = scores.shape
    assert gt.dtype in (torch.float32, torch.float64) and gt.is_contiguous() and ious.dtype == torch.float64
    out = torch.empty((B, ious.numel()), dtype=torch.int32, device=det.device)
    check(lib().drn_eval_recall(_p(det), _p(scores), _p(counts), B, counts.shape[1], R, _p(gt), int(gt.dtype == torch.float64), _p(ious),
                                ious.numel(), int(max_topk), _p(out), _stream()), "drn_eval_recall")
    return out


# ---------------------------------------------------------------------------------------------
# query-encoder glue (drn_amd/csrc/qenc.hip)
# ---------------------------------------------------------------------------------------------
def qe_embed_fwd(tokens, table, out_tm, B, L, E):
    _need_gpu(tokens, table, out_tm)
    check(lib().drn_qe_embed_fwd(_p(tokens), _p(table), _p(out_tm), B, L, E, _stream()), "drn_qe_embed_fwd")


def qe_embed_bwd(tokens, demb_tm, dtable, B, L, E, V, padding_idx):
    check(lib().drn_qe_embed_bwd(_p(tokens), _p(demb_tm), _p(dtable), B, L, E, V, padding_idx, _stream()), "drn_qe_embed_bwd")


def qe_qvec_fwd(out, lens, qvec, B, L, C):
    check(lib().drn_qe_qvec_fwd(_p(out), _p(lens), _p(qvec), B, L, C, _stream()), "drn_qe_qvec_fwd")


def qe_qvec_bwd(dqvec, lens, dout, B, L, C):
    check(lib().drn_qe_qvec_bwd(_p(dqvec), _p(lens), _p(dout), B, L, C, _stream()), "drn_qe_qvec_bwd")


def qe_attn_fwd(out, qcmd, w, bias, lens, att, cmds, B, L, C):
    _need_gpu(out, qcmd, cmds)
    check(lib().drn_qe_attn_fwd(_p(out), _p(qcmd), _p(w), _p(bias), _p(lens), _p(att), _p(cmds), B, L, C, _stream()),
          "drn_qe_attn_fwd")


def qe_attn_bwd(dcmds, att, out, qcmd, w, lens, dqcmd, dout, dw_part, dbias_part, B, L, C):
    check(lib().drn_qe_attn_bwd(_p(dcmds[0]), _p(dcmds[1]), _p(dcmds[2]), _p(att), _p(out), _p(qcmd), _p(w), _p(lens), _p(dqcmd),
                                _p(dout), _p(dw_part), _p(dbias_part), B, L, C, _stream()), "drn_qe_attn_bwd")


def colsum_segs(X, ld, M, segs):
    """segs: [(dst fp32 tensor, first column, columns)]: dst[j] = sum_m X[m][col0 + j]."""
    _need_gpu(X)
    arr = (_lib.ColSeg * len(segs))(*[_lib.ColSeg(dst=_p(d), col0=c0, n=n) for d, c0, n in segs])
    check(lib().drn_colsum_segs(_p(X), ld, M, arr, len(segs), _stream()), "drn_colsum_segs")


# ---------------------------------------------------------------------------------------------
# language-guided pooling
# ---------------------------------------------------------------------------------------------
def lgp_fwd(x, ldx, qn, out, att, B, t, C, dtype):
    check(lib().drn_lgp_fwd(_p(x), ldx, _p(qn), _p(out), C, _p(att), B, t, C, dtype, _stream()), "drn_lgp_fwd")


def lgp_bwd(x, ldx, qn, att, dout, dx, dqn, B, t, C, dtype):
    ws = workspace(B * ((t // 2 + 3) // 4) * C, dqn.device)
    check(lib().drn_lgp_bwd(_p(x), ldx, _p(qn), _p(att), _p(dout), C, _p(dx), C, _p(dqn), _p(ws), B, t, C, dtype, _stream()),
          "drn_lgp_bwd")


def mfma_sustained(iters=20000, zero_operands=False):
    """MEASUREMENT (bench.py): what the chip sustains on bf16 MFMA under its power budget -- a register-only v_mfma_f32_32x32x16_bf16 loop at
    the issue floor, random bf16 operands (or zeros).  Returns dict(tflops, clock_ghz, cycles_per_mfma).  See include/drn_hip.h."""
    dev = torch.device("cuda", torch.cuda.current_device())
    ws = torch.empty(int(lib().drn_diag_mfma_ws_bytes()), dtype=torch.uint8, device=dev)
    tf, ghz, cyc = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    check(lib().drn_diag_mfma_sustained(_p(ws), int(iters), int(bool(zero_operands)), ctypes.byref(tf), ctypes.byref(ghz), ctypes.byref(cyc),
                                        _stream()), "drn_diag_mfma_sustained")
    return {"tflops": tf.value, "clock_ghz": ghz.value, "cycles_per_mfma": cyc.value}
