"""ctypes loader for libdrn_hip.so (the C-ABI in include/drn_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails,
an exception is raised.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C drn_amd/csrc`.
"""
import ctypes

import torch  # noqa: F401  (first: libdrn_hip.so must bind to the HIP runtime PyTorch-ROCm already loaded)
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DRN_LIB_PATH") or os.path.join(_HERE, "libdrn_hip.so")     # DRN_LIB_PATH: instrumented builds (scripts/experiments)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "drn_hip.h")
_lib = None

c_int, c_void_p, c_float, c_int64, c_int32 = ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_int64, ctypes.c_int32
MAX_GROUPS = 4
QD_MAX, QD_COUNTERS = 16, 2048
LOSS_MAX_BUMPS = 32


class DrnError(RuntimeError):
    pass


class GemmDesc(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("C2", c_void_p),
                ("bias", c_void_p), ("gate", c_void_p), ("stats", c_void_p),
                ("M", c_int32), ("N", c_int32),
                ("Cin", c_int32), ("taps", c_int32), ("stride", c_int32), ("pad", c_int32), ("mode", c_int32),
                ("Lout", c_int32), ("Lsrc", c_int32),
                ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32), ("ldg", c_int32),
                ("accumulate", c_int32), ("ldc2", c_int32), ("out_f32", c_int32), ("sumsq", c_void_p),
                ("gb_act", c_void_p), ("gb_dct", c_void_p), ("gb_dgate", c_void_p), ("gb_dsum", c_void_p),
                ("gb_ld_act", c_int32), ("gb_ldt", c_int32)]


class WgradDesc(ctypes.Structure):
    _fields_ = [("dY", c_void_p), ("X", c_void_p),
                ("M", c_int32), ("Lout", c_int32), ("Lsrc", c_int32),
                ("ldy", c_int32), ("ldx", c_int32)]


class PackDesc(ctypes.Structure):
    _fields_ = [("in_", c_void_p), ("out", c_void_p), ("sa", ctypes.c_int64), ("sb", ctypes.c_int64), ("sc", ctypes.c_int64),
                ("A", c_int32), ("B", c_int32), ("C", c_int32), ("ldo", ctypes.c_int64)]


class SkinnyDesc(ctypes.Structure):
    _fields_ = [("X", c_void_p), ("W", c_void_p), ("bias", c_void_p), ("mask", c_void_p), ("Y", c_void_p),
                ("ldx", c_int32), ("ldy", c_int32), ("ldm", c_int32), ("M", c_int32), ("N", c_int32), ("K", c_int32),
                ("relu", c_int32), ("x_dtype", c_int32)]


class OuterDesc(ctypes.Structure):
    _fields_ = [("dY", c_void_p), ("X", c_void_p), ("dW", c_void_p), ("db", c_void_p), ("db2", c_void_p),
                ("ldy", c_int32), ("ldx", c_int32), ("ldw", c_int32), ("M", c_int32), ("N", c_int32), ("K", c_int32)]


class ColSeg(ctypes.Structure):
    _fields_ = [("dst", c_void_p), ("col0", c_int32), ("n", c_int32)]


class BnGroup(ctypes.Structure):
    _fields_ = [("stats", c_void_p), ("tiles", c_int32), ("M", c_int32), ("scale_shift", c_void_p), ("save", c_void_p)]


class BnFinDesc(ctypes.Structure):
    _fields_ = [("stats", c_void_p), ("tiles", c_int32), ("M", c_int32), ("scale_shift", c_void_p), ("save", c_void_p),
                ("gamma", c_void_p), ("beta", c_void_p), ("conv_bias", c_void_p), ("running_mean", c_void_p),
                ("running_var", c_void_p), ("momentum", ctypes.c_float), ("eps", ctypes.c_float)]


class BnApplyDesc(ctypes.Structure):
    _fields_ = [("raw", c_void_p), ("scale_shift", c_void_p), ("out", c_void_p), ("up", c_void_p), ("gate", c_void_p),
                ("gated", c_void_p), ("ld_raw", c_int32), ("ld_out", c_int32), ("ld_up", c_int32), ("ldg", c_int32),
                ("ld_gated", c_int32), ("M", c_int32), ("L", c_int32)]


class BnTrainDesc(ctypes.Structure):
    _fields_ = [("stats", c_void_p), ("scale_shift", c_void_p), ("save", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("conv_bias", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p), ("raw", c_void_p), ("out", c_void_p),
                ("up", c_void_p), ("gate", c_void_p), ("gated", c_void_p), ("momentum", ctypes.c_float), ("eps", ctypes.c_float),
                ("tiles", c_int32), ("ld_raw", c_int32), ("ld_out", c_int32), ("ld_up", c_int32), ("ldg", c_int32),
                ("ld_gated", c_int32), ("M", c_int32), ("L", c_int32)]


class BnBwdDesc(ctypes.Structure):
    _fields_ = [("dout", c_void_p), ("raw", c_void_p), ("scale_shift", c_void_p), ("save", c_void_p), ("gamma", c_void_p),
                ("draw", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p), ("ld_dout", c_int32), ("ld_raw", c_int32),
                ("ld_draw", c_int32), ("accumulate", c_int32), ("M", c_int32),
                ("gb_dg", c_void_p), ("gb_gate", c_void_p), ("gb_dgate", c_void_p), ("gb_ld_dg", c_int32), ("gb_ldg", c_int32), ("gb_L", c_int32)]


class HeadGroup(ctypes.Structure):
    _fields_ = [("X", c_void_p), ("dX", c_void_p), ("ldx", c_int32), ("M", c_int32), ("L", c_int32), ("scale", c_void_p)]


class HeadCall(ctypes.Structure):
    _fields_ = [("groups", c_void_p), ("ngroups", c_int32), ("N", c_int32), ("C", c_int32), ("taps", c_int32), ("exp_mode", c_int32),
                ("accumulate_dx", c_int32), ("accumulate_dw", c_int32), ("dscale_stride", c_int32), ("W", c_void_p), ("bias", c_void_p), ("out", c_void_p),
                ("z", c_void_p), ("dout", c_void_p), ("dW", c_void_p), ("dbias", c_void_p), ("dscale", c_void_p), ("ws", c_void_p)]


WGRAD_PEND_MAX = 24


class WgradPendItem(ctypes.Structure):
    _fields_ = [("ws", c_void_p), ("out", c_void_p), ("nsplit", c_int32), ("N", c_int32), ("Cin", c_int32), ("taps", c_int32),
                ("w_layout", c_int32), ("accumulate", c_int32)]


class WgradPending(ctypes.Structure):
    """DrnWgradPending (include/drn_hip.h): the caller-owned list of deferred weight-gradient reduces."""
    _fields_ = [("it", WgradPendItem * WGRAD_PEND_MAX), ("blk_start", c_int32 * (WGRAD_PEND_MAX + 1)), ("n", c_int32),
                ("sumsq", c_void_p)]


class AdamTiledItem(ctypes.Structure):
    _fields_ = [("p", c_void_p), ("off", c_int64), ("m1", c_void_p), ("m2", c_void_p), ("ld1", c_int64), ("ld2", c_int64),
                ("R", c_int32), ("C", c_int32), ("k", c_int32), ("code1", c_int32), ("code2", c_int32), ("tiles_c", c_int32)]


class CounterBump(ctypes.Structure):
    _fields_ = [("counter", c_void_p), ("inc", c_int32)]


class LossLevel(ctypes.Structure):
    _fields_ = [("L", c_int32), ("stride", c_float), ("lo", c_float), ("hi", c_float)]


def lib():
    """Load (once) and return the ctypes handle; raises DrnError loudly when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DrnError("libdrn_hip.so not found at %s -- the HIP extension is required (no CPU fallback). "
                           "Run __graft_entry__.build() or `make -C drn_amd/csrc`." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.drn_last_error.restype = ctypes.c_char_p
        for fn in ("drn_wgrad_ws_elems", "drn_skinny_group_ws_elems", "drn_opt_nblocks", "drn_gemm_nt_splitk_ws_elems", "drn_gemm_nt_splitk256_ws_elems", "drn_heads_ws_elems",
                   "drn_conv_tail_bwd_ws_elems", "drn_conv_bn_train_ws_bytes", "drn_wgrad_pending_bytes", "drn_bn_bwd_one_ws_bytes", "drn_lstm_seq_fwd_ws_bytes",
                   "drn_diag_mfma_ws_bytes"):
            if hasattr(_lib, fn):
                getattr(_lib, fn).restype = c_int64
    return _lib


def check(rc, what):
    if rc != 0:
        raise DrnError("%s failed (%d): %s" % (what, rc, lib().drn_last_error().decode()))


def declared_symbols():
    """Every function declared in include/drn_hip.h (parsed from the header itself)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(drn_[a-z0-9_]+)\s*\(", txt)))
