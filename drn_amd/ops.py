"""Thin Python wrappers over the C-ABI (include/drn_hip.h): tensors in, kernel launches out.

Nothing here computes: every function marshals device pointers / shapes into a
libdrn_hip.so call on torch's current HIP stream.  Missing library or a non-zero
return code raises (no fallback).
"""
import ctypes

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
              ldc=None, bias=None, gate=None, ldg=0, stats=None, C2=None, accumulate=False):
    _need_gpu(A, B, C, bias, gate, stats, C2)
    Lout = M if Lout is None else Lout
    Lsrc = Lout if Lsrc is None else Lsrc
    return GemmDesc(A=_p(A), B=_p(B), C=_p(C), C2=_p(C2), bias=_p(bias), gate=_p(gate), stats=_p(stats),
                    M=M, N=N, Cin=Cin, taps=taps, stride=stride, pad=pad, mode=mode, Lout=Lout, Lsrc=Lsrc,
                    lda=Cin if lda is None else lda, ldb=taps * Cin if ldb is None else ldb,
                    ldc=N if ldc is None else ldc, ldg=ldg, accumulate=int(accumulate))


def gemm_nt(descs, dtype):
    arr = (GemmDesc * len(descs))(*descs)
    check(lib().drn_gemm_nt(arr, len(descs), dtype, _stream()), "drn_gemm_nt")


def wgrad_desc(dY, X, M, Lout=None, Lsrc=None, ldy=None, ldx=None):
    _need_gpu(dY, X)
    Lout = M if Lout is None else Lout
    return WgradDesc(dY=_p(dY), X=_p(X), M=M, Lout=Lout, Lsrc=Lout if Lsrc is None else Lsrc,
                     ldy=dY.shape[-1] if ldy is None else ldy, ldx=X.shape[-1] if ldx is None else ldx)


def gemm_wgrad(descs, dW, N, Cin, taps=1, stride=1, pad=0, w_layout=0, accumulate=False, dtype=F32):
    """dW (fp32) = sum over all groups / rows of dY^T * im2col(X); see include/drn_hip.h."""
    _need_gpu(dW)
    assert dW.dtype == torch.float32 and dW.is_contiguous()
    m_total = sum(d.M for d in descs)
    n_ws = lib().drn_wgrad_ws_elems(m_total, N, Cin, taps)
    ws = torch.empty(max(int(n_ws), 1), dtype=torch.float32, device=dW.device)
    arr = (WgradDesc * len(descs))(*descs)
    check(lib().drn_gemm_wgrad(arr, len(descs), _p(dW), N, Cin, taps, stride, pad, w_layout, int(accumulate),
                               _p(ws), dtype, _stream()), "drn_gemm_wgrad")
