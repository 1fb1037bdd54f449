"""GPU parity of the MFMA implicit-GEMM kernels (drn_gemm_nt / drn_gemm_wgrad) through the C-ABI,
against plain PyTorch fp64/fp32 CPU references of the same op."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "bf16": torch.bfloat16}
# fp32: exact-f32 MFMA vs fp64 reference; bf16: inputs rounded to bf16 on both sides, fp32 accumulation
TOL = {"f32": 2e-5, "bf16": 1e-2}



def tune(monkeypatch, key, value):
    """drn_tune(key, value) for the duration of one test (defaults restored afterwards)."""
    from drn_amd import _lib
    defaults = {"tn3_minrows": 4096, "tn_fused": 1, "nt_w4": 1, "nt_w4c": 1, "exp0": 0, "nt_w4h": 160, "w4h_tapil": 2048, "w4h_halo": 1}
    _lib.check(_lib.lib().drn_tune(key.encode(), int(value)), "drn_tune")
    _RESTORE.append((key, defaults[key]))


_RESTORE = []


@pytest.fixture(autouse=True)
def _restore_tuning():
    yield
    if _RESTORE:
        from drn_amd import _lib
        while _RESTORE:
            k, v = _RESTORE.pop()
            _lib.lib().drn_tune(k.encode(), v)

def dev():
    return torch.device("cuda:0")


def rnd(shape, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    return x.to(dtype)


def close(got, ref, tol, what):
    got = got.detach().double().cpu()
    ref = ref.double()
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= tol * scale, "%s: max abs err %.3e > %.3e (scale %.3g)" % (what, err, tol * scale, scale)


def merged_stats(stats, M):
    """Merge per-128-row-slab (sum, M2) partials (drn_gemm_nt's BN statistics) into total (sum, M2) -- Chan et al."""
    st = stats.double().cpu()
    tiles = st.shape[0]
    n_t = torch.tensor([min(128, M - 128 * t) for t in range(tiles)], dtype=torch.float64)[:, None]
    total = st[:, 0].sum(0)
    mean = total / M
    m2 = (st[:, 1] + n_t * (st[:, 0] / n_t - mean) ** 2).sum(0)
    return total, m2


def nlc(x_ncl):
    return x_ncl.permute(0, 2, 1).contiguous()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K,T", [(300, 200, 264, 4), (128, 128, 64, 4), (1024, 512, 4096, 4), (64, 40, 72, 4), (1024, 512, 256, 64),
                                     (768, 320, 128, 96), (4096, 4096, 128, 32)])
def test_linear_fwd_bias_gate(dt, M, N, K, T):
    """T = rows per gate row (sequence length).  T = 4: every 32-row store chunk spans several sequences (the per-element epilogue);
    T = 64 / 32: chunks lie inside one sequence (bf16: the quad-transposed fast path, gate row loaded once per chunk); T = 96: both
    kinds in one launch; the last shape runs the 256x256 tile."""
    from drn_amd import ops
    A, W = rnd((M, K), 1, DT[dt]), rnd((N, K), 2, DT[dt])
    bias = rnd((N,), 3, torch.float32)
    gate = rnd((M // T, N), 4, torch.float32)
    pre = A.double() @ W.double().t() + bias.double()
    ref = pre * gate.double().repeat_interleave(T, 0)
    Ad, Wd = A.to(dev()), W.to(dev())
    C = torch.full((M, N), float("nan"), dtype=DT[dt], device=dev())
    C2 = torch.full((M, N), float("nan"), dtype=DT[dt], device=dev())
    bias_d, gate_d = bias.to(dev()), gate.to(dev())       # descriptors hold raw pointers: keep the tensors alive
    d = ops.gemm_desc(Ad, Wd, C, M, N, K, Lout=T, bias=bias_d, gate=gate_d, ldg=N, C2=C2)
    ops.gemm_nt([d], ops.dtype_code(Ad))
    torch.cuda.synchronize()
    close(C2, pre, TOL[dt] * np.sqrt(K / 64), "pre-gate")
    close(C, ref, TOL[dt] * np.sqrt(K / 64), "gated")
    # the same product into buffers whose row pitch is no 16-byte multiple takes the element-by-element epilogue: same bits
    Cw = torch.full((M, N + 2), float("nan"), dtype=DT[dt], device=dev())
    C2w = torch.full((M, N + 2), float("nan"), dtype=DT[dt], device=dev())
    d2 = ops.gemm_desc(Ad, Wd, Cw, M, N, K, Lout=T, bias=bias_d, gate=gate_d, ldg=N, C2=C2w, ldc=N + 2, ldc2=N + 2)
    ops.gemm_nt([d2], ops.dtype_code(Ad))
    torch.cuda.synchronize()
    assert torch.equal(Cw[:, :N], C) and torch.equal(C2w[:, :N], C2)
    assert bool(torch.isnan(Cw[:, N:].float()).all())          # nothing written beyond N


def conv_case(dt, B, L, Cin, Cout, k, s, seed=0):
    x = rnd((B, Cin, L), seed + 1, DT[dt])
    w = rnd((Cout, Cin, k), seed + 2, DT[dt]) / np.sqrt(Cin * k)
    w = w.to(DT[dt])
    return x, w


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,L,Cin,Cout,k,s", [(3, 50, 72, 136, 3, 1), (2, 64, 64, 128, 3, 2), (2, 33, 40, 24, 1, 1),
                                              (4, 256, 320, 256, 3, 1), (2, 30, 64, 64, 3, 2)])
def test_conv_fwd_and_stats(dt, B, L, Cin, Cout, k, s):
    from drn_amd import ops
    x, w = conv_case(dt, B, L, Cin, Cout, k, s)
    pad = (k - 1) // 2
    ref = F.conv1d(x.double(), w.double(), stride=s, padding=pad)           # (B,Cout,Lo)
    Lo = ref.shape[-1]
    M = B * Lo
    xd = nlc(x).to(dev())                                                   # (B,L,Cin)
    wp = w.permute(0, 2, 1).contiguous().to(dev())                          # (Cout,k,Cin)
    C = torch.full((M, Cout), float("nan"), dtype=DT[dt], device=dev())
    tiles_m = (M + 127) // 128
    stats = torch.full((tiles_m, 2, Cout), float("nan"), dtype=torch.float32, device=dev())
    d = ops.gemm_desc(xd, wp, C, M, Cout, Cin, taps=k, stride=s, pad=pad, Lout=Lo, Lsrc=L, stats=stats)
    ops.gemm_nt([d], ops.dtype_code(xd))
    torch.cuda.synchronize()
    refm = ref.permute(0, 2, 1).reshape(M, Cout)
    close(C, refm, TOL[dt], "conv out")
    tot, m2 = merged_stats(stats, M)
    close(tot, refm.sum(0), TOL[dt] * 4, "col sum")
    close(m2, ((refm - refm.mean(0)) ** 2).sum(0), TOL[dt] * 4, "col M2")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,L,Cin,Cout,k,s", [(3, 50, 72, 136, 3, 1), (2, 64, 64, 128, 3, 2), (2, 33, 40, 24, 1, 1),
                                              (2, 30, 64, 64, 3, 2)])
def test_conv_dgrad(dt, B, L, Cin, Cout, k, s):
    from drn_amd import ops
    x, w = conv_case(dt, B, L, Cin, Cout, k, s)
    pad = (k - 1) // 2
    xr = x.double().requires_grad_()
    y = F.conv1d(xr, w.double(), stride=s, padding=pad)
    Lo = y.shape[-1]
    dy = rnd(tuple(y.shape), 9, DT[dt])
    y.backward(dy.double())
    ref = xr.grad.permute(0, 2, 1).reshape(B * L, Cin)
    dyd = nlc(dy).to(dev())                                                 # (B,Lo,Cout)
    wd = w.permute(1, 2, 0).contiguous().to(dev())                          # (Cin,k,Cout)
    dX = torch.full((B * L, Cin), float("nan"), dtype=DT[dt], device=dev())
    d = ops.gemm_desc(dyd, wd, dX, B * L, Cin, Cout, taps=k, stride=s, pad=pad, mode=1, Lout=L, Lsrc=Lo)
    ops.gemm_nt([d], ops.dtype_code(dyd))
    torch.cuda.synchronize()
    close(dX, ref, TOL[dt], "dgrad")
    # accumulate flag: run again into the same buffer -> 2x
    d.accumulate = 1
    ops.gemm_nt([d], ops.dtype_code(dyd))
    torch.cuda.synchronize()
    close(dX, 2 * ref, TOL[dt] * 2, "dgrad accumulate")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_grouped_shared_weights(dt):
    """Three pyramid levels through one launch with shared weights (model/fcos.py:93-102)."""
    from drn_amd import ops
    B, Cin, Cout = 2, 64, 192
    w = (rnd((Cout, Cin, 3), 5, DT[dt]) / np.sqrt(Cin * 3)).to(DT[dt])
    wp = w.permute(0, 2, 1).contiguous().to(dev())
    descs, outs, refs, stats = [], [], [], []
    for lvl, L in enumerate((64, 32, 16)):
        x = rnd((B, Cin, L), 10 + lvl, DT[dt])
        ref = F.conv1d(x.double(), w.double(), padding=1).permute(0, 2, 1).reshape(B * L, Cout)
        xd = nlc(x).to(dev())
        C = torch.full((B * L, Cout), float("nan"), dtype=DT[dt], device=dev())
        st = torch.zeros(((B * L + 127) // 128, 2, Cout), dtype=torch.float32, device=dev())
        descs.append(ops.gemm_desc(xd, wp, C, B * L, Cout, Cin, taps=3, pad=1, Lout=L, Lsrc=L, stats=st))
        outs.append((xd, C))
        refs.append(ref)
        stats.append(st)
    ops.gemm_nt(descs, ops.dtype_code(wp))
    torch.cuda.synchronize()
    for lvl in range(3):
        close(outs[lvl][1], refs[lvl], TOL[dt], "level %d" % lvl)
        tot, m2 = merged_stats(stats[lvl], refs[lvl].shape[0])
        close(tot, refs[lvl].sum(0), TOL[dt] * 4, "stats level %d" % lvl)
        close(m2, ((refs[lvl] - refs[lvl].mean(0)) ** 2).sum(0), TOL[dt] * 4, "M2 level %d" % lvl)


def test_bad_args_raise():
    from drn_amd import ops, _lib
    A = torch.zeros(16, 12, device=dev())
    with pytest.raises(_lib.DrnError):
        ops.gemm_nt([ops.gemm_desc(A, A, A, 16, 16, 12 + 1)], 0)            # Cin not multiple of 4
    with pytest.raises(_lib.DrnError):
        ops.gemm_desc(torch.zeros(4, 4), A, A, 4, 4, 4)                      # CPU tensor


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("B,L,Cin,Cout,k,s,layout", [(3, 50, 72, 136, 3, 1, 1), (2, 64, 64, 128, 3, 2, 0),
                                                     (2, 33, 40, 24, 1, 1, 1), (8, 256, 320, 256, 3, 1, 1),
                                                     (2, 30, 64, 64, 3, 2, 1), (1, 4096, 256, 128, 1, 1, 0)])
def test_conv_wgrad(dt, B, L, Cin, Cout, k, s, layout):
    from drn_amd import ops
    x, w = conv_case(dt, B, L, Cin, Cout, k, s)
    pad = (k - 1) // 2
    wr = w.double().requires_grad_()
    y = F.conv1d(x.double(), wr, stride=s, padding=pad)
    Lo = y.shape[-1]
    dy = rnd(tuple(y.shape), 9, DT[dt])
    y.backward(dy.double())
    ref = wr.grad                                                           # (Cout,Cin,k)
    if layout == 0:
        ref = ref.permute(0, 2, 1)
    xd, dyd = nlc(x).to(dev()), nlc(dy).to(dev())
    dW = torch.full(tuple(ref.shape), float("nan"), dtype=torch.float32, device=dev())
    d = ops.wgrad_desc(dyd, xd, B * Lo, Lout=Lo, Lsrc=L)
    ops.gemm_wgrad([d], dW, Cout, Cin, taps=k, stride=s, pad=pad, w_layout=layout, dtype=ops.dtype_code(xd))
    torch.cuda.synchronize()
    close(dW, ref, TOL[dt] * np.sqrt(B * Lo / 64), "wgrad")
    ops.gemm_wgrad([d], dW, Cout, Cin, taps=k, stride=s, pad=pad, w_layout=layout, accumulate=True,
                   dtype=ops.dtype_code(xd))
    torch.cuda.synchronize()
    close(dW, 2 * ref, TOL[dt] * 2 * np.sqrt(B * Lo / 64), "wgrad accumulate")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_wgrad_grouped_levels(dt):
    """Shared-weight head conv: dW sums over the three pyramid levels (model/fcos.py:93-102)."""
    from drn_amd import ops
    B, Cin, Cout = 2, 64, 192
    w = (rnd((Cout, Cin, 3), 5, DT[dt]) / np.sqrt(Cin * 3)).double().requires_grad_()
    descs, keep, loss = [], [], 0
    for lvl, L in enumerate((64, 32, 16)):
        x = rnd((B, Cin, L), 10 + lvl, DT[dt])
        dy = rnd((B, Cout, L), 20 + lvl, DT[dt])
        loss = loss + (F.conv1d(x.double(), w, padding=1) * dy.double()).sum()
        xd, dyd = nlc(x).to(dev()), nlc(dy).to(dev())
        keep.append((xd, dyd))
        descs.append(ops.wgrad_desc(dyd, xd, B * L, Lout=L, Lsrc=L))
    loss.backward()
    dW = torch.full((Cout, Cin, 3), float("nan"), dtype=torch.float32, device=dev())
    ops.gemm_wgrad(descs, dW, Cout, Cin, taps=3, pad=1, w_layout=1, dtype=ops.dtype_code(keep[0][0]))
    torch.cuda.synchronize()
    close(dW, w.grad, TOL[dt] * 2, "grouped wgrad")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("ksplit", [2, 5])
def test_conv_fwd_splitk_with_stats_bias_gate(dt, ksplit):
    """drn_gemm_nt_splitk: K loop split over workgroups, the last-arriving split of a tile sums the partial tiles in split
    order (deterministic) and runs the epilogue -- one launch."""
    import ctypes
    from drn_amd import ops, _lib
    B, L, Cin, Cout, k = 3, 40, 192, 136, 3
    x, w = conv_case(dt, B, L, Cin, Cout, k, 1)
    ref = F.conv1d(x.double(), w.double(), padding=1).permute(0, 2, 1).reshape(B * L, Cout)
    bias = rnd((Cout,), 3, torch.float32)
    gate = rnd((B, Cout), 4, torch.float32)
    M = B * L
    xd, wp = nlc(x).to(dev()), w.permute(0, 2, 1).contiguous().to(dev())
    C = torch.full((M, Cout), float("nan"), dtype=DT[dt], device=dev())
    C2 = torch.full((M, Cout), float("nan"), dtype=DT[dt], device=dev())
    stats = torch.full(((M + 127) // 128, 2, Cout), float("nan"), dtype=torch.float32, device=dev())
    bias_d, gate_d = bias.to(dev()), gate.to(dev())       # descriptors hold raw pointers: keep the tensors alive
    d = ops.gemm_desc(xd, wp, C, M, Cout, Cin, taps=k, pad=1, Lout=L, Lsrc=L, stats=stats, bias=bias_d,
                      gate=gate_d, ldg=Cout, C2=C2)
    L_ = _lib.lib()
    L_.drn_gemm_nt_splitk_ws_elems.restype = ctypes.c_int64
    ws = torch.empty(int(L_.drn_gemm_nt_splitk_ws_elems(M, Cout, ksplit)), dtype=torch.float32, device=dev())
    counters = torch.zeros(16, dtype=torch.int32, device=dev())
    arr = (_lib.GemmDesc * 1)(d)
    for _ in range(2):                    # twice: the counters must come back re-armed
        _lib.check(L_.drn_gemm_nt_splitk(arr, ksplit, ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(counters.data_ptr()),
                                         ops.dtype_code(xd), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "splitk")
    torch.cuda.synchronize()
    assert int(counters.abs().sum()) == 0
    pre = ref + bias.double()
    close(C2, pre, TOL[dt] * 2, "pre-gate")
    close(C, pre * gate.double().repeat_interleave(L, 0), TOL[dt] * 2, "gated")
    tot, m2 = merged_stats(stats, M)
    close(tot, ref.sum(0), TOL[dt] * 4, "col sum (raw conv)")
    close(m2, ((ref - ref.mean(0)) ** 2).sum(0), TOL[dt] * 4, "col M2")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("ksplit", [2, 4])
def test_grouped_splitk_equals_plain_grouped_launch(dt, ksplit):
    """drn_gemm_nt_splitk_grouped: three problems of one launch (shared-weight conv on three pyramid levels, BN statistics in the
    epilogue), each K loop split: same outputs and statistics as the unsplit grouped launch up to fp32 summation order."""
    import ctypes
    from drn_amd import ops, _lib
    B, Cin, Cout, k = 3, 192, 136, 3
    code = ops.BF16 if dt == "bf16" else ops.F32
    x0, w = conv_case(dt, B, 40, Cin, Cout, k, 1)
    wp = w.permute(0, 2, 1).contiguous().to(dev())
    keep, plain, split = [], [], []
    for li, L in enumerate((40, 24, 9)):
        x = rnd((B, Cin, L), 30 + li, DT[dt])
        xd = nlc(x).to(dev())
        keep.append(xd)
        M = B * L
        for lst in (plain, split):
            C = torch.full((M, Cout), float("nan"), dtype=DT[dt], device=dev())
            st = torch.full(((M + 127) // 128, 2, Cout), float("nan"), dtype=torch.float32, device=dev())
            lst.append((C, st, ops.gemm_desc(xd, wp, C, M, Cout, Cin, taps=k, pad=1, Lout=L, Lsrc=L, stats=st)))
    L_ = _lib.lib()
    _lib.check(L_.drn_gemm_nt((_lib.GemmDesc * 3)(*[t[2] for t in plain]), 3, code,
                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "grouped")
    tiles = sum(((t[0].shape[0] + 127) // 128) * ((Cout + 127) // 128) for t in split)
    ws = torch.full((ksplit * tiles * 128 * 128,), float("nan"), dtype=torch.float32, device=dev())
    counters = torch.zeros(2048, dtype=torch.int32, device=dev())
    for _ in range(2):
        _lib.check(L_.drn_gemm_nt_splitk_grouped((_lib.GemmDesc * 3)(*[t[2] for t in split]), 3, ksplit, ctypes.c_void_p(ws.data_ptr()),
                                                 ctypes.c_void_p(counters.data_ptr()), code,
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "grouped split-K")
    torch.cuda.synchronize()
    assert int(counters.abs().sum()) == 0
    for (Cp, sp, _), (Cs, ss, _) in zip(plain, split):
        close(Cs, Cp.double().cpu(), TOL[dt] * 2, "grouped split-K output")
        close(ss, sp.double().cpu(), TOL[dt] * 4, "grouped split-K statistics")


@pytest.mark.parametrize("M,N,K,ksplit", [(8192, 256, 6528, 4), (8192, 256, 13056, 3), (8192, 256, 13056, 2), (64, 512, 3072, 8)])
def test_splitk_exchange_under_load(M, N, K, ksplit):
    """The one-launch split-K exchange with every CU holding two workgroups (512 of them), workspace poisoned before each of
    many launches: every launch must reproduce the first bit for bit and match an fp32 product.  (A first version issued the
    partial-tile stores as separate inline-asm statements; the compiler reused a store's data registers two instructions
    later and the > 64-bit-store hazard corrupted a few tiles per launch -- only at this residency.)"""
    import ctypes
    from drn_amd import ops, _lib
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g).to(dev()).to(torch.bfloat16)
    B = torch.randn(N, K, generator=g).to(dev()).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    arr = (_lib.GemmDesc * 1)(ops.gemm_desc(A, B, C, M, N, K))
    L_ = _lib.lib()
    ws = torch.empty(int(L_.drn_gemm_nt_splitk_ws_elems(M, N, ksplit)), dtype=torch.float32, device=dev())
    counters = torch.zeros(2048, dtype=torch.int32, device=dev())
    want = A.float() @ B.float().t()
    first = None
    for it in range(25):
        ws.fill_(float("nan"))
        _lib.check(L_.drn_gemm_nt_splitk(arr, ksplit, ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(counters.data_ptr()), ops.BF16,
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "splitk")
        out = C.float()
        if first is None:
            first = out.clone()
            assert float((out - want).abs().max()) <= 1e-2 * float(want.abs().max())
        else:
            assert torch.equal(out, first), ("launch", it, float((out - first).abs().max()))
    assert int(counters.abs().sum()) == 0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(256, 256, 512), (200, 136, 72), (4096, 512, 1024)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_nt_fp32_output_is_a_weight_gradient(dt, M, N, K, accumulate):
    """out_f32: C (fp32) (+)= A (M,K) x B (N,K)^T in the compute dtype with fp32 accumulation -- the prop_fc weight gradient
    dW = dZ^T x as an NT product of the transposed operands (drn_amd/functional.py, _InputStageFn.backward)."""
    from drn_amd import ops
    g = torch.Generator().manual_seed(11)
    A = torch.randn(M, K, generator=g).to(dev()).to(dt)
    B = torch.randn(N, K, generator=g).to(dev()).to(dt)
    C0 = torch.randn(M, N, generator=g).to(dev())
    C = C0.clone()
    ops.gemm_nt([ops.gemm_desc(A, B, C, M, N, K, out_f32=True, accumulate=accumulate)], ops.dtype_code(A))
    want = A.double() @ B.double().t() + (C0.double() if accumulate else 0)
    err = float((C.double() - want).abs().max())
    assert err <= 2e-5 * float(want.abs().max()) + 1e-4, err


def test_prop_fc_weight_gradient_nt_path_matches_tn_kernel():
    """The same gradient through drn_gemm_wgrad (TN) and through transposes + drn_gemm_nt(out_f32): fp32 accumulation of
    identical bf16 products, only the summation order differs."""
    from drn_amd import ops
    g = torch.Generator().manual_seed(12)
    R, D = 2048, 512
    dZ = torch.randn(R, D, generator=g).to(dev()).to(torch.bfloat16)
    X = torch.randn(R, D, generator=g).to(dev()).to(torch.bfloat16)
    a = torch.empty(D, D, device=dev())
    b = torch.empty(D, D, device=dev())
    ops.gemm_wgrad([ops.wgrad_desc(dZ, X, R)], a.view(D, D, 1), D, D, taps=1, w_layout=0, dtype=ops.BF16)
    dZT, XT = ops.transpose2d(dZ, ops.BF16), ops.transpose2d(X, ops.BF16)      # descriptors hold raw pointers: keep them alive
    ops.gemm_nt([ops.gemm_desc(dZT, XT, b, D, D, R, out_f32=True)], ops.BF16)
    want = dZ.double().t() @ X.double()
    assert float((a.double() - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert float((b.double() - want).abs().max()) <= 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_wgrad_multi_equals_separate_launches(dt):
    """drn_gemm_wgrad_multi: independent problems (different weights and row counts, FPN levels) in one launch."""
    from drn_amd import ops
    g = torch.Generator().manual_seed(21)
    B, Cout, Cin, k = 4, 136, 72, 3
    code = ops.dtype_code(torch.empty(1, dtype=dt))
    descs, keep, want, outs = [], [], [], []
    for L in (256, 64, 20):
        dY = torch.randn(B * L, Cout, generator=g).to(dev()).to(dt)
        X = torch.randn(B * L, Cin, generator=g).to(dev()).to(dt)
        keep.append((dY, X))
        d = ops.wgrad_desc(dY, X, B * L, Lout=L, Lsrc=L)
        descs.append(d)
        ref = torch.empty(Cout, Cin, k, device=dev())
        ops.gemm_wgrad([d], ref, Cout, Cin, taps=k, pad=1, w_layout=1, dtype=code)
        want.append(ref)
        outs.append(torch.full((Cout, Cin, k), float("nan"), device=dev()))
    ops.gemm_wgrad_multi(descs, outs, Cout, Cin, taps=k, pad=1, w_layout=1, dtype=code)
    for a, b in zip(outs, want):
        assert torch.isfinite(a).all()
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-4 * float(b.abs().max()))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_wgrad_multi_with_different_input_channels(dt):
    """The FPN 1x1 laterals (model/FPN.py:36: 256 / 512 / 1024 -> 512 channels on levels of L, L/2, L/4 positions): one launch
    for problems that share N and taps but not Cin, against one launch per problem."""
    from drn_amd import ops
    g = torch.Generator().manual_seed(22)
    B, Cout = 4, 136
    code = ops.dtype_code(torch.empty(1, dtype=dt))
    descs, keep, want, outs, cins = [], [], [], [], []
    for L, Cin in ((256, 72), (128, 136), (64, 264)):
        dY = torch.randn(B * L, Cout, generator=g).to(dev()).to(dt)
        X = torch.randn(B * L, Cin, generator=g).to(dev()).to(dt)
        keep.append((dY, X))
        d = ops.wgrad_desc(dY, X, B * L, Lout=L, Lsrc=L)
        descs.append(d)
        ref = torch.empty(Cout, Cin, 1, device=dev())
        ops.gemm_wgrad([d], ref, Cout, Cin, taps=1, w_layout=1, dtype=code)
        want.append(ref)
        outs.append(torch.full((Cout, Cin, 1), float("nan"), device=dev()))
        cins.append(Cin)
    ops.gemm_wgrad_multi(descs, outs, Cout, cins, taps=1, w_layout=1, dtype=code)
    for a, b in zip(outs, want):
        assert torch.isfinite(a).all()
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-4 * float(b.abs().max()))


# ---- fused 3-tap weight gradient (conv_wgrad3_tn_kernel): k = 3, stride 1, pad 1, bf16 ------------------------------------
@pytest.mark.parametrize("B,L,Cin,Cout,layout", [(3, 50, 72, 136, 1), (2, 64, 64, 128, 0), (5, 20, 40, 24, 1), (8, 256, 320, 256, 1),
                                                 (1, 63, 128, 128, 0), (4, 65, 136, 200, 1), (7, 3, 64, 64, 1), (2, 1000, 64, 96, 0)])
def test_conv_wgrad_fused_taps(monkeypatch, B, L, Cin, Cout, layout):
    """All three taps from one staged X block (padded row space, halo rows): against the fp64 autograd gradient, plus
    accumulate; sequence lengths below / at / above the 64-row block, ragged channel counts, one sequence."""
    from drn_amd import ops
    tune(monkeypatch, "tn3_minrows", 0)
    x, w = conv_case("bf16", B, L, Cin, Cout, 3, 1)
    wr = w.double().requires_grad_()
    y = F.conv1d(x.double(), wr, stride=1, padding=1)
    dy = rnd(tuple(y.shape), 9, DT["bf16"])
    y.backward(dy.double())
    ref = wr.grad if layout == 1 else wr.grad.permute(0, 2, 1)
    xd, dyd = nlc(x).to(dev()), nlc(dy).to(dev())
    d = ops.wgrad_desc(dyd, xd, B * L, Lout=L, Lsrc=L)
    for fused in ("1", "0"):                      # the per-tap kernel on the same inputs keeps the comparison honest
        tune(monkeypatch, "tn_fused", int(fused))
        dW = torch.full(tuple(ref.shape), float("nan"), dtype=torch.float32, device=dev())
        ops.gemm_wgrad([d], dW, Cout, Cin, taps=3, stride=1, pad=1, w_layout=layout, dtype=ops.dtype_code(xd))
        torch.cuda.synchronize()
        close(dW, ref, TOL["bf16"] * np.sqrt(B * L / 64 + 1), "wgrad fused=%s" % fused)
        ops.gemm_wgrad([d], dW, Cout, Cin, taps=3, stride=1, pad=1, w_layout=layout, accumulate=True, dtype=ops.dtype_code(xd))
        torch.cuda.synchronize()
        close(dW, 2 * ref, TOL["bf16"] * 2 * np.sqrt(B * L / 64 + 1), "wgrad accumulate fused=%s" % fused)


@pytest.mark.parametrize("multi", [False, True])
def test_conv_wgrad_fused_taps_levels(monkeypatch, multi):
    """Pyramid levels through the fused kernel: summed into one gradient (shared tower weights) and as independent problems
    (the FPN level convs); must agree with the per-tap kernel to fp32 rounding."""
    from drn_amd import ops
    tune(monkeypatch, "tn3_minrows", 0)
    g = torch.Generator().manual_seed(33)
    B, Cout, Cin = 4, 136, 200
    descs, keep = [], []
    for L in (256, 64, 20):
        dY = torch.randn(B * L, Cout, generator=g).to(dev()).to(torch.bfloat16)
        X = torch.randn(B * L, Cin, generator=g).to(dev()).to(torch.bfloat16)
        keep.append((dY, X))
        descs.append(ops.wgrad_desc(dY, X, B * L, Lout=L, Lsrc=L))
    res = {}
    for fused in ("0", "1"):
        tune(monkeypatch, "tn_fused", int(fused))
        if multi:
            outs = [torch.full((Cout, Cin, 3), float("nan"), device=dev()) for _ in descs]
            ops.gemm_wgrad_multi(descs, outs, Cout, Cin, taps=3, pad=1, w_layout=1, dtype=ops.BF16)
        else:
            outs = [torch.full((Cout, Cin, 3), float("nan"), device=dev())]
            ops.gemm_wgrad(descs, outs[0], Cout, Cin, taps=3, pad=1, w_layout=1, dtype=ops.BF16)
        torch.cuda.synchronize()
        res[fused] = outs
    for a, b in zip(res["1"], res["0"]):
        assert torch.isfinite(a).all()
        assert torch.allclose(a, b, rtol=1e-5, atol=2e-6 * float(b.abs().max()) * np.sqrt(B * 256 / 64))


# ---- the 4-wave hand-scheduled kernel of the large plain bf16 products (gemm_nt_w4.hip) -----------------------------------------
W4_CASES = [
    # B, Lout, N, K, bias, gate, C2, out_f32
    (1, 256, 256, 128, False, False, False, False),      # K/64 = 2: prologue + tail only
    (1, 256, 256, 192, True, False, False, False),       # one trip of the main loop
    (2, 256, 512, 512, True, True, True, False),         # the prop_fc epilogue: bias, gate, pre-gate copy
    (4, 256, 768, 1152, True, True, False, False),
    (1, 512, 256, 640, False, False, False, True),       # fp32 destination (the weight gradient as an NT product)
    (1, 768, 512, 2048, True, False, False, True),
    (16, 80, 256, 256, True, True, True, False),         # 32-row store chunks straddle sequences: the per-element gate path
]


@pytest.mark.parametrize("case", W4_CASES)
def test_w4_kernel_is_bit_identical_to_the_general_kernel_and_close_to_torch(monkeypatch, case):
    """gemm_nt_w4_kernel (4 waves, 128 x 128 per wave, asm main loop) runs the same MFMAs in the same K order through the same
    epilogue code as conv_gemm_nt_kernel<bf16, 2, true, 2, 4, 8, 4>: equal bits, and both within bf16 tolerance of torch."""
    from drn_amd import ops
    B, Lo, N, K, bias, gate, c2, f32out = case
    M = B * Lo
    A = rnd((M, K), 11, torch.bfloat16).to(dev())
    W = (rnd((N, K), 12, torch.float32) * 0.05).to(torch.bfloat16).to(dev())
    bias_t = rnd((N,), 13, torch.float32).to(dev()) if bias else None
    gate_t = torch.rand(B, N, generator=torch.Generator().manual_seed(14)).to(dev()) if gate else None
    tune(monkeypatch, "exp0", 1)             # 256 x 256 tiles from one big tile on
    outs = []
    for w4 in (0, 1):
        tune(monkeypatch, "nt_w4", w4)
        C = torch.full((M, N), 7.0, device=dev(), dtype=torch.float32 if f32out else torch.bfloat16)
        C2 = torch.full((M, N), 5.0, device=dev(), dtype=torch.bfloat16) if c2 else None
        d = ops.gemm_desc(A, W, C, M, N, K, Lout=Lo, Lsrc=Lo, bias=bias_t, gate=gate_t, ldg=N, C2=C2, out_f32=f32out)
        ops.gemm_nt([d], ops.BF16)
        torch.cuda.synchronize()
        outs.append((C, C2))
    assert torch.equal(outs[0][0], outs[1][0])
    if c2:
        assert torch.equal(outs[0][1], outs[1][1])
    ref = A.double().cpu() @ W.double().cpu().t()
    if bias:
        ref = ref + bias_t.double().cpu()
    if c2:
        close(outs[1][1], ref, TOL["bf16"], "pre-gate copy")
    if gate:
        ref = ref * gate_t.double().cpu().repeat_interleave(Lo, 0)
    close(outs[1][0], ref, TOL["bf16"] if not f32out else 2e-3, "w4 output")


def test_w4_kernel_declines_what_it_cannot_run(monkeypatch):
    """Shapes off the 256 / 64 grid, conv taps, strided operands with odd leading dimensions: the general kernel runs, results right."""
    from drn_amd import ops
    tune(monkeypatch, "exp0", 1)
    for (M, N, K) in [(256, 256, 96), (384, 256, 128), (256, 320, 128)]:
        A = rnd((M, K), 21, torch.bfloat16).to(dev())
        W = (rnd((N, K), 22, torch.float32) * 0.05).to(torch.bfloat16).to(dev())
        C = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
        ops.gemm_nt([ops.gemm_desc(A, W, C, M, N, K)], ops.BF16)
        torch.cuda.synchronize()
        close(C, A.double().cpu() @ W.double().cpu().t(), TOL["bf16"], "declined %s" % ((M, N, K),))
    # padded leading dimensions ARE eligible (row stride != K)
    M, N, K = 512, 256, 256
    Ap = rnd((M, K + 64), 23, torch.bfloat16).to(dev())
    Wp = (rnd((N, K + 128), 24, torch.float32) * 0.05).to(torch.bfloat16).to(dev())
    outs = []
    for w4 in (0, 1):
        tune(monkeypatch, "nt_w4", w4)
        C = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
        ops.gemm_nt([ops.gemm_desc(Ap, Wp, C, M, N, K, lda=K + 64, ldb=K + 128)], ops.BF16)
        torch.cuda.synchronize()
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    close(outs[1], Ap[:, :K].double().cpu() @ Wp[:, :K].double().cpu().t(), TOL["bf16"], "padded rows")


def test_w4_kernel_stress_many_launches_stay_bit_identical(monkeypatch):
    """60 back-to-back launches on fresh data (the hand-written loop carries its own waits and hazard padding: a missing one shows
    as rare wrong tiles, not as a crash), each compared with the general kernel on the same inputs."""
    from drn_amd import ops
    tune(monkeypatch, "exp0", 1)
    M, N, K = 1024, 512, 2048
    g = torch.Generator(device="cuda").manual_seed(5)
    bad = 0
    for it in range(60):
        A = torch.randn(M, K, generator=g, device=dev()).to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g, device=dev()) * 0.05).to(torch.bfloat16)
        outs = []
        for w4 in (1, 0):
            tune(monkeypatch, "nt_w4", w4)
            C = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
            ops.gemm_nt([ops.gemm_desc(A, W, C, M, N, K)], ops.BF16)
            outs.append(C)
        torch.cuda.synchronize()
        bad += int(not torch.equal(outs[0], outs[1]))
    assert bad == 0, "%d of 60 launches differ" % bad


@pytest.mark.parametrize("f32out", [False, True])
def test_w4_kernel_accumulates_like_the_general_kernel(monkeypatch, f32out):
    """accumulate = 1 (C += A W^T) takes the epilogue's read-modify-write path in both kernels: equal bits."""
    from drn_amd import ops
    M, N, K = 512, 256, 384
    A = rnd((M, K), 31, torch.bfloat16).to(dev())
    W = (rnd((N, K), 32, torch.float32) * 0.05).to(torch.bfloat16).to(dev())
    C0 = rnd((M, N), 33, torch.float32).to(dev()).to(torch.float32 if f32out else torch.bfloat16)
    tune(monkeypatch, "exp0", 1)
    outs = []
    for w4 in (0, 1):
        tune(monkeypatch, "nt_w4", w4)
        C = C0.clone()
        ops.gemm_nt([ops.gemm_desc(A, W, C, M, N, K, accumulate=True, out_f32=f32out)], ops.BF16)
        torch.cuda.synchronize()
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    close(outs[1], C0.double().cpu() + A.double().cpu() @ W.double().cpu().t(), 2e-2, "accumulate")


# ---- the 4-wave kernel for k = 3 / stride 1 convolutions (gemm_nt_w4c_kernel) ------------------------------------------------------
W4C_CASES = [
    # B, L, N, Cin, mode, bias, stats, gate, lda padding
    (1, 256, 256, 64, 0, False, False, False, 0),        # one K-step per tap: a tap switch after every staged item
    (2, 128, 256, 128, 0, True, True, False, 0),         # two sequences per tile: zero rows inside the tile
    (4, 64, 512, 192, 0, False, True, False, 0),
    (2, 256, 256, 256, 1, False, False, False, 0),       # data gradient: the taps run the other way
    (8, 32, 256, 128, 1, True, False, True, 0),
    (2, 128, 256, 128, 0, True, False, True, 64),        # row stride != channels
]


@pytest.mark.parametrize("case", W4C_CASES)
def test_w4c_conv_kernel_is_bit_identical_to_the_general_kernel(monkeypatch, case):
    """Same MFMAs, same K order, same epilogue statements and the same statistics order as conv_gemm_nt_kernel<bf16, 2, true, 2, 4, 8, 4>;
    the zero rows at the sequence edges come from out-of-range lanes of the staging loads instead of the zero page."""
    from drn_amd import ops
    B, L, N, Cin, mode, bias, stats, gate, pad = case
    M = B * L
    A = rnd((M, Cin + pad), 41, torch.bfloat16).to(dev())
    W = (rnd((N, 3 * Cin), 42, torch.float32) * 0.05).to(torch.bfloat16).to(dev())
    bias_t = rnd((N,), 43, torch.float32).to(dev()) if bias else None
    gate_t = torch.rand(B, N, generator=torch.Generator().manual_seed(44)).to(dev()) if gate else None
    tune(monkeypatch, "exp0", 1)
    outs = []
    for flag in (0, 1):
        tune(monkeypatch, "nt_w4c", flag)
        C = torch.full((M, N), 7.0, device=dev(), dtype=torch.bfloat16)
        st = torch.full((M // 128, 2, N), float("nan"), device=dev()) if stats else None
        d = ops.gemm_desc(A, W, C, M, N, Cin, taps=3, pad=1, mode=mode, Lout=L, Lsrc=L, lda=Cin + pad, bias=bias_t, gate=gate_t, ldg=N, stats=st)
        ops.gemm_nt([d], ops.BF16)
        torch.cuda.synchronize()
        outs.append((C, st))
    assert torch.equal(outs[0][0], outs[1][0])
    if stats:         # (round 6: gemm_nt_w4c_kernel<true> sums the slab statistics in its own fixed order)
        assert stats_agree(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[1][0].float()).all()


def test_w4c_conv_forward_matches_torch(monkeypatch):
    from drn_amd import ops
    B, L, Cin, Cout = 2, 128, 128, 256
    x, w = conv_case("bf16", B, L, Cin, Cout, 3, 1)
    ref = F.conv1d(x.double(), w.double(), padding=1).permute(0, 2, 1).reshape(B * L, Cout)
    xd, wp = nlc(x).to(dev()), w.permute(0, 2, 1).contiguous().to(dev())
    tune(monkeypatch, "exp0", 1)
    C = torch.empty(B * L, Cout, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt([ops.gemm_desc(xd, wp, C, B * L, Cout, Cin, taps=3, pad=1, Lout=L, Lsrc=L)], ops.BF16)
    torch.cuda.synchronize()
    close(C, ref, TOL["bf16"] * 2, "w4c conv forward")


@pytest.mark.parametrize("shape", [(2, 128, 128, 256, 3, 0), (1, 512, 320, 256, 5, 0), (4, 64, 192, 512, 3, 0), (2, 256, 128, 256, 2, 1)])
def test_conv_splitk256_two_launches(monkeypatch, shape):
    """drn_gemm_nt_splitk256 (conv0's forward): gemm_nt_w4c_kernel on 256x256 tiles, every split writes an fp32 partial plane, the
    second launch adds the planes in split order (+ bias) and writes the output and the per-slab BatchNorm statistics."""
    import ctypes
    from drn_amd import ops, _lib
    B, L, Cin, Cout, ksplit, mode = shape
    M = B * L
    L_ = _lib.lib()
    ws = torch.empty(int(L_.drn_gemm_nt_splitk256_ws_elems(M, Cout, ksplit)), dtype=torch.float32, device=dev())
    bias = rnd((Cout,), 3, torch.float32)
    bias_d = bias.to(dev())
    if mode == 0:
        x, w = conv_case("bf16", B, L, Cin, Cout, 3, 1)
        ref = F.conv1d(x.double(), w.double(), padding=1).permute(0, 2, 1).reshape(M, Cout)
        xd, wp = nlc(x).to(dev()), w.permute(0, 2, 1).contiguous().to(dev())
    else:      # any weights will do for mode 1: the reference is the one-launch kernel on the same descriptor
        xd = rnd((M, Cin), 51, torch.bfloat16).to(dev())
        wp = (rnd((Cout, 3 * Cin), 52, torch.float32) * 0.05).to(torch.bfloat16).to(dev())
        tune(monkeypatch, "exp0", 1)
        Cr = torch.empty(M, Cout, device=dev(), dtype=torch.float32)
        tune(monkeypatch, "nt_w4c", 0)
        Cb = torch.empty(M, Cout, device=dev(), dtype=torch.bfloat16)
        ops.gemm_nt([ops.gemm_desc(xd, wp, Cb, M, Cout, Cin, taps=3, pad=1, mode=1, Lout=L, Lsrc=L)], ops.BF16)
        torch.cuda.synchronize()
        ref = Cb.double().cpu()
    outs = []
    for rep in range(2):                  # deterministic: the planes are added in split order
        C = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=dev())
        stats = torch.full((M // 128, 2, Cout), float("nan"), dtype=torch.float32, device=dev())
        d = ops.gemm_desc(xd, wp, C, M, Cout, Cin, taps=3, pad=1, mode=mode, Lout=L, Lsrc=L, stats=stats, bias=bias_d)
        _lib.check(L_.drn_gemm_nt_splitk256((_lib.GemmDesc * 1)(d), ksplit, ctypes.c_void_p(ws.data_ptr()), ops.BF16,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "splitk256")
        torch.cuda.synchronize()
        outs.append((C, stats))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    close(outs[0][0], ref + bias.double(), TOL["bf16"] * 2, "output")
    tot, m2 = merged_stats(outs[0][1], M)
    close(tot, ref.sum(0), TOL["bf16"] * 4, "col sum (raw conv)")
    close(m2, ((ref - ref.mean(0)) ** 2).sum(0), TOL["bf16"] * 4, "col M2")


def test_splitk256_refuses_what_it_cannot_do():
    import ctypes
    from drn_amd import ops, _lib
    M, N, K = 256, 256, 256
    A = rnd((M, K), 1, torch.bfloat16).to(dev())
    W = rnd((N, K), 2, torch.bfloat16).to(dev())
    C = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    ws = torch.empty(4 * M * N, dtype=torch.float32, device=dev())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gate = torch.rand(1, N, device=dev())
    d = ops.gemm_desc(A, W, C, M, N, K, gate=gate, ldg=N)
    assert _lib.lib().drn_gemm_nt_splitk256((_lib.GemmDesc * 1)(d), 4, ctypes.c_void_p(ws.data_ptr()), ops.BF16, stream) != 0
    assert b"not supported" in _lib.lib().drn_last_error()
    d = ops.gemm_desc(A, W, C, M, N, K)             # a plain product (taps = 1): not a problem of the conv kernel
    assert _lib.lib().drn_gemm_nt_splitk256((_lib.GemmDesc * 1)(d), 2, ctypes.c_void_p(ws.data_ptr()), ops.BF16, stream) != 0


def test_w4c_conv_kernel_stress_many_launches_stay_bit_identical(monkeypatch):
    """40 back-to-back launches on fresh data, forward and data gradient alternating, four sequences per tile (zero rows inside the
    tiles, a tap switch every second K-step), statistics on: each compared with the general kernel on the same inputs."""
    from drn_amd import ops
    tune(monkeypatch, "exp0", 1)
    B, L, N, Cin = 8, 64, 256, 128
    M = B * L
    g = torch.Generator(device="cuda").manual_seed(9)
    bad = 0
    for it in range(40):
        mode = it & 1
        A = torch.randn(M, Cin, generator=g, device=dev()).to(torch.bfloat16)
        W = (torch.randn(N, 3 * Cin, generator=g, device=dev()) * 0.05).to(torch.bfloat16)
        outs = []
        for flag in (1, 0):
            tune(monkeypatch, "nt_w4c", flag)
            C = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
            st = torch.empty(M // 128, 2, N, device=dev()) if mode == 0 else None
            ops.gemm_nt([ops.gemm_desc(A, W, C, M, N, Cin, taps=3, pad=1, mode=mode, Lout=L, Lsrc=L, stats=st)], ops.BF16)
            outs.append((C, st))
        torch.cuda.synchronize()
        bad += int(not torch.equal(outs[0][0], outs[1][0]) or (mode == 0 and not stats_agree(outs[0][1], outs[1][1])))
    assert bad == 0, "%d of 40 launches differ" % bad


# ---- the 4-wave loop on 256 x 128 tiles (gemm_nt_w4h_kernel): launches that would run 128 x 128 tiles ------------------------------
W4H_CASES = [
    # levels [(B, L)], N, Cin (one int or one per level), taps, mode, bias, stats, gate, lda padding
    ([(1, 256)], 128, 128, 1, 0, False, False, False, 0),            # two K-steps: the loop runs zero trips, only its tail
    ([(2, 128)], 256, 192, 1, 0, True, True, False, 0),              # plain product, statistics, three K-steps
    ([(4, 64), (2, 128)], 256, [256, 128], 1, 1, False, False, False, 0),   # grouped 1x1 data gradients, K differs per problem
    ([(1, 256)], 128, 64, 3, 0, False, True, False, 0),              # k = 3: one K-step per tap, a tap switch after every staged item
    ([(2, 128), (4, 64)], 256, 128, 3, 0, True, True, False, 0),     # two pyramid levels, sequences end inside the tiles
    ([(2, 256), (2, 128), (8, 32)], 512, 128, 3, 1, False, False, False, 0),    # data gradient: the taps run the other way
    ([(8, 32)], 128, 128, 3, 1, True, False, True, 0),
    ([(2, 128)], 256, 128, 3, 0, True, False, True, 64),             # row stride != channels
]


def stats_agree(sh, sg):
    """Per-slab BatchNorm statistics (sum, M2) of two kernels that add the same fp32 accumulators in different (fixed) orders: equal to
    fp32 rounding of a 128-term sum, not bit for bit."""
    assert torch.isfinite(sh).all() and torch.isfinite(sg).all()
    scale = float(sg[:, 0].abs().max()) + 1.0
    return bool((sh[:, 0] - sg[:, 0]).abs().max() <= 2e-5 * scale) and \
        bool(((sh[:, 1] - sg[:, 1]).abs() <= 2e-5 * sg[:, 1].abs() + 1e-5 * scale).all())


def stats_agree_loose(sa, sb):
    """... of two kernels whose bf16-rounded OUTPUTS differ in the last place here and there (another K order): the sums agree to what
    that rounding moves"""
    assert torch.isfinite(sa).all() and torch.isfinite(sb).all()
    scale = float(sb[:, 0].abs().max()) + 1.0
    return bool((sa[:, 0] - sb[:, 0]).abs().max() <= 1e-3 * scale) and bool(((sa[:, 1] - sb[:, 1]).abs() <= 1e-2 * sb[:, 1].abs() + 1e-3 * scale).all())


def _w4h_launch(ops, case, flag, monkeypatch, seed=0):
    levels, N, Cin, taps, mode, bias, stats, gate, pad = case
    tune(monkeypatch, "nt_w4h", flag)
    descs, keep, outs = [], [], []
    for li, (B, L) in enumerate(levels):
        cin = Cin[li] if isinstance(Cin, list) else Cin
        M = B * L
        A = rnd((M, cin + pad), 61 + li + seed, torch.bfloat16).to(dev())
        W = (rnd((N, taps * cin), 71 + li + seed, torch.float32) * 0.05).to(torch.bfloat16).to(dev())
        bias_t = rnd((N,), 81 + li, torch.float32).to(dev()) if bias else None
        gate_t = torch.rand(B, N, generator=torch.Generator().manual_seed(44 + li)).to(dev()) if gate else None
        C = torch.full((M, N), 7.0, device=dev(), dtype=torch.bfloat16)
        st = torch.full((M // 128, 2, N), float("nan"), device=dev()) if stats else None
        descs.append(ops.gemm_desc(A, W, C, M, N, cin, taps=taps, pad=1 if taps == 3 else 0, mode=mode, Lout=L, Lsrc=L, lda=cin + pad,
                                   bias=bias_t, gate=gate_t, ldg=N, stats=st))
        keep.append((A, W, bias_t, gate_t))
        outs.append((C, st))
    kind = ops.gemm_nt_plan(descs, ops.BF16)
    ops.gemm_nt(descs, ops.BF16)
    torch.cuda.synchronize()
    return kind, outs, keep


@pytest.mark.parametrize("case", W4H_CASES)
def test_w4h_kernel_is_bit_identical_to_the_128_tile_kernel(monkeypatch, case):
    """Same MFMAs (operands exchanged: the tiles come out transposed), same K order per output element, same bias / gate / rounding:
    the OUTPUT the 4-wave 256 x 128 kernel writes equals what conv_gemm_nt_kernel<bf16, 2, true, 2, 4, 4, 2> writes, bit for bit.  The
    per-slab BatchNorm statistics are summed in the kernel's own fixed order since round 6 (per lane over the 8 row tiles, then over a
    DPP row): equal to fp32 rounding, and bit-identical from run to run."""
    from drn_amd import ops
    tune(monkeypatch, "w4h_halo", 0)                                    # the tap-major walk: the general kernel's K order
    kind_h, outs_h, _ = _w4h_launch(ops, case, 1, monkeypatch)          # from ONE 256 x 128 tile on
    kind_g, outs_g, keep = _w4h_launch(ops, case, 0, monkeypatch)
    assert kind_h == ops.NT_KIND_W4H and kind_g == ops.NT_KIND_TILE128
    for (Ch, sh), (Cg, sg) in zip(outs_h, outs_g):
        assert torch.equal(Ch, Cg)
        assert torch.isfinite(Ch.float()).all()
        if sh is not None:
            assert stats_agree(sh, sg)
    outs_h2 = _w4h_launch(ops, case, 1, monkeypatch)[1]
    for (Ch, sh), (C2, s2) in zip(outs_h, outs_h2):                  # deterministic: the same bits again
        assert torch.equal(Ch, C2) and (sh is None or torch.equal(sh, s2))
    # the shipped walk of the k = 3 launches whose sequences are multiples of 64 rows (round 6: a channel block staged ONCE for its three
    # taps, K order (channel block, tap)): another fixed summation order -- equal to fp32 re-association, bit-identical from run to run
    tune(monkeypatch, "w4h_halo", 1)
    outs_x = _w4h_launch(ops, case, 1, monkeypatch)[1]
    outs_x2 = _w4h_launch(ops, case, 1, monkeypatch)[1]
    for (Cx, sx), (Cx2, sx2), (Ch, sh) in zip(outs_x, outs_x2, outs_h):
        assert torch.equal(Cx, Cx2) and (sx is None or torch.equal(sx, sx2))
        close(Cx, Ch.double().cpu(), 1e-2, "staged-once walk vs tap-major")
        if sx is not None:
            assert stats_agree_loose(sx, sh)
    # ... and close to the product itself (first level, plain cases without a gate)
    levels, N, Cin, taps, mode, bias, stats, gate, pad = case
    if taps == 1 and not gate:
        A, W, bias_t, _ = keep[0]
        cin = Cin[0] if isinstance(Cin, list) else Cin
        ref = A[:, :cin].double().cpu() @ W.double().cpu().t() + (bias_t.double().cpu() if bias else 0.0)
        close(outs_h[0][0], ref, TOL["bf16"] * 2, "w4h plain product")


def test_w4h_conv_forward_matches_torch(monkeypatch):
    from drn_amd import ops
    B, L, Cin, Cout = 2, 128, 128, 256
    x, w = conv_case("bf16", B, L, Cin, Cout, 3, 1)
    ref = F.conv1d(x.double(), w.double(), padding=1).permute(0, 2, 1).reshape(B * L, Cout)
    xd, wp = nlc(x).to(dev()), w.permute(0, 2, 1).contiguous().to(dev())
    tune(monkeypatch, "nt_w4h", 1)
    C = torch.empty(B * L, Cout, device=dev(), dtype=torch.bfloat16)
    d = ops.gemm_desc(xd, wp, C, B * L, Cout, Cin, taps=3, pad=1, Lout=L, Lsrc=L)
    assert ops.gemm_nt_plan([d], ops.BF16) == ops.NT_KIND_W4H
    ops.gemm_nt([d], ops.BF16)
    torch.cuda.synchronize()
    close(C, ref, TOL["bf16"] * 2, "w4h conv forward")


def test_w4h_kernel_takes_the_pyramid_launches_and_declines_the_rest(monkeypatch):
    """With the shipped threshold (>= 160 tiles of 256 x 128) the grouped FPN / head launches of the benchmarked shape go to the
    4-wave kernel; strided convolutions, fp32 destinations, short pyramids (Charades-STA's T = 32) and fp32 models do not."""
    from drn_amd import ops
    A = torch.zeros(8192, 1536, device=dev(), dtype=torch.bfloat16)
    W = torch.zeros(1024, 3 * 1024, device=dev(), dtype=torch.bfloat16)
    C = torch.zeros(8192, 1024, device=dev(), dtype=torch.bfloat16)

    def levels(B, T, N, Cin, **kw):
        return [ops.gemm_desc(A, W, C, B * (T >> l), N, Cin, Lout=T >> l, Lsrc=T >> l, lda=Cin, **kw) for l in range(3)]
    assert ops.gemm_nt_plan(levels(32, 256, 512, 512, taps=3, pad=1), ops.BF16) == ops.NT_KIND_W4H          # FPN output convs
    assert ops.gemm_nt_plan(levels(32, 256, 512, 512, taps=3, pad=1, mode=1), ops.BF16) == ops.NT_KIND_W4H  # ... their data gradient
    assert ops.gemm_nt_plan(levels(32, 256, 512, 1024), ops.BF16) == ops.NT_KIND_W4H                          # mix_fc (1x1)
    assert ops.gemm_nt_plan(levels(32, 256, 1024, 512, taps=3, pad=1), ops.BF16) == ops.NT_KIND_W4C          # the towers keep 256 x 256 tiles
    assert ops.gemm_nt_plan(levels(32, 32, 512, 512, taps=3, pad=1), ops.BF16) == ops.NT_KIND_TILE128        # T = 32: 28 tiles
    assert ops.gemm_nt_plan(levels(32, 256, 512, 512, taps=3, pad=1), ops.F32) == ops.NT_KIND_TILE128
    d = ops.gemm_desc(A, W, C, 4096, 512, 256, taps=3, stride=2, pad=1, Lout=128, Lsrc=256, lda=256)             # conv1: stride 2
    assert ops.gemm_nt_plan([d], ops.BF16) == ops.NT_KIND_TILE128
    Cf = torch.zeros(8192, 512, device=dev(), dtype=torch.float32)
    d = ops.gemm_desc(A, W, Cf, 8192, 512, 1024, out_f32=True)
    assert ops.gemm_nt_plan([d] * 3, ops.BF16) == ops.NT_KIND_TILE128


def test_w4h_kernel_stress_many_launches_stay_bit_identical(monkeypatch):
    """40 back-to-back grouped launches on fresh data, forward (statistics on) and data gradient alternating, three levels with
    sequences that end inside the tiles: each compared with the 128 x 128 kernel on the same inputs."""
    from drn_amd import ops
    N, Cin = 256, 128
    levels = [(4, 128), (4, 64), (8, 32)]
    g = torch.Generator(device="cuda").manual_seed(19)
    bad = 0
    for it in range(40):
        mode = it & 1
        data = [(torch.randn(B * L, Cin, generator=g, device=dev()).to(torch.bfloat16),
                 (torch.randn(N, 3 * Cin, generator=g, device=dev()) * 0.05).to(torch.bfloat16)) for B, L in levels]
        outs = []
        for flag in (1, 0):
            tune(monkeypatch, "nt_w4h", flag)
            descs, res = [], []
            for (B, L), (A, W) in zip(levels, data):
                C = torch.empty(B * L, N, device=dev(), dtype=torch.bfloat16)
                st = torch.empty(B * L // 128, 2, N, device=dev()) if mode == 0 else None
                descs.append(ops.gemm_desc(A, W, C, B * L, N, Cin, taps=3, pad=1, mode=mode, Lout=L, Lsrc=L, stats=st))
                res.append((C, st))
            ops.gemm_nt(descs, ops.BF16)
            outs.append(res)
        torch.cuda.synchronize()
        for (Ch, sh), (Cg, sg) in zip(*outs):
            bad += int(not torch.equal(Ch, Cg) or (mode == 0 and not stats_agree(sh, sg)))
    assert bad == 0, "%d level outputs of 40 launches differ" % bad


@pytest.mark.parametrize("shape", [(2, 256, 2048, 128, 3, 0, True), (1, 256, 6144, 128, 1, 0, False), (2, 128, 2048, 256, 3, 1, False),
                                   (4, 128, 2112, 128, 3, 0, True)])
def test_w4h_kernel_splitk_in_launch(monkeypatch, shape):
    """conv0's forward shape class: few 256 x 128 tiles, a long K loop split inside the launch (ops._ksplit_w4h ->
    drn_gemm_nt_splitk_grouped -> gemm_nt_w4h_kernel): partial accumulators exchanged through the workspace, the last-arriving split sums
    them in split order.  Deterministic (two runs equal bits), equal to the unsplit kernel within fp32 re-association, statistics included,
    counters left re-armed."""
    from drn_amd import ops
    B, L, Cin, N, taps, mode, stats = shape
    M = B * L
    A = rnd((M, Cin), 91, torch.bfloat16).to(dev())
    W = (rnd((N, taps * Cin), 92, torch.float32) * 0.02).to(torch.bfloat16).to(dev())
    bias = rnd((N,), 93, torch.float32).to(dev())
    tune(monkeypatch, "nt_w4h", 1)

    def run(split):
        monkeypatch.setattr(ops, "KSPLIT_W4H", split)
        C = torch.full((M, N), 7.0, device=dev(), dtype=torch.bfloat16)
        st = torch.full((M // 128, 2, N), float("nan"), device=dev()) if stats else None
        d = ops.gemm_desc(A, W, C, M, N, Cin, taps=taps, pad=1 if taps == 3 else 0, mode=mode, Lout=L, Lsrc=L, bias=bias, stats=st)
        ks = ops._ksplit_w4h([d], ops.BF16)
        ops.gemm_nt([d], ops.BF16)
        torch.cuda.synchronize()
        return ks, C, st
    tune(monkeypatch, "w4h_halo", 0)
    ks0, C0, s0 = run(False)
    outs = []
    # three walks of K: a channel block staged once for its three taps (round 6, the shipped one where every sequence is a multiple of 64
    # rows); split k = 3 launches over >= 2048 channels as (channel block, tap) with an A item per K-step; tap-major as unsplit
    for halo, tapil in ((1, 2048), (0, 2048), (0, 0)):
        tune(monkeypatch, "w4h_halo", halo)
        tune(monkeypatch, "w4h_tapil", tapil)
        ks1, C1, s1 = run(True)
        ks2, C2, s2 = run(True)
        assert ks1 >= 2 and ks0 == 1
        assert torch.equal(C1, C2) and (not stats or torch.equal(s1, s2))
        close(C1, C0.double().cpu(), 1e-2, "split vs unsplit output")
        if stats:
            t1, q1 = merged_stats(s1, M)
            t0, q0 = merged_stats(s0, M)
            close(t1, t0, 1e-4, "column sums")
            close(q1, q0, 1e-4, "column M2")
        assert int(ops._counters(dev()).abs().sum()) == 0
        outs.append(C1)
    if taps == 3:                    # another summation order: the walks agree to rounding -- and not bit for bit: equal bits
        close(outs[0], outs[2].double().cpu(), 1e-2, "staged-once vs tap-major")            # would mean the flag never reached the kernel
        close(outs[1], outs[2].double().cpu(), 1e-2, "interleaved vs tap-major")
        if Cin >= 2048:
            assert not torch.equal(outs[1], outs[2]), "the interleaved-tap walk was not taken"
        if L % 64 == 0 and Cin % 64 == 0:
            assert not torch.equal(outs[0], outs[2]), "the staged-once walk was not taken"


@pytest.mark.parametrize("kind", ["general", "w4h"])
def test_split_exchange_confirmation_changes_no_bits(monkeypatch, kind):
    """drn_tune "xchg_confirm" (ops.xchg_need: set while another queue's kernels may run beside the launch): every partial-tile store
    of the in-launch split-K exchange is followed by a returning OR-with-zero on its address (1) or an sc1 load of it (2) before the
    ticket.  Values untouched:
    the launch with and without it gives the same bits, counters re-armed, for the general 128 x 128 kernel and for gemm_nt_w4h_kernel
    (tap-interleaved and plain)."""
    import ctypes
    from drn_amd import ops, _lib
    L_ = _lib.lib()
    outs = []
    for confirm in (1, 0, 2):                    # returning atomics / nothing / sc1 read-back
        monkeypatch.setattr(ops, "XCHG_CONFIRM", str(confirm))       # (-> the DRN_KSPLIT_CONFIRM_* bit of the call's `ksplit`)
        if kind == "general":
            M, N, K, ks = 512, 384, 4096, 4
            g = torch.Generator().manual_seed(5)
            A = torch.randn(M, K, generator=g).to(dev()).to(torch.bfloat16)
            B = torch.randn(N, K, generator=g).to(dev()).to(torch.bfloat16)
            C = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
            arr = (_lib.GemmDesc * 1)(ops.gemm_desc(A, B, C, M, N, K))
            L_.drn_gemm_nt_splitk_ws_elems.restype = ctypes.c_int64
            ws = torch.full((int(L_.drn_gemm_nt_splitk_ws_elems(M, N, ks)),), float("nan"), dtype=torch.float32, device=dev())
            counters = torch.zeros(2048, dtype=torch.int32, device=dev())
            for _ in range(3):
                _lib.check(L_.drn_gemm_nt_splitk(arr, ops._ksplit_arg(ks), ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(counters.data_ptr()), ops.BF16,
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "splitk")
            torch.cuda.synchronize()
            assert int(counters.abs().sum()) == 0
            close(C, (A.float() @ B.float().t()).double().cpu(), 2e-2, "product")
            outs.append([C.clone()])
        else:
            tune(monkeypatch, "nt_w4h", 1)
            monkeypatch.setattr(ops, "KSPLIT_W4H", True)
            res = []
            for tapil in (2048, 0):
                tune(monkeypatch, "w4h_tapil", tapil)
                M, L, Cin, N = 512, 256, 2048, 128
                A = rnd((M, Cin), 91, torch.bfloat16).to(dev())
                W = (rnd((N, 3 * Cin), 92, torch.float32) * 0.02).to(torch.bfloat16).to(dev())
                C = torch.full((M, N), 7.0, device=dev(), dtype=torch.bfloat16)
                d = ops.gemm_desc(A, W, C, M, N, Cin, taps=3, pad=1, Lout=L, Lsrc=L)
                assert ops._ksplit_w4h([d], ops.BF16) >= 2
                ops.gemm_nt([d], ops.BF16)
                torch.cuda.synchronize()
                assert int(ops._counters(dev()).abs().sum()) == 0
                res.append(C)
            outs.append(res)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)


def test_deferred_wgrad_reduce_passes_in_one_launch_equal_the_immediate_ones():
    """drn_wgrad_defer / drn_wgrad_reduce_pending: three weight gradients (fused-tap k = 3, per-tap stride 2, a grouped multi launch)
    whose reduce passes run as ONE launch at the end give the bits of the launches that reduce themselves."""
    from drn_amd import ops, _lib
    g = torch.Generator().manual_seed(3)

    def case(B, L, Cin, Cout, k, stride):
        x = torch.randn(B, L, Cin, generator=g).to(torch.bfloat16).to(dev())
        Lo = (L + 2 * ((k - 1) // 2) - k) // stride + 1
        dy = torch.randn(B, Lo, Cout, generator=g).to(torch.bfloat16).to(dev())
        return x, dy, (B, L, Lo, Cin, Cout, k, stride)
    cases = [case(32, 256, 256, 256, 3, 1), case(16, 128, 128, 256, 3, 2), case(32, 128, 512, 128, 1, 1)]
    levels = [case(32, 256, 128, 128, 3, 1), case(32, 128, 128, 128, 3, 1), case(32, 64, 128, 128, 3, 1)]

    def run():
        outs = []
        for x, dy, (B, L, Lo, Cin, Cout, k, stride) in cases:
            dW = torch.full((Cout, Cin, k), float("nan"), device=dev())
            ops.gemm_wgrad([ops.wgrad_desc(dy, x, B * Lo, Lout=Lo, Lsrc=L, ldy=Cout, ldx=Cin)], dW, Cout, Cin, taps=k, stride=stride,
                           pad=(k - 1) // 2, w_layout=1, dtype=ops.BF16)
            outs.append(dW)
        dWs = [torch.full((128, 128, 3), float("nan"), device=dev()) for _ in levels]
        ops.gemm_wgrad_multi([ops.wgrad_desc(dy, x, m[0] * m[2], Lout=m[2], Lsrc=m[1], ldy=128, ldx=128) for x, dy, m in levels], dWs, 128, 128,
                             taps=3, stride=1, pad=1, w_layout=1, dtype=ops.BF16)
        return outs + dWs
    ref = run()
    torch.cuda.synchronize()
    # the list is the CALLER's (DrnWgradPending): armed over the whole address space here, a launch finds it by where its dW lives
    pend = ops.WgradPending([(0, 1 << 62)])
    ops.wgrad_arm(pend)
    try:
        got = run()
        n_pending = len(pend)
        assert len(pend.ws) >= 4                              # the workspaces stay alive until the flush
        ops.wgrad_reduce_pending(pend)
    finally:
        ops.wgrad_disarm(pend)
    torch.cuda.synchronize()
    assert n_pending >= 3 and len(pend) == 0 and pend.ws == []
    for a, b in zip(ref, got):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_deferred_wgrad_lists_are_independent_and_refuse_what_they_cannot_order():
    """Two lists in one process (two models): each records only the launches whose gradients live in its ranges, flushing one leaves
    the other's items alone, both give the immediate path's bits.  An accumulating reduce is never recorded (it reads `out`); a second
    reduce onto an output that already has one recorded is an error before anything is launched."""
    import ctypes
    from drn_amd import ops, _lib
    g = torch.Generator().manual_seed(4)
    B, L, Cin, Cout = 32, 256, 256, 256
    x = torch.randn(B, L, Cin, generator=g).to(torch.bfloat16).to(dev())
    dys = [torch.randn(B, L, Cout, generator=g).to(torch.bfloat16).to(dev()) for _ in range(2)]

    def wgrad(dy, dW, accumulate=False):
        ops.gemm_wgrad([ops.wgrad_desc(dy, x, B * L, Lout=L, Lsrc=L, ldy=Cout, ldx=Cin)], dW, Cout, Cin, taps=3, stride=1, pad=1,
                       w_layout=1, accumulate=accumulate, dtype=ops.BF16)
    ref = [torch.empty(Cout, Cin, 3, device=dev()) for _ in range(2)]
    for dy, dW in zip(dys, ref):
        wgrad(dy, dW)
    acc_ref = ref[0].clone()
    wgrad(dys[1], acc_ref, accumulate=True)
    torch.cuda.synchronize()
    outs = [torch.full((Cout, Cin, 3), float("nan"), device=dev()) for _ in range(2)]
    span = lambda t: (t.data_ptr(), t.data_ptr() + t.numel() * 4)
    pa, pb = ops.WgradPending([span(outs[0])]), ops.WgradPending([span(outs[1])])
    ops.wgrad_arm(pa)
    ops.wgrad_arm(pb)
    try:
        wgrad(dys[0], outs[0])
        wgrad(dys[1], outs[1])
        assert len(pa) == 1 and len(pb) == 1 and pa.outputs()[0][0] == outs[0].data_ptr() and pb.outputs()[0][0] == outs[1].data_ptr()
        with pytest.raises(_lib.DrnError, match="already has a deferred reduce pending"):
            wgrad(dys[1], outs[0])
        assert len(pa) == 1
        ops.wgrad_reduce_pending(pa)
        assert len(pa) == 0 and len(pb) == 1
        torch.cuda.synchronize()
        assert torch.equal(outs[0], ref[0]) and not torch.isfinite(outs[1]).all()      # b's reduce has not run
        wgrad(dys[1], outs[0], accumulate=True)              # accumulating: reduced at once, never recorded
        assert len(pa) == 0
        ops.wgrad_reduce_pending(pb)
    finally:
        ops.wgrad_disarm(pa)
        ops.wgrad_disarm(pb)
    torch.cuda.synchronize()
    assert torch.equal(outs[1], ref[1]) and torch.equal(outs[0], acc_ref)


@pytest.mark.parametrize("B,L,Cout,D", [(2, 256, 256, 512), (4, 128, 128, 256), (8, 64, 64, 256), (16, 32, 64, 512), (32, 256, 256, 4096)])
def test_w4c_data_gradient_with_the_gate_backward_in_its_epilogue(monkeypatch, B, L, Cout, D):
    """DrnGemmDesc::gb_*: conv0's data gradient (k = 3, gemm_nt_w4c_kernel) followed by the input stage's gate backward, in one launch,
    against the two launches it replaces (drn_gemm_nt + drn_gate_bwd_t): the transposed gated gradient bit for bit (the epilogue rounds
    the product to bf16 before gating, exactly what the second launch read back), the per-clip sums to fp32 re-association.  Clips of
    256 / 128 / 64 / 32 rows (one clip over two waves ... eight clips per tile) and the benchmarked shape."""
    from drn_amd import ops
    M = B * L
    P = 64
    Cin = D + P
    dY = rnd((M, Cout), 41, torch.bfloat16).to(dev())
    Wd = (rnd((D, 3 * Cout), 42, torch.float32) * 0.05).to(torch.bfloat16).to(dev())          # (Cin - P, k, Cout) as the GEMM's B operand
    Z = rnd((M, D), 43, torch.bfloat16).to(dev())
    gate = (rnd((B, D), 44, torch.float32) * 0.5 + 1.0).to(dev())
    tune(monkeypatch, "exp0", 1)                       # 256 x 256 tiles whatever the tile count (the toy shapes)
    dx = torch.zeros((M, Cin), device=dev(), dtype=torch.bfloat16)
    d0 = ops.gemm_desc(dY, Wd, dx, M, D, Cout, taps=3, pad=1, mode=1, Lout=L, Lsrc=L, ldc=Cin)
    assert ops.gemm_nt_plan([d0], ops.BF16) == ops.NT_KIND_W4C
    ops.gemm_nt([d0], ops.BF16)
    dZT0 = torch.empty((D, M), device=dev(), dtype=torch.bfloat16)
    dgate0 = torch.empty((B, D), device=dev())
    dsum0 = torch.empty((B, D), device=dev())
    ops.gate_bwd_t(dx, Cin, Z, D, gate, dZT0, dgate0, B, L, D, ops.BF16, dsum=dsum0)
    dx1 = torch.full((M, Cin), 3.0, device=dev(), dtype=torch.bfloat16)
    dZT1 = torch.full((D, M), float("nan"), device=dev(), dtype=torch.bfloat16)
    dgate1 = torch.full((B, D), float("nan"), device=dev())
    dsum1 = torch.full((B, D), float("nan"), device=dev())
    d1 = ops.gemm_desc(dY, Wd, dx1, M, D, Cout, taps=3, pad=1, mode=1, Lout=L, Lsrc=L, ldc=Cin, gate=gate, ldg=D,
                       gate_bwd=dict(act=Z, ld_act=D, dct=dZT1, ldt=M, dgate=dgate1, dsum=dsum1))
    ops.gemm_nt([d1], ops.BF16)
    torch.cuda.synchronize()
    assert torch.equal(dZT1, dZT0), "max |d| = %g" % float((dZT1.float() - dZT0.float()).abs().max())
    assert float((dx1.float() - 3.0).abs().max()) == 0.0, "the plain gradient must stay unwritten"
    close(dgate1, dgate0.double().cpu(), 2e-5 * max(1.0, L ** 0.5), "dgate")
    close(dsum1, dsum0.double().cpu(), 2e-5 * max(1.0, L ** 0.5), "dsum")
    # refused where the kernel cannot do it: another kernel, clips that are not 32 / 64 / 128 / 256 rows
    tune(monkeypatch, "nt_w4c", 0)
    with pytest.raises(Exception, match="gb_"):
        ops.gemm_nt([d1], ops.BF16)


@pytest.mark.gpu
def test_sustained_mfma_measurement_is_at_the_issue_floor_and_data_dependent():
    """drn_diag_mfma_sustained (bench.py's `roofline.sustained`): the loop issues one v_mfma_f32_32x32x16_bf16 per 32 cycles and SIMD whatever
    the operands; the CLOCK is what the operands change (power), so zeros never sustain less than random data, and neither exceeds the
    datasheet figure."""
    from drn_amd import ops, _lib
    r = ops.mfma_sustained(5000)
    z = ops.mfma_sustained(5000, zero_operands=True)
    for m in (r, z):
        assert 31.5 < m["cycles_per_mfma"] < 34.0, m
        assert 0.8 < m["clock_ghz"] < 2.45 and 500.0 < m["tflops"] < 2520.0, m
    assert z["tflops"] >= 0.98 * r["tflops"], (r, z)
    with pytest.raises(_lib.DrnError):
        _lib.check(_lib.lib().drn_diag_mfma_sustained(None, 0, 0, None, None, None, None), "drn_diag_mfma_sustained")


@pytest.mark.gpu
@pytest.mark.parametrize("levels,Cin,N", [([(4, 64)], 128, 128), ([(2, 128), (4, 64)], 192, 256), ([(1, 256), (2, 128), (4, 64)], 320, 128),
                                          ([(3, 256)], 448, 256)])
def test_w4h_staged_once_walk_matches_the_convolution(monkeypatch, levels, Cin, N):
    """The k = 3 walk that stages a channel block once for its three taps (W4HX_LOOP_ASM: rows -1 .. 256 of a tile as four 66-row blocks, the
    halo rows zero where a sequence ends) against torch's float64 convolution: forward with statistics and the data gradient, grouped
    levels whose sequences end at block edges (L = 64), inside a wave's rows (128) and nowhere (256), 2 to 7 channel blocks; and the same
    product split over K inside the launch.  Checked to be the walk taken: the tap-major one gives other bits."""
    from drn_amd import ops
    tune(monkeypatch, "nt_w4h", 1)
    w = (rnd((N, Cin, 3), 5, torch.float32) / np.sqrt(Cin * 3)).to(torch.bfloat16)
    for mode in (0, 1):
        res = {}
        for halo in (1, 0):
            tune(monkeypatch, "w4h_halo", halo)
            descs, outs, refs = [], [], []
            for li, (B, L) in enumerate(levels):
                if mode == 0:
                    x = rnd((B, Cin, L), 20 + li, torch.bfloat16)
                    refs.append(F.conv1d(x.double(), w.double(), padding=1).permute(0, 2, 1).reshape(B * L, N))
                    A, Wd, K, Nout = nlc(x).to(dev()), w.permute(0, 2, 1).contiguous().to(dev()), Cin, N
                else:
                    dy = rnd((B, N, L), 30 + li, torch.bfloat16)
                    refs.append(F.conv_transpose1d(dy.double(), w.double(), padding=1).permute(0, 2, 1).reshape(B * L, Cin))
                    A, Wd, K, Nout = nlc(dy).to(dev()), w.permute(1, 2, 0).contiguous().to(dev()), N, Cin
                C = torch.full((B * L, Nout), float("nan"), dtype=torch.bfloat16, device=dev())
                st = torch.full((B * L // 128, 2, Nout), float("nan"), device=dev()) if mode == 0 else None
                descs.append(ops.gemm_desc(A, Wd, C, B * L, Nout, K, taps=3, pad=1, mode=mode, Lout=L, Lsrc=L, stats=st))
                outs.append((A, Wd, C, st))
            if mode == 1 and Cin % 128:          # (the data gradient's output width: 256 x 128 tiles)
                continue
            assert ops.gemm_nt_plan(descs, ops.BF16) == ops.NT_KIND_W4H
            ops.gemm_nt(descs, ops.BF16)
            torch.cuda.synchronize()
            for (_, _, C, st), ref, (B, L) in zip(outs, refs, levels):
                close(C, ref, TOL["bf16"] * 2, "mode %d halo %d L %d" % (mode, halo, L))
                if st is not None:
                    tot, m2 = merged_stats(st, B * L)              # (of the fp32 accumulators, not of the rounded outputs)
                    close(tot, ref.sum(0), TOL["bf16"] * 4, "column sums")
                    close(m2, ((ref - ref.mean(0)) ** 2).sum(0), TOL["bf16"] * 4, "column M2")
            res[halo] = [o[2].clone() for o in outs]
        if len(res) == 2:
            assert any(not torch.equal(a, b) for a, b in zip(res[1], res[0])), "the staged-once walk was not taken"


@pytest.mark.gpu
def test_w4h_staged_once_walk_same_bits_launch_after_launch():
    """A race in the hand-written loop (a wait that counts one load too few, a slot overwritten before its last reader) shows as bits that
    change from launch to launch, and more readily with cold operands and a busy memory system.  The two shapes of the benchmarked step
    that take the walk -- the pyramid-level convolutions (three levels, 224 tiles) and conv0's forward (64 tiles, K split four ways
    inside the launch) -- 30 launches each, a 1 GiB fill between launches: every output equal to the first launch's, bit for bit."""
    from drn_amd import ops
    g = torch.Generator(device="cuda").manual_seed(23)
    big = torch.empty(1 << 28, device=dev())

    def many(descs, outs, n=30):
        first = None
        for it in range(n):
            if it % 3 != 2:
                big.fill_(float(it))               # (two cold launches, one hot)
            ops.gemm_nt(descs, ops.BF16)
            torch.cuda.synchronize()
            got = [o.clone() for o in outs]
            if first is None:
                first = got
                assert all(torch.isfinite(o.float()).all() for o in got)
            else:
                assert all(torch.equal(a, b) for a, b in zip(first, got)), "launch %d differs from launch 0" % it

    W = (torch.randn(512, 3 * 512, generator=g, device=dev()) * 0.03).to(torch.bfloat16)
    descs, outs, keep = [], [], []
    for B, L in ((32, 256), (32, 128), (32, 64)):
        A = torch.randn(B * L, 512, generator=g, device=dev()).to(torch.bfloat16)
        C = torch.empty(B * L, 512, device=dev(), dtype=torch.bfloat16)
        st = torch.empty(B * L // 128, 2, 512, device=dev())
        descs.append(ops.gemm_desc(A, W, C, B * L, 512, 512, taps=3, pad=1, Lout=L, Lsrc=L, stats=st))
        outs += [C, st]
        keep.append(A)                 # (a descriptor holds addresses, not tensors)
    assert ops.gemm_nt_plan(descs, ops.BF16) == ops.NT_KIND_W4H
    many(descs, outs)
    A0 = torch.randn(8192, 4352, generator=g, device=dev()).to(torch.bfloat16)
    W0 = (torch.randn(256, 3 * 4352, generator=g, device=dev()) * 0.01).to(torch.bfloat16)
    C0 = torch.empty(8192, 256, device=dev(), dtype=torch.bfloat16)
    s0 = torch.empty(64, 2, 256, device=dev())
    d0 = ops.gemm_desc(A0, W0, C0, 8192, 256, 4352, taps=3, pad=1, Lout=256, Lsrc=256, stats=s0)
    assert ops._ksplit_w4h([d0], ops.BF16) == 4
    many([d0], [C0, s0])
    assert int(ops._counters(dev()).abs().sum()) == 0
