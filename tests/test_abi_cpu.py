"""CPU checks of the drop-in boundary: libdrn_hip.so loads and exports every symbol include/drn_hip.h declares;
the product path refuses to run without a GPU / without the library (no CPU fallback, no oracle import)."""
import ast
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from drn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 20, names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.drn_abi_version() == 9
    assert lib.drn_last_error() is not None


def test_product_path_has_no_cpu_fallback():
    from drn_amd import _lib, ops
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg, synthetic_batch
    m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("TINY", 64, 1)))
    with pytest.raises(_lib.DrnError):
        m(*synthetic_batch(2, 32, 64))
    with pytest.raises(_lib.DrnError):
        ops.gemm_desc(torch.zeros(4, 4), torch.zeros(4, 4), torch.zeros(4, 4), 4, 4, 4)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "drn_amd")):
        for f in files:
            if f.endswith(".py"):
                tree = ast.parse(open(os.path.join(dirpath, f)).read())
                for node in ast.walk(tree):
                    mods = []
                    if isinstance(node, ast.Import):
                        mods = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        mods = [node.module or ""]
                    assert not any(m.split(".")[0] == "oracle" for m in mods), (f, mods)


def test_state_dict_keys_match_reference():
    import json
    from drn_amd.model import mainModel
    from drn_amd.utils.synthetic import VOCAB_SIZE, as_namespace, default_cfg
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys.json")))
    m = mainModel(VOCAB_SIZE, as_namespace(default_cfg("C3D", 4096, 1)))
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == ref


def test_generated_w4_loop_is_what_the_generator_emits(tmp_path):
    """drn_amd/csrc/gemm_nt_w4_loop.inc (the hand-scheduled main loop of gemm_nt_w4_kernel / gemm_nt_w4c_kernel, 2 k lines of one
    asm statement) is a GENERATED file: the committed copy must be exactly what scripts/gen_w4_loop.py writes today."""
    import subprocess
    import sys
    out = tmp_path / "loop.inc"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "gen_w4_loop.py"), "--out", str(out)])
    want = open(os.path.join(ROOT, "drn_amd", "csrc", "gemm_nt_w4_loop.inc")).read()
    assert out.read_text() == want, "gemm_nt_w4_loop.inc is stale: run python scripts/gen_w4_loop.py"


def test_iou_loss_module_is_exported_and_has_no_cpu_fallback():
    from drn_amd import _lib
    from drn_amd.model.layers import IOULoss, SigmoidFocalLoss      # noqa: F401  (what model/layers/__init__.py exports on this path)
    with pytest.raises(_lib.DrnError):
        IOULoss()(torch.ones(3, 2), torch.ones(3, 2))


def test_one_launch_batchnorm_backward_plan_is_host_logic():
    """drn_bn_bwd_one_ws_bytes answers from shapes alone (no device call): the launch plan -- which row block, whether the grid fits the
    chip at once, whether a gated level's clips lie inside one row block -- pinned here on the shapes the benchmarked step runs."""
    from drn_amd import _lib
    L = _lib.lib()
    BF16, F32 = 1, 0

    def ws(levels, C, dtype=BF16):
        arr = (_lib.BnBwdDesc * len(levels))()
        for d, (M, gl) in zip(arr, levels):
            d.M = M
            if gl:
                d.gb_dg, d.gb_L = 1, gl              # (any non-NULL pointer: only its presence and the clip length matter to the plan)
        return int(L.drn_bn_bwd_one_ws_bytes(arr, len(levels), C, dtype))
    pyr = [(8192, 0), (4096, 0), (2048, 0)]
    assert ws(pyr, 1024) == 64 + 3 * 65 * 2 * 1024 * 8       # heads stage: 448 workgroups of 512 rows
    assert ws(pyr, 512) > 0 and ws(pyr, 512, F32) > 0
    assert ws(pyr, 1024, F32) == 0                             # fp32: 256-row blocks at most -> 896 workgroups: two launches
    assert ws([(40000, 0)], 1024) == 0                         # 79 row blocks of 512 rows
    assert ws([(8192, 0)], 96) == 0                            # C % 64 != 0
    # gated levels: clips inside one row block, passes inside one clip
    assert ws([(4096, 128)], 512) > 0 and ws([(2048, 64)], 1024) > 0 and ws([(8192, 256)], 256) > 0
    assert ws([(8192, 48)], 256) == 0                          # 48-row clips: not a multiple of a 32-row pass
    assert ws([(8192, 1024)], 256) == 0                        # a clip longer than the largest gated row block (256 rows)
    assert ws([(8192, 256), (4096, 0)], 256) > 0               # (mixing is refused at launch, not by the plan)


def test_norm_pass_block_classes_are_host_logic():
    """drn_sumsq_block_classes: 0 = a block clear of every skipped range, 1 = inside one, 2 = on a boundary (host helper of
    drn_sumsq_partials_skip)."""
    import ctypes
    from drn_amd import _lib
    L = _lib.lib()
    L.drn_opt_nblocks.restype = ctypes.c_int64
    n = 40 * 4096 + 100
    nb = int(L.drn_opt_nblocks(ctypes.c_int64(n)))
    assert nb == 41
    lo = (ctypes.c_int64 * 2)(4096 * 3, 4096 * 10 + 7)
    hi = (ctypes.c_int64 * 2)(4096 * 6, 4096 * 12)
    out = (ctypes.c_ubyte * nb)()
    assert L.drn_sumsq_block_classes(ctypes.c_int64(n), lo, hi, 2, out) == 0
    got = list(out)
    want = [0] * nb
    for b in (3, 4, 5, 11):
        want[b] = 1
    want[10] = 2
    assert got == want, got


def test_split_exchange_confirmation_travels_with_the_call(monkeypatch):
    """The K-split exchanges' confirmation mode is an argument of the launch (DRN_KSPLIT_CONFIRM_* bits of `ksplit`, include/drn_hip.h),
    not process state: no flag = read-back (what ships, everywhere); DRN_XCHG_CONFIRM=1 / 0 select atomics / nothing for stress
    tests.  The product path never calls drn_tune."""
    import re
    from drn_amd import ops
    monkeypatch.setattr(ops, "XCHG_CONFIRM", "2")
    assert ops._ksplit_arg(4) == 4
    monkeypatch.setattr(ops, "XCHG_CONFIRM", "1")
    assert ops._ksplit_arg(4) == 4 | 0x20000
    monkeypatch.setattr(ops, "XCHG_CONFIRM", "0")
    assert ops._ksplit_arg(4) == 4 | 0x80000
    hdr = open(os.path.join(ROOT, "include", "drn_hip.h")).read()
    assert re.search(r"#define DRN_KSPLIT_CONFIRM_ATOMIC\s+0x20000", hdr) and re.search(r"#define DRN_KSPLIT_CONFIRM_NONE\s+0x80000", hdr)
    # drn_tune is a test / experiment switch: only the BatchNorm-backward workgroup budget override (DRN_BN1_MAXWG, an experiment
    # environment variable) may reach it from the package
    for dirpath, _, files in os.walk(os.path.join(ROOT, "drn_amd")):
        for f in files:
            if f.endswith(".py"):
                for i, line in enumerate(open(os.path.join(dirpath, f)).read().split("\n")):
                    if "drn_tune(" in line and not line.strip().startswith("#"):
                        assert "bn1_maxwg" in line, (f, i + 1, line)


def test_library_keeps_no_per_step_state():
    """re-entrant, no mutable globals.  File-scope statics of the library sources: the thread-local error string and the
    test-only tune table, nothing else (function-local `static bool attr_set` one-time attribute flags are idempotent)."""
    import glob
    import re
    found = []
    for f in sorted(glob.glob(os.path.join(ROOT, "drn_amd", "csrc", "*.hip"))):
        for line in open(f):
            if re.match(r"^static\s+(thread_local\s+)?[A-Za-z_][A-Za-z0-9_<>\s\*]*\s+\**g_[a-z_]+(\[\d*\])?\s*(=|;)", line):
                found.append((os.path.basename(f), line.split("=")[0].strip()))
    host = [x for x in found if "__device__" not in x[1]]
    assert host == [("api.hip", "static thread_local char g_err[512]"), ("api.hip", "static int g_tune[16]")], found
    # device side: only the watchdog counters of the in-launch exchanges (diagnostics that drn_amd.ops.check_watchdogs reads and clears)
    assert all(x[1].endswith("_timeouts;") for x in found if "__device__" in x[1]), found
    src = open(os.path.join(ROOT, "drn_amd", "csrc", "elementwise.hip")).read()
    assert "hipMalloc" not in src


def test_two_models_do_not_share_deferred_reduce_state():
    """The deferred weight-gradient reduces live in caller-owned lists (DrnWgradPending, one per GradReducer) that a launch finds by
    where its dW lives: two models in one process own two lists over disjoint gradient buckets; a gradient outside every armed list
    is reduced at once; disarming one leaves the other armed; reset() drops stale items (a backward that raised)."""
    import ctypes
    from drn_amd import _lib, ops

    class FakeGrad(object):
        def __init__(self, ptr):
            self.ptr = ptr

        def data_ptr(self):
            return self.ptr
    a, b = ops.WgradPending([(1000, 2000), (5000, 6000)]), ops.WgradPending([(2000, 3000)])
    assert ctypes.sizeof(a.c) == ctypes.sizeof(_lib.WgradPending) and len(a) == 0 and a.c is not b.c
    assert ops.pending_for([FakeGrad(1500)]) is None                      # nobody armed: reduce at once
    ops.wgrad_arm(a)
    ops.wgrad_arm(b)
    try:
        assert ops.pending_for([FakeGrad(1500)]) is a and ops.pending_for([FakeGrad(5999)]) is a
        assert ops.pending_for([FakeGrad(2000)]) is b and ops.pending_for([FakeGrad(2999)]) is b
        assert ops.pending_for([FakeGrad(3000)]) is None and ops.pending_for([FakeGrad(999)]) is None
        assert ops.pending_for([FakeGrad(1500), FakeGrad(2500)]) is None   # a grouped launch across two owners: not deferred
        a.c.n = 3                                                          # (as if a backward had recorded three and then raised)
        a.ws.append(object())
        a.reset()
        assert len(a) == 0 and a.ws == [] and len(b) == 0
        ops.wgrad_disarm(a)
        assert ops.pending_for([FakeGrad(1500)]) is None and ops.pending_for([FakeGrad(2500)]) is b
    finally:
        ops.wgrad_disarm(a)
        ops.wgrad_disarm(b)
    assert ops._armed == []
    # the struct mirrors include/drn_hip.h: 24 items of (2 pointers + 6 ints) + 25 ints + n + pointer
    assert _lib.WGRAD_PEND_MAX == 24 and ctypes.sizeof(_lib.WgradPendItem) == 40




def test_abi9_argument_checks_answer_before_anything_is_launched():
    """The per-call conventions of ABI 9 are enforced at the boundary -- these calls fail (or succeed trivially) in argument validation,
    before any device work, so they run on a box without a GPU: unknown bits in `ksplit`, a corrupt / an empty DrnWgradPending, a second
    reduce onto an output that already has one recorded."""
    import ctypes
    from drn_amd import _lib
    L = _lib.lib()
    arr = (_lib.GemmDesc * 1)(_lib.GemmDesc())
    assert L.drn_gemm_nt_splitk(arr, 4 | 0x100000, None, None, 1, None) != 0 and b"unknown bits in ksplit" in L.drn_last_error()
    assert L.drn_gemm_nt_splitk_grouped(arr, 1, 2 | 0x10000, None, None, 1, None) != 0
    p = _lib.WgradPending()
    assert L.drn_wgrad_reduce_pending(ctypes.byref(p), None, None) == 0 and L.drn_wgrad_pending_blocks(ctypes.byref(p)) == 0      # empty: nothing to do
    p.n = -1
    assert L.drn_wgrad_reduce_pending(ctypes.byref(p), None, None) != 0 and b"bad list" in L.drn_last_error()
    p.n = 1
    p.it[0].out = 0x1000
    wd = (_lib.WgradDesc * 1)(_lib.WgradDesc(dY=0x2000, X=0x3000, M=128, Lout=128, Lsrc=128, ldy=128, ldx=128))
    rc = L.drn_gemm_wgrad(wd, 1, ctypes.c_void_p(0x1000), 128, 128, 1, 1, 0, 0, 0, None, 1, ctypes.byref(p), None)
    assert rc != 0 and b"already has a deferred reduce pending" in L.drn_last_error()
    assert p.n == 1                                                       # (the list is untouched by the refused call)
