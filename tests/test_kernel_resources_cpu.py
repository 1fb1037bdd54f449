"""Register budgets of the built GEMM kernels, read from the code objects inside libdrn_hip.so (no GPU needed).

Why a test: occupancy is part of these kernels' design and nothing else notices when it is lost.  Round 4 routed the general NT
kernel's epilogue through two helper functions; its bf16 128x128 variant went from 118 to 141 VGPRs -- one 8-wave workgroup per CU
instead of two, the six 448-workgroup launches of a step 13-38 us slower each -- with every parity test green."""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "drn_amd", "libdrn_hip.so")


def kernel_table(tmp_path):
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not found")
    if not os.path.exists(LIB):
        pytest.skip("libdrn_hip.so not built")
    so = os.path.join(str(tmp_path), "lib.so")
    shutil.copy(LIB, so)
    subprocess.run([objdump, "--offloading", so], cwd=str(tmp_path), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    table = {}
    for f in sorted(os.listdir(str(tmp_path))):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", os.path.join(str(tmp_path), f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            get = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
            table[name.group(1)] = {"agpr": int(blk.split()[0]), "vgpr": get("vgpr_count"), "spill": get("vgpr_spill_count"),
                                    "scratch": get("private_segment_fixed_size")}
    assert table, "no gfx950 kernels found in the library"
    return table


def test_gemm_kernels_keep_their_register_budgets(tmp_path):
    table = kernel_table(tmp_path)
    # conv_gemm_nt_kernel<T, STAGES, FAST, WM, WN, MI, NI, BNF = false, CHAIN = false> -- Itanium mangling of the template arguments
    small = [k for k in table if re.search(r"conv_gemm_nt_kernelI(f|DF16b)Li[24]ELb[01]ELi2ELi4ELi4ELi2ELb0ELb0E", k)]
    big = [k for k in table if re.search(r"conv_gemm_nt_kernelI(f|DF16b)Li2ELb[01]ELi2ELi4ELi8ELi4ELb0ELb0E", k)]
    assert len(small) == 8 and len(big) == 4, (len(small), len(big))
    for k in small:      # 128x128 tile, 8 waves: two workgroups per CU need <= 128 registers per lane
        r = table[k]
        assert r["vgpr"] <= 128 and r["spill"] == 0 and r["scratch"] == 0 and r["agpr"] == 0, (k, r)
    for k in big:        # 256x256 tile, 8 waves = 2 per SIMD: <= 256, and nothing in scratch
        r = table[k]
        assert r["vgpr"] <= 256 and r["spill"] == 0 and r["scratch"] == 0, (k, r)
    w4 = [k for k in table if "gemm_nt_w4_kernel" in k or "gemm_nt_w4c_kernel" in k]
    assert len(w4) >= 2, "gemm_nt_w4_kernel / gemm_nt_w4c_kernel missing"
    for k in w4:         # one wave per SIMD: the statement owns a[0:255] and v[124:255]; the compiler must not spill around it
        r = table[k]
        assert r["agpr"] == 256 and r["vgpr"] == 512 and r["spill"] == 0 and r["scratch"] == 0, (k, r)


def test_one_launch_batchnorm_backward_holds_its_rows_in_registers(tmp_path):
    """bn_bwd_one_kernel<T, NP> keeps 2 x NP 16-byte vectors per thread between its two halves and waits for its siblings in between:
    a variant that spills is refused at launch (bn_resident_capacity), and past 256 registers only one workgroup per CU would fit."""
    table = kernel_table(tmp_path)
    ks = [k for k in table if "bn_bwd_one_kernel" in k]
    assert len(ks) == 14, ks            # 2 dtypes x NP in {2, 4, 8, 16} + the gated variants (NP <= 8)
    for k in ks:
        r = table[k]
        assert r["vgpr"] <= 256 and r["spill"] == 0 and r["scratch"] == 0, (k, r)


def test_compiler_never_touches_the_accumulators_of_the_hand_scheduled_kernels(tmp_path):
    """gemm_nt_w4 / w4c / w4h keep their results in AGPRs that only inline asm reads and writes -- the compiler does not know they are
    live after the loop statement and is free to park its own values there (gfx950: a unified 512-register file, `v_accvgpr_write` as a
    cheap spill).  Round 6 met exactly that: with the statistics folded into the store pass of the 256 x 256 epilogue it postponed 85
    additions and kept their operands in a58..a71 -- accumulator tiles not stored yet; wrong outputs, zero spills in the metadata.
    So: compile the two files to assembly and refuse ANY AGPR access outside the asm statements."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "drn_amd", "csrc")
    procs = []
    for name in ("gemm_nt_w4", "gemm_nt_w4h"):
        out = os.path.join(str(tmp_path), name + ".s")
        procs.append((name, out, subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
                                                   "-Wno-unused-function", "--cuda-device-only", "-S", "-o", out, name + ".hip"],
                                                  cwd=csrc, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    for name, out, p in procs:
        assert p.wait(timeout=900) == 0, name
        inasm, bad, statements = False, [], 0
        for ln in open(out):
            if "#ASMSTART" in ln:
                inasm, statements = True, statements + 1
            elif "#ASMEND" in ln:
                inasm = False
            elif not inasm and "v_accvgpr_" in ln:
                bad.append(ln.strip())
        assert statements > 100, (name, statements)          # (the accumulator reads of the epilogues are asm statements of their own)
        assert not bad, (name, len(bad), bad[:8])
        # ... and no FLAT memory instruction: the descriptor's addresses are assembled from lanes (nt_fetch), which makes them generic
        # pointers unless as_global() says otherwise -- 2000 flat loads + 1856 flat stores in gemm_nt_w4.hip before round 6; a FLAT access
        # counts on the LDS counter too, so every wait for the staging patch waited for the tile stores in flight (common.h)
        flat = [ln.strip() for ln in open(out) if ln.lstrip().startswith(("flat_load", "flat_store", "flat_atomic"))]
        assert not flat, (name, len(flat), flat[:4])
