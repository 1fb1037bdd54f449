"""Generates drn_amd/csrc/gemm_nt_w4_loop.inc: the hand-scheduled main loop of gemm_nt_w4_kernel (gemm_nt_w4.hip) as ONE asm
statement, plus its clobber list.  usage: python scripts/gen_w4_loop.py [--experiments]

Workgroup = 4 waves (one per SIMD, 512 registers each), 256 x 256 output tile, 128 x 128 per wave, bf16 operands, fp32
accumulators in a[0:255] (tile (mi, ni) of the wave's 8 x 8 grid of 16 x 16 MFMA tiles = a[(mi*8+ni)*4 .. +3]).

LDS: a ring of FIVE 32 KB slots (all 160 KB); a slot holds ONE operand's K-step: 256 rows x 128 bytes (64 bf16), the 16-byte chunk
c of row r at position c ^ ((r >> 1) & 7) -- the layout of the general kernel (gemm_nt_kernel.h).  The operand stream is
A_0, B_0, A_1, B_1, ...; item q lives in slot q % 5.  A staging piece is one LDS-DMA wave instruction = 8 rows x 128 bytes = whole
128-byte lines of the source.  (The first version staged 64-byte half rows into 32 KB [A|B] half-step slots: every line was
fetched into the L1 twice, half used each time, and the loads alone took as long as the whole 8-wave kernel -- 217 us against
142 us for the same bytes as whole lines, scripts/experiments/sweep_w4.py.)

K-step j = two half-steps of 64 MFMAs per wave (k-slices 0 and 1 of the 64); the fragments of a k-slice sit in one of two
register sets and are read one half-step ahead:
  half-step 2j   : MFMAs (j, k0) on set 0 | reads (j, k1) -> set 1 out of A_j, B_j       | stages A_{j+2} (8 pieces per wave)
  half-step 2j+1 : MFMAs (j, k1) on set 1 | reads (j+1, k0) -> set 0 out of A/B_{j+1}    | stages B_{j+2}
One barrier per K-step, at the top of half-step 2j+1: before it every wave waits for its own pieces of A/B_{j+1} (vmcnt(8): only
A_{j+2}'s are younger) and for its own fragment reads of K-step j -- so after it A/B_{j+1} are visible to everybody and the slots
of A_j, B_j are free.  A piece is in flight for one to three half-steps before anybody needs it.

Register plan (fixed inside the statement; everything the compiler allocates stays below v124 / outside s[80:91]):
  v[128:159] A fragments set 0   v[160:191] B set 0   v[192:223] A set 1   v[224:255] B set 1   v124 / v125 fragment base of A / B
  s[80:81] / s[82:83] A / B row-panel base + k offset (+128 bytes per K-step), s84 trip count, s85 LDS address of this wave's
  first piece in slot 0, s86 byte offset of the slot being staged, s87 / s88 of the slots being read (A / B), s89-s90 scratch.
"""
import argparse, os

ap = argparse.ArgumentParser()
ap.add_argument("--experiments", action="store_true", help="also emit the ablation / schedule variants (W4_LOOP_ASM_1 ...)")
ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "drn_amd", "csrc", "gemm_nt_w4_loop.inc"))
args = ap.parse_args()


class Cfg:
    ds_every = 4        # one fragment read after every N-th MFMA (16 reads per half-step)
    ds_first = 1
    dma_every = 8       # one LDS-DMA piece after every N-th MFMA (8 pieces per half-step)
    dma_first = 3
    no_dma = False      # ablations (wrong results, timing only)
    no_ds = False
    no_barrier = False
    no_mfma = False

    def __init__(self, **kw):
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)
        self.desc = ", ".join("%s=%s" % kv for kv in sorted(kw.items())) or "shipped"


cfg = Cfg()

SLOT = 32768
NSLOT = 5
RING = SLOT * NSLOT
A_SET = [128, 192]
B_SET = [160, 224]


def frag(base, i):
    return "v[%d:%d]" % (base + 4 * i, base + 4 * i + 3)


def acc(mi, ni):
    b = (mi * 8 + ni) * 4
    return "a[%d:%d]" % (b, b + 3)


def ds_reads(dst_set):
    """the 16 fragment reads of one k-slice into register set dst_set (bases in v124 / v125): A first (the previous half-step's
    last MFMAs still name B registers of this set)."""
    out = ["ds_read_b128 %s, v124 offset:%d" % (frag(A_SET[dst_set], i), i * 2048) for i in range(8)]
    out += ["ds_read_b128 %s, v125 offset:%d" % (frag(B_SET[dst_set], i), i * 2048) for i in range(8)]
    return out


def frag_bases(ks):
    return ["v_add_u32 v124, s87, %%[la%d]" % ks, "v_add_u32 v125, s88, %%[lb%d]" % ks]


def dma_item(op):
    """this wave's 8 LDS-DMA pieces of one operand's K-step into the slot at offset s86 (s89 = s85 + s86)"""
    src = "s[80:81]" if op == "a" else "s[82:83]"
    return [["s_add_u32 m0, s89, %d" % (i * 1024), "s_nop 0", "global_load_lds_dwordx4 %%[vo%s%d], %s" % (op, i, src)] for i in range(8)]


def advance_stage(op):
    ptr = (80, 81) if op == "a" else (82, 83)
    return ["s_add_u32 s%d, s%d, 128" % (ptr[0], ptr[0]), "s_addc_u32 s%d, s%d, 0" % (ptr[1], ptr[1]),
            "s_add_u32 s86, s86, %d" % SLOT, "s_cmp_lt_u32 s86, %d" % RING, "s_cselect_b32 s86, s86, 0", "s_add_u32 s89, s85, s86"]


def advance_read():
    """the slots being read move on by one K-step = two slots (mod the ring)"""
    L = []
    for r in ("s87", "s88"):
        L += ["s_add_u32 %s, %s, %d" % (r, r, 2 * SLOT), "s_sub_u32 s90, %s, %d" % (r, RING), "s_cmp_lt_u32 %s, %d" % (r, RING),
              "s_cselect_b32 %s, %s, s90" % (r, r)]
    return L


def half_step(ks, reads=True, dma=None, wait=None, barrier=False, pre=()):
    """MFMAs of k-slice ks on set ks; `reads`: the fragments of the NEXT k-slice (1 - ks) into the other set; dma: "a" / "b" / None"""
    L = []
    if wait:
        L.append("s_waitcnt %s" % wait)
    if barrier and not cfg.no_barrier:
        L.append("s_barrier")
    L += list(pre)
    do_reads = reads and not cfg.no_ds
    if do_reads:
        L += frag_bases(1 - ks)
    rs = ds_reads(1 - ks) if do_reads else []
    ps = dma_item(dma) if dma and not cfg.no_dma else []
    n = 0
    for mi in range(8):
        for ni in range(8):
            if not cfg.no_mfma:
                L.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (acc(mi, ni), frag(A_SET[ks], mi), frag(B_SET[ks], ni), acc(mi, ni)))
            if rs and n >= cfg.ds_first and (n - cfg.ds_first) % cfg.ds_every == 0:
                L.append(rs.pop(0))
            if ps and n >= cfg.dma_first and (n - cfg.dma_first) % cfg.dma_every == 0:
                L.extend(ps.pop(0))
            n += 1
    L += rs
    for p in ps:
        L.extend(p)
    if dma:
        L += advance_stage(dma)
    return L


def k_step(dma=True, last_wait="vmcnt(8) lgkmcnt(0)", next_reads=True):
    L = half_step(0, reads=True, dma="a" if dma else None, wait="lgkmcnt(0)")
    L += half_step(1, reads=next_reads, dma="b" if dma else None, wait=last_wait, barrier=next_reads, pre=advance_read() if next_reads else ())
    return L


def build():
    lines = ["s_mov_b64 s[80:81], %[sa]", "s_mov_b64 s[82:83], %[sb]", "s_mov_b32 s84, %[cnt]", "s_mov_b32 s85, %[lw]",
             "s_mov_b32 s86, 0", "s_mov_b32 s87, 0", "s_mov_b32 s88, %d" % SLOT, "s_mov_b32 s89, s85"]
    # prologue: A_0, B_0, A_1, B_1 into slots 0..3
    for q in range(4):
        op = "ab"[q & 1]
        for p in dma_item(op):
            lines.extend(p)
        lines += advance_stage(op)
    # accumulators = 0, while the first K-steps are on their way
    for i in range(256):
        lines.append("v_accvgpr_write_b32 a%d, 0" % i)
    lines += ["s_waitcnt vmcnt(16)", "s_barrier"]
    if not cfg.no_ds:
        lines += frag_bases(0) + ds_reads(0)
    # main loop: K/64 - 2 trips of one K-step
    lines += ["s_cmp_eq_u32 s84, 0", "s_cbranch_scc1 L_w4_tail%=", "L_w4_loop%=:"]
    lines += k_step()
    lines += ["s_sub_u32 s84, s84, 1", "s_cmp_lg_u32 s84, 0", "s_cbranch_scc1 L_w4_loop%=", "L_w4_tail%=:"]
    # the last two K-steps: nothing left to stage
    lines += k_step(dma=False, last_wait="vmcnt(0) lgkmcnt(0)")
    lines += k_step(dma=False, last_wait="lgkmcnt(0)", next_reads=False)
    # MFMA results -> v_accvgpr_read in the epilogue: drain the matrix pipe
    lines += ["s_nop 15", "s_nop 15"]
    return lines


clob = ["memory", "scc"] + ["s%d" % i for i in range(80, 91)] + ["v%d" % i for i in range(124, 256)] + ["a%d" % i for i in range(256)]
VARIANTS = [Cfg()]
if args.experiments:
    VARIANTS += [Cfg(no_dma=True), Cfg(no_ds=True), Cfg(no_barrier=True), Cfg(no_dma=True, no_ds=True), Cfg(no_mfma=True),
                 Cfg(ds_every=2, ds_first=0), Cfg(dma_every=4, dma_first=32), Cfg(dma_every=4, dma_first=1), Cfg(ds_every=1, ds_first=0)]
with open(args.out, "w") as f:
    f.write("// GENERATED by scripts/gen_w4_loop.py%s -- do not edit.\n" % (" --experiments" if args.experiments else ""))
    for vi, c in enumerate(VARIANTS):
        cfg = c
        lines = build()
        f.write("// variant %d: %s (%d instructions)\n" % (vi, c.desc, len(lines)))
        f.write("#define W4_LOOP_ASM_%d \\\n" % vi)
        for ln in lines:
            f.write('  "%s\\n\\t" \\\n' % ln)
        f.write('  ""\n')
    f.write("#define W4_VARIANTS %d\n" % len(VARIANTS))
    f.write("#define W4_LOOP_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob))
print("wrote %s: %d variant(s)" % (args.out, len(VARIANTS)))
