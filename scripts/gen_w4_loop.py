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

Conv mode (W4C_LOOP_ASM, gemm_nt_w4c_kernel): k = 3 / stride 1 / pad 1 convolutions, forward and data gradient.  The A operand of
K-step j is channel block c0(j) of the source rows shifted by the tap of j; A is staged through a buffer descriptor
(`buffer_load_dwordx4 v, s[92:95], s96 offen lds`): v[116:123] = the lanes' row offsets for the CURRENT tap of the staging stream,
0x80000000 -- out of range, the hardware writes zeros into those LDS cells -- where the shifted row leaves the lane's sequence;
s96 = channel byte offset inside the tap.  After every staged A item one K-step less is left in the tap (s97); at zero the stream
moves to the next tap: s96 = 0, new row shift (%[sh0..2]), new zero lanes (bit i of %[ma] / %[mc] for piece i at the first / last
tap).  Nothing else differs from the plain loop: MFMAs, fragment reads, B staging, waits and barriers do not know about taps.

Register plan (fixed inside the statement; everything the compiler allocates stays below v113 / outside s[79:99]):
  v[128:159] A fragments set 0   v[160:191] B set 0   v[192:223] A set 1   v[224:255] B set 1   v124 / v125 fragment base of A / B
  s[80:81] / s[82:83] A / B row-panel base + k offset (+128 bytes per K-step), s84 trip count, s85 LDS address of this wave's
  first piece in slot 0, s86 byte offset of the slot being staged, s87 / s88 of the slots being read (A / B), s89 = s85 + s86,
  s90 / s91 scratch; conv mode: s[92:95] A descriptor, s96-s99 tap state, s79 scratch, v114 scratch, v115 = 0x80000000,
  v[116:123] lane offsets of the current tap.
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
    conv = False        # k = 3 / stride 1 convolutions: A staged through a buffer descriptor, per-tap row shift, zero rows at the sequence edges
    mfma32 = False      # timing-only ablation: half as many v_mfma_f32_32x32x16_bf16 (same pipe time, twice the issue slack per gap)
    hoist = True        # scalar bookkeeping and M0 writes inside the MFMA stream (False: after it / in front of each piece)
    adv = 128           # bytes the operand pointers advance per K-step (0: every K-step re-reads the first one -- L2-hit ablation)
    a_pieces = 8        # timing-only ablation (W4_A_PIECES=3 in the environment): conv mode stages only this many of an A item's 8 pieces
                        # -- a third of the A traffic, what staging a channel block ONCE for its three taps would load (wrong results)
    tapil = False       # conv mode with the taps INTERLEAVED: K-step j = (channel block j / 3, tap j % 3) -- see advance_tap_il
    half = False        # 256 x 128 output tile (gemm_nt_w4h_kernel): 128 x 64 per wave, B items of 128 rows -- see the notes at `Geo`
    swap = False        # MFMA operands exchanged (weights fragment first): the 16 x 16 accumulator tiles come out TRANSPOSED -- lane l holds
                        # C[16 mi + (l & 15)][16 ni + 4 (l >> 4) + r], four consecutive COLUMNS of one row, which is what a row-major bf16
                        # store wants (8 bytes per lane, no transposition through DPP / LDS in the epilogue); same products, same K order

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


class Geo:
    """Tile geometry of the variant being generated.  Full (shipped W4 / W4C): 256 x 256 per workgroup, 128 x 128 per wave (8 x 8 MFMA
    tiles, a[0:255]), A and B items of 256 rows = 32 KB each, five 32 KB slots.
    Half (W4H / W4HC, cfg.half): 256 x 128 per workgroup for the launches whose N = 512 gives too few 256 x 256 tiles to fill the
    chip (the pyramid-level convolutions: 112 of them on 256 CUs) and which are LDS-bound on 128 x 128 tiles of eight 64 x 32
    waves (96 KB of fragment reads per 128 x 128 x 64 MACs: 1024 LDS cycles per K-step and workgroup against 512 of MFMA).  Here a
    wave owns 128 x 64 (8 x 4 MFMA tiles, a[0:127]): 24 KB of fragment reads per 128 x 64 x 64 MACs -- 1152 LDS cycles per K-step
    and CU against 1024 of MFMA.  A items stay 256 rows = 32 KB (8 pieces per wave), B items are 128 rows = 16 KB (4 pieces per
    wave); the ring holds THREE K-steps as [A | B] pairs of 48 KB (144 KB): K-step j lives in pair j % 3, its A at +0, its B at
    +32 KB.  Same schedule: A_{j+2} is staged during half-step 2j into the pair K-step j-1 left at the barrier of half-step 2j-1,
    B_{j+2} during half-step 2j+1.  The stream offset s86 advances by 32 KB after an A item and by 16 KB after a B item and wraps
    at 144 KB; the wave's first piece sits at s85 + s86 for A items (s85 = LDS base + 8 KB x wave) and at s100 + s86 for B items
    (s100 = LDS base + 4 KB x wave)."""

    def __init__(self, half):
        self.half = half
        self.NI = 4 if half else 8
        self.PB = 4 if half else 8                      # B pieces per wave and K-step
        self.A_BYTES = SLOT
        self.B_BYTES = SLOT // 2 if half else SLOT
        self.RING = 3 * (SLOT + SLOT // 2) if half else RING
        self.NM = 8 * self.NI                           # MFMAs per half-step
        self.NACC = 32 * self.NI                        # accumulator registers


def geo():
    return Geo(cfg.half)


def frag(base, i):
    return "v[%d:%d]" % (base + 4 * i, base + 4 * i + 3)


def acc(mi, ni):
    b = (mi * geo().NI + ni) * 4
    return "a[%d:%d]" % (b, b + 3)


def ds_reads(dst_set):
    """the 16 fragment reads of one k-slice into register set dst_set (bases in v124 / v125): A first (the previous half-step's
    last MFMAs still name B registers of this set)."""
    out = ["ds_read_b128 %s, v124 offset:%d" % (frag(A_SET[dst_set], i), i * 2048) for i in range(8)]
    out += ["ds_read_b128 %s, v125 offset:%d" % (frag(B_SET[dst_set], i), i * 2048) for i in range(geo().NI)]
    return out


def frag_bases(ks):
    return ["v_add_u32 v124, s87, %%[la%d]" % ks, "v_add_u32 v125, s88, %%[lb%d]" % ks]


def dma_item(op):
    """this wave's 8 LDS-DMA pieces of one operand's K-step into the slot at offset s86 (s89 = s85 + s86)"""
    if op == "a" and cfg.conv:
        # conv mode: A through a buffer descriptor (s[92:95]): lane offset v[116 + i] = row offset of the CURRENT tap, 0x80000000
        # (out of range -> the piece gets zeros) where the tap leaves the lane's sequence; s96 = channel byte offset inside the tap
        soff = "s98" if cfg.tapil else "s96"       # (interleaved taps: s98 = channel offset + the tap's row shift)
        return [["s_add_u32 m0, s89, %d" % (i * 1024), "s_nop 0", "buffer_load_dwordx4 v%d, s[92:95], %s offen lds" % (116 + i, soff)]
                for i in range(int(os.environ.get("W4_A_PIECES", cfg.a_pieces)))]
    src = "s[80:81]" if op == "a" else "s[82:83]"
    return [["s_add_u32 m0, s89, %d" % (i * 1024), "s_nop 0", "global_load_lds_dwordx4 %%[vo%s%d], %s" % (op, i, src)]
            for i in range(8 if op == "a" else geo().PB)]


def apply_tap(tag):
    """conv mode: lane offsets of the staging stream's tap s99 (0 / 1 / 2) into v[116:123]: row shift %[sh0..2] on top of the lane's
    own offset, then 0x80000000 (v115) for the lanes whose source row leaves the sequence -- tap 0: bit i of %[ma], tap 2: of %[mc]"""
    L = ["s_cmp_eq_u32 s99, 0", "s_cselect_b32 s79, %[sh0], %[sh1]", "s_cmp_eq_u32 s99, 2", "s_cselect_b32 s79, %[sh2], s79"]
    L += ["v_add_u32 v%d, s79, %%[voa%d]" % (116 + i, i) for i in range(8)]
    L += ["s_cmp_eq_u32 s99, 1", "s_cbranch_scc1 L_w4_tapdone_%s%%=" % tag, "s_cmp_eq_u32 s99, 0", "s_cbranch_scc0 L_w4_tap2_%s%%=" % tag]
    for name, msk in (("tap0", "%[ma]"), ("tap2", "%[mc]")):
        if name == "tap2":
            L.append("L_w4_tap2_%s%%=:" % tag)
        for i in range(8):
            L += ["v_and_b32 v114, %d, %s" % (1 << i, msk), "v_cmp_ne_u32 vcc, 0, v114", "v_cndmask_b32 v%d, v%d, v115, vcc" % (116 + i, 116 + i)]
        if name == "tap0":
            L.append("s_branch L_w4_tapdone_%s%%=" % tag)
    L.append("L_w4_tapdone_%s%%=:" % tag)
    return L


_tap_seq = [0]


def advance_tap():
    """conv mode, after an A item: one K-step less in the staging stream's tap; at zero the stream moves to the next tap
    (s97 = K-steps left in the tap, s98 = K-steps per tap, s96 = channel byte offset)"""
    _tap_seq[0] += 1
    tag = "t%d" % _tap_seq[0]
    L = ["s_sub_u32 s97, s97, 1", "s_cmp_lg_u32 s97, 0", "s_cbranch_scc1 L_w4_same_%s%%=" % tag,
         "s_mov_b32 s97, s98", "s_mov_b32 s96, 0", "s_add_u32 s99, s99, 1"]
    L += apply_tap(tag)
    L.append("L_w4_same_%s%%=:" % tag)
    return L


def tap_il_select():
    """interleaved taps: soffset s98 = s96 + row shift of tap s99; lane offsets v[116:123] = the tap's set -- tap 1: the lanes' own
    offsets (%[voa..]), tap 0 / 2: v[97:104] / v[105:112], the same with 0x80000000 where the shifted row leaves the lane's sequence"""
    L = ["s_cmp_eq_u32 s99, 0", "s_cselect_b32 s79, %[sh0], %[sh1]", "s_cmp_eq_u32 s99, 2", "s_cselect_b32 s79, %[sh2], s79",
         "s_add_u32 s98, s96, s79", "s_cmp_eq_u32 s99, 0", "s_cselect_b64 vcc, -1, 0"]
    L += ["v_cndmask_b32 v%d, %%[voa%d], v%d, vcc" % (116 + i, i, 97 + i) for i in range(8)]
    L += ["s_cmp_eq_u32 s99, 2", "s_cselect_b64 vcc, -1, 0"]
    L += ["v_cndmask_b32 v%d, v%d, v%d, vcc" % (116 + i, 116 + i, 105 + i) for i in range(8)]
    return L


def advance_tap_il():
    """conv mode with interleaved taps, after an A item: the staging stream moves to the next tap of the SAME channel block, and to
    the next channel block (+128 bytes) after tap 2.  The three taps of a channel block read rows m-1 .. m+256 of the same 128-byte
    column one K-step after the other: the second and third find the lines in L2.  (Tap-major order reads each of conv0's 71 MB
    three times 68 K-steps apart, every 128-byte piece in a DRAM page of its own.)"""
    L = ["s_add_u32 s99, s99, 1", "s_cmp_eq_u32 s99, 3", "s_cselect_b32 s99, 0, s99",
         "s_cmp_eq_u32 s99, 0", "s_cselect_b32 s79, 128, 0", "s_add_u32 s96, s96, s79"]
    return L + tap_il_select()


def advance_b_il():
    """... and the B stream (its own tap counter s97): + one tap (%[dstep] = Cin * 2 bytes) twice, then back two taps and on by one
    channel block (%[dwrap] = 128 - 2 * Cin * 2, negative: the high word gets -1)"""
    return ["s_add_u32 s97, s97, 1", "s_cmp_eq_u32 s97, 3", "s_cselect_b32 s97, 0, s97",
            "s_cmp_eq_u32 s97, 0", "s_cselect_b32 s79, %[dwrap], %[dstep]", "s_cselect_b32 s101, -1, 0",
            "s_add_u32 s82, s82, s79", "s_addc_u32 s83, s83, s101"]


def advance_stage_groups(op):
    """source pointer of the operand + 128 bytes, staging slot + 1 (mod the ring); groups of instructions that stay adjacent
    (producer and consumer of SCC)"""
    ptr = (80, 81) if op == "a" else (82, 83)
    first = ["s_add_u32 s%d, s%d, %d" % (ptr[0], ptr[0], cfg.adv), "s_addc_u32 s%d, s%d, 0" % (ptr[1], ptr[1])]
    if op == "a" and cfg.conv:
        first = ["s_add_u32 s96, s96, 128"]
    if cfg.tapil:
        first = ["s_nop 0"] if op == "a" else advance_b_il()      # (A: the channel offset moves in advance_tap_il)
    G = geo()
    if G.half:      # the item after an A item is a B item (wave base s100) and vice versa (s85)
        return [first, ["s_add_u32 s86, s86, %d" % (G.A_BYTES if op == "a" else G.B_BYTES)],
                ["s_cmp_lt_u32 s86, %d" % G.RING, "s_cselect_b32 s86, s86, 0"], ["s_add_u32 s89, %s, s86" % ("s100" if op == "a" else "s85")]]
    return [first, ["s_add_u32 s86, s86, %d" % SLOT], ["s_cmp_lt_u32 s86, %d" % RING, "s_cselect_b32 s86, s86, 0"], ["s_add_u32 s89, s85, s86"]]


def advance_stage(op):
    return [i for g in advance_stage_groups(op) for i in g]


def advance_read_groups():
    """the slots being read move on by one K-step = two slots (mod the ring)"""
    G = []
    g = geo()
    step, ring = (g.A_BYTES + g.B_BYTES, g.RING) if g.half else (2 * SLOT, RING)
    for r, t in (("s87", "s90"), ("s88", "s91")):
        G += [["s_add_u32 %s, %s, %d" % (r, r, step), "s_sub_u32 %s, %s, %d" % (t, r, ring)],
              ["s_cmp_lt_u32 %s, %d" % (r, ring), "s_cselect_b32 %s, %s, %s" % (r, r, t)]]
    return G


def half_step(ks, reads=True, dma=None, wait=None, barrier=False, adv_read=False):
    """MFMAs of k-slice ks on set ks; `reads`: the fragments of the NEXT k-slice (1 - ks) into the other set; dma: "a" / "b" / None;
    adv_read: move the read slots on by one K-step once this half-step's fragment bases have been formed.
    Everything that is not an MFMA goes INTO the MFMA stream (one or two instructions per gap: a 16-cycle MFMA leaves ~3 issue
    slots): the scalar bookkeeping, and M0 is written one MFMA ahead of its LDS-DMA piece (the MFMA is the wait state)."""
    L = []
    if wait:
        L.append("s_waitcnt %s" % wait)
    if barrier and not cfg.no_barrier:
        L.append("s_barrier")
    do_reads = reads and not cfg.no_ds
    if do_reads:
        L += frag_bases(1 - ks)
    NM, NI = geo().NM, geo().NI
    fill = {n: [] for n in range(NM)}
    rs = ds_reads(1 - ks) if do_reads else []
    for k, r in enumerate(rs):
        fill[min(NM - 1, cfg.ds_first + k * cfg.ds_every)].append(r)
    ps = dma_item(dma) if dma and not cfg.no_dma else []
    last_dma = 0
    for k, (m0w, nop, ld) in enumerate(ps):
        n = min(NM - 1, cfg.dma_first + k * cfg.dma_every)
        if cfg.hoist and n >= 1 and not cfg.no_mfma:
            fill[n - 1].append(m0w)
            fill[n].append(ld)
        else:
            fill[n] += [m0w, nop, ld]
        last_dma = n
    tail = []
    rd_groups = advance_read_groups() if adv_read else []
    st_groups = advance_stage_groups(dma) if dma else []
    if cfg.hoist:
        # the read slots may move as soon as the bases are formed; the staging pointers only after the last piece has been issued
        for k, g in enumerate(rd_groups):
            fill[4 + 2 * k] += g
        pos = last_dma + 1
        for g in st_groups:
            if pos <= NM - 1:
                fill[pos] += g
                pos += 1
            else:
                tail += g
    else:
        tail = [i for g in rd_groups + st_groups for i in g]
    if dma == "a" and cfg.conv:
        tail += advance_tap_il() if cfg.tapil else advance_tap()
    for n in range(NM):
        if not cfg.no_mfma:
            mi, ni = n // NI, n % NI
            if cfg.mfma32:
                if n % 2 == 0:
                    q = (n // 2) % 16
                    L.append("v_mfma_f32_32x32x16_bf16 a[%d:%d], %s, %s, a[%d:%d]" % (16 * q, 16 * q + 15, frag(A_SET[ks], mi), frag(B_SET[ks], ni), 16 * q, 16 * q + 15))
            else:
                ops_ = (frag(B_SET[ks], ni), frag(A_SET[ks], mi)) if cfg.swap else (frag(A_SET[ks], mi), frag(B_SET[ks], ni))
                L.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (acc(mi, ni), ops_[0], ops_[1], acc(mi, ni)))
        L += fill[n]
    L += tail
    return L


def k_step(dma=True, last_wait="vmcnt(8) lgkmcnt(0)", next_reads=True):
    L = half_step(0, reads=True, dma="a" if dma else None, wait="lgkmcnt(0)", adv_read=next_reads)
    L += half_step(1, reads=next_reads, dma="b" if dma else None, wait=last_wait, barrier=next_reads)
    return L


def build():
    _tap_seq[0] = 0
    G = geo()
    lines = ["s_mov_b64 s[82:83], %[sb]", "s_mov_b32 s84, %[cnt]", "s_mov_b32 s85, %[lw]",
             "s_mov_b32 s86, 0", "s_mov_b32 s87, 0", "s_mov_b32 s88, %d" % SLOT, "s_mov_b32 s89, s85"]
    if G.half:
        lines += ["s_mov_b32 s100, %[lwb]"]
    if cfg.conv:
        lines += ["s_mov_b32 s92, %[d0]", "s_mov_b32 s93, %[d1]", "s_mov_b32 s94, %[d2]", "s_mov_b32 s95, %[d3]", "s_mov_b32 s96, %[c0]"]
        if cfg.tapil:
            lines += ["s_mov_b32 s97, 0", "s_mov_b32 s99, 0", "v_mov_b32 v115, 0x80000000"]
            for base, msk in ((97, "%[ma]"), (105, "%[mc]")):      # the lane offsets of tap 0 / tap 2: zero rows at the sequence edges
                for i in range(8):
                    lines += ["v_and_b32 v114, %d, %s" % (1 << i, msk), "v_cmp_ne_u32 vcc, 0, v114",
                              "v_cndmask_b32 v%d, %%[voa%d], v115, vcc" % (base + i, i)]
            lines += ["s_nop 1"] + tap_il_select()
        else:
            lines += ["s_mov_b32 s97, %[left]", "s_mov_b32 s98, %[per]", "s_mov_b32 s99, %[tap]", "v_mov_b32 v115, 0x80000000"]
            lines += apply_tap("init")
        lines += ["s_nop 4"]          # v[116:123] / s96 written just above: settle before the first piece reads them
    else:
        lines += ["s_mov_b64 s[80:81], %[sa]"]
    # prologue: A_0, B_0, A_1, B_1 into slots 0..3
    for q in range(4):
        op = "ab"[q & 1]
        for p in dma_item(op):
            lines.extend(p)
        lines += advance_stage(op)
        if op == "a" and cfg.conv:
            lines += advance_tap_il() if cfg.tapil else advance_tap()
    # accumulators = 0, while the first K-steps are on their way
    for i in range(G.NACC):
        lines.append("v_accvgpr_write_b32 a%d, 0" % i)
    lines += ["s_waitcnt vmcnt(%d)" % (8 + G.PB), "s_barrier"]
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


# ---------------------------------------------------------------------------------------------------------------------------
# k = 3 convolutions with the channel block staged ONCE for its three taps (W4HX_LOOP_ASM, gemm_nt_w4h_kernel; round 6).
# The tap loops above stage an A item per K-step: the three taps of a channel block load the same rows shifted by one -- 3 x the
# A traffic, and with cold operands these loops run at half the MFMA issue rate on what the L2 delivers (conv0's forward: 2170 cycles
# per K-step against a floor of 1024; with 3 of 8 A pieces staged -- timing only -- the loop went from 57.2 to 41.5 us).  Here:
#   * K order (channel block, tap); the loop body is one channel block = three K-steps with the tap static in each;
#   * an A item = the 64 channels of rows -1 .. 256 of the tile as FOUR blocks, one per wave: block b holds positions 0 .. 71 =
#     source rows 64 b - 1 .. 64 b + 70 of the tile in NINE 1 KB pieces (positions 66 .. 71 are never read); position p of a block
#     sits at p * 128 bytes, its 16-byte chunk c at (c ^ ((p >> 1) & 7)) * 16; halo positions 0 / 65 whose row lies in another
#     sequence come in as zeros (lane offset 0x80000000 -> out of range), which is all the padding there is (L % 64 == 0);
#   * tap t (0 .. 2) of output row r reads position (r % 64) + t of block r / 64: a per-lane base per (t, k-slice) -- six operands
#     %[la<t><ks>] -- because the XOR swizzle follows the position; the 16-row steps keep immediate offsets (2048 per 16 rows, 9216
#     per block);
#   * rings: TWO A slots of 36 KB (A(cb + 1) is staged while cb is multiplied: 5 pieces in tap 0's first half-step, 4 in tap 1's;
#     needed at tap 2's barrier) and FIVE B slots of 16 KB behind them: B_{j+4} is staged during K-step j -- four K-steps of latency
#     cover instead of one (the first version, 3 + 3 slots with B_{j+2}, gained a third of what the traffic ablation promised);
#   * waits: at the barrier of K-step j the wave needs its own pieces of B_{j+1}, and of A(cb + 1) when j is a tap 2.  Loads complete
#     in order, so the wait is vmcnt(number of loads issued after the last one needed) -- computed by SIMULATING the issue sequence
#     for 2 .. 7 channel blocks and taking, per emitted wait, the smallest count any execution needs (hx_waits).
# Registers: s[82:83] B pointer, s84 trips, s85 / s100 the wave's A / B staging base, s86 / s97 A / B staging slot, s87 / s88 A / B
# read slot, s89 / s98 = s85 + s86 / s100 + s97, s[92:95] A descriptor, s96 channel byte offset of the item being staged.
HX_ABLK, HX_ASLOT, HX_NA, HX_BSLOT, HX_NB, HX_BAHEAD = 9216, 36864, 2, 16384, 5, 4
HX_ARING, HX_BRING = HX_ASLOT * HX_NA, HX_BSLOT * HX_NB
HX_A_SPLIT = {0: range(0, 5), 1: range(5, 9), 2: range(0)}          # A(cb + 1)'s pieces by the tap whose first half-step issues them


def hx_a_pieces(ks_list):
    return [["s_add_u32 m0, s89, %d" % (k * 1024), "s_nop 0", "buffer_load_dwordx4 %%[voa%d], s[92:95], s96 offen lds" % k] for k in ks_list]


def hx_b_pieces():
    return [["s_add_u32 m0, s98, %d" % (i * 1024), "s_nop 0", "global_load_lds_dwordx4 %%[vob%d], s[82:83]" % i] for i in range(geo().PB)]


def hx_adv_a_stage():
    return [["s_add_u32 s96, s96, 128"], ["s_add_u32 s86, s86, %d" % HX_ASLOT], ["s_cmp_lt_u32 s86, %d" % HX_ARING, "s_cselect_b32 s86, s86, 0"],
            ["s_add_u32 s89, s85, s86"]]


def hx_adv_b_stage(staged_tap):
    """after a B item of tap `staged_tap`: the pointer moves one tap on (%[dstep] = Cin * 2 bytes), or from tap 2 back two taps and on
    by one channel block (%[dwrap] = 128 - 2 * Cin * 2 < 0: the high word takes the borrow)"""
    ptr = ["s_add_u32 s82, s82, %[dwrap]", "s_addc_u32 s83, s83, -1"] if staged_tap == 2 else ["s_add_u32 s82, s82, %[dstep]", "s_addc_u32 s83, s83, 0"]
    return [ptr, ["s_add_u32 s97, s97, %d" % HX_BSLOT], ["s_cmp_lt_u32 s97, %d" % HX_BRING, "s_cselect_b32 s97, s97, 0"], ["s_add_u32 s98, s100, s97"]]


def hx_adv_read(which):
    r, t, step, ring = ("s87", "s90", HX_ASLOT, HX_ARING) if which == "a" else ("s88", "s91", HX_BSLOT, HX_BRING)
    return [["s_add_u32 %s, %s, %d" % (r, r, step), "s_sub_u32 %s, %s, %d" % (t, r, ring)], ["s_cmp_lt_u32 %s, %d" % (r, ring), "s_cselect_b32 %s, %s, %s" % (r, r, t)]]


def hx_reads(dst_set):
    out = ["ds_read_b128 %s, v124 offset:%d" % (frag(A_SET[dst_set], i), (i >> 2) * HX_ABLK + (i & 3) * 2048) for i in range(8)]
    out += ["ds_read_b128 %s, v125 offset:%d" % (frag(B_SET[dst_set], i), i * 2048) for i in range(geo().NI)]
    return out


def hx_half_step(ks, read_tap, pieces, wait, barrier, groups_early, groups_late):
    """MFMAs of k-slice ks out of register set ks; read_tap: None or the tap whose k-slice (1 - ks) fragments go into the other set;
    pieces: LDS-DMA pieces issued inside the MFMA stream; groups_early: scalar groups that may run once the fragment bases exist;
    groups_late: after the last piece."""
    L = []
    if wait:
        L.append("s_waitcnt %s" % wait)
    if barrier:
        L.append("s_barrier")
    if read_tap is not None:
        L += ["v_add_u32 v124, s87, %%[la%d%d]" % (read_tap, 1 - ks), "v_add_u32 v125, s88, %%[lb%d]" % (1 - ks)]
    NM, NI = geo().NM, geo().NI
    fill = {n: [] for n in range(NM)}
    for k, r in enumerate(hx_reads(1 - ks) if read_tap is not None else []):
        fill[min(NM - 1, cfg.ds_first + k * cfg.ds_every)].append(r)
    last = 0
    for k, (m0w, nop, ld) in enumerate(pieces):
        n = min(NM - 1, cfg.dma_first + k * cfg.dma_every)
        fill[n - 1].append(m0w)
        fill[n].append(ld)
        last = n
    for k, g in enumerate(groups_early):
        fill[4 + 2 * k] += g
    pos, tail = last + 1, []
    for g in groups_late:
        if pos <= NM - 1:
            fill[pos] += g
            pos += 1
        else:
            tail += g
    for n in range(NM):
        mi, ni = n // NI, n % NI
        L.append("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (acc(mi, ni), frag(B_SET[ks], ni), frag(A_SET[ks], mi), acc(mi, ni)))
        L += fill[n]
    return L + tail


def hx_program(ncb):
    """the K-steps one workgroup executes for ncb channel blocks: (site, tap, stage_a, stage_b); site = which emitted copy runs"""
    out = []
    for cb in range(ncb):
        site = "main" if cb <= ncb - 3 else ("tail1" if cb == ncb - 2 else "tail2")
        for tap in range(3):
            j = 3 * cb + tap
            out.append((site, tap, cb + 1 <= ncb - 1, j + HX_BAHEAD <= 3 * ncb - 1))
    return out


def hx_waits():
    """vmcnt at the barrier of every emitted K-step: loads issued after the last one the next K-step needs, the minimum over every execution"""
    need = {}
    for ncb in range(2, 8):
        log = [("A", 0)] * 9
        for jb in range(HX_BAHEAD):
            log += [("B", jb)] * geo().PB
        for j, (site, tap, sa, sb) in enumerate(hx_program(ncb)):
            cb = j // 3
            if sa:
                log += [("A", cb + 1)] * len(HX_A_SPLIT[tap])
            if j + 1 <= 3 * ncb - 1:
                req = [("B", j + 1)] + ([("A", cb + 1)] if tap == 2 else [])
                last = max(i for i, e in enumerate(log) if e in req)
                n = len(log) - 1 - last
                need[(site, tap)] = min(need.get((site, tap), 63), n)
            if sb:
                log += [("B", j + HX_BAHEAD)] * geo().PB
    # a site must stage the same things in every execution (the emitted code is one)
    for ncb in range(2, 8):
        for site, tap, sa, sb in hx_program(ncb):
            assert (sa, sb) == hx_site_stages(site, tap), (ncb, site, tap)
    return need


def hx_site_stages(site, tap):
    if site == "main":
        return True, True
    if site == "tail1":
        return True, tap <= 1
    return False, False


def hx_k_step(site, tap, waits, next_reads=True):
    """K-step (cb, tap) of the emitted copy `site`"""
    sa, sb = hx_site_stages(site, tap)
    a_ps = hx_a_pieces(HX_A_SPLIT[tap]) if sa else []
    early = hx_adv_read("b") + (hx_adv_read("a") if tap == 2 else []) if next_reads else []
    L = hx_half_step(0, tap, a_ps, "lgkmcnt(0)", False, early, hx_adv_a_stage() if (a_ps and tap == 1) else [])
    if next_reads:
        L += hx_half_step(1, (tap + 1) % 3, hx_b_pieces() if sb else [], "vmcnt(%d) lgkmcnt(0)" % waits[(site, tap)], True, [],
                          hx_adv_b_stage((tap + HX_BAHEAD) % 3) if sb else [])
    else:
        L += hx_half_step(1, None, [], "lgkmcnt(0)", False, [], [])
    return L


def build_halo():
    G = geo()
    assert G.half and cfg.swap
    waits = hx_waits()
    lines = ["s_mov_b64 s[82:83], %[sb]", "s_mov_b32 s84, %[cnt]", "s_mov_b32 s85, %[lw]", "s_mov_b32 s100, %[lwb]",
             "s_mov_b32 s86, 0", "s_mov_b32 s97, 0", "s_mov_b32 s87, 0", "s_mov_b32 s88, 0", "s_mov_b32 s89, s85", "s_mov_b32 s98, s100",
             "s_mov_b32 s92, %[d0]", "s_mov_b32 s93, %[d1]", "s_mov_b32 s94, %[d2]", "s_mov_b32 s95, %[d3]", "s_mov_b32 s96, %[c0]", "s_nop 4"]
    # prologue: A(cb0), B_0 .. B_3
    for pc in hx_a_pieces(range(9)):
        lines.extend(pc)
    lines += [i for g in hx_adv_a_stage() for i in g]
    for jb in range(HX_BAHEAD):
        for pc in hx_b_pieces():
            lines.extend(pc)
        lines += [i for g in hx_adv_b_stage(jb % 3) for i in g]
    for i in range(G.NACC):
        lines.append("v_accvgpr_write_b32 a%d, 0" % i)
    lines += ["s_waitcnt vmcnt(%d)" % ((HX_BAHEAD - 1) * G.PB), "s_barrier"]
    lines += ["v_add_u32 v124, s87, %[la00]", "v_add_u32 v125, s88, %[lb0]"] + hx_reads(0)
    # main loop: channel blocks 0 .. ncb - 3 (everything staged), then ncb - 2 (the last A item, two more B items), then ncb - 1
    lines += ["s_cmp_eq_u32 s84, 0", "s_cbranch_scc1 L_w4_tail%=", "L_w4_loop%=:"]
    for tap in range(3):
        lines += hx_k_step("main", tap, waits)
    lines += ["s_sub_u32 s84, s84, 1", "s_cmp_lg_u32 s84, 0", "s_cbranch_scc1 L_w4_loop%=", "L_w4_tail%=:"]
    for tap in range(3):
        lines += hx_k_step("tail1", tap, waits)
    lines += hx_k_step("tail2", 0, waits) + hx_k_step("tail2", 1, waits) + hx_k_step("tail2", 2, waits, next_reads=False)
    lines += ["s_nop 15", "s_nop 15"]
    return lines


def check_scc(lines):
    """every consumer of SCC must see the producer it was written for (the generator moves scalar instructions around)"""
    last = None
    for ln in lines:
        op = ln.split()[0]
        args_ = [a.strip(",") for a in ln.split()[1:]]
        if op.endswith(":") or op == "s_branch":
            last = None
            continue
        if op == "s_cselect_b32":
            assert last and (last[0] == "s_cmp_eq_u32" or (last[0] == "s_cmp_lt_u32" and last[1][0] == args_[1])), (last, ln)
        elif op == "s_addc_u32":
            assert last and last[0] == "s_add_u32" and int(last[1][0][1:]) + 1 == int(args_[0][1:]), (last, ln)
        elif op.startswith("s_cbranch_scc"):
            assert last and last[0].startswith("s_cmp"), (last, ln)
        if op in ("s_add_u32", "s_sub_u32", "s_addc_u32") or op.startswith("s_cmp"):
            last = (op, args_)
    # M0: each LDS-DMA piece is preceded by exactly one M0 write with at least one instruction between them
    m0_age = None
    for ln in lines:
        if ln.startswith("s_add_u32 m0"):
            assert m0_age is None, "M0 written twice before its piece"
            m0_age = 0
        elif ln.startswith("global_load_lds") or ln.startswith("buffer_load_dwordx4"):
            assert m0_age is not None and m0_age >= 1, "piece without a settled M0: %s" % ln
            m0_age = None
        elif m0_age is not None:
            m0_age += 1


clob = ["memory", "scc", "vcc"] + ["s%d" % i for i in range(79, 100)] + ["v%d" % i for i in range(113, 256)] + ["a%d" % i for i in range(256)]
VARIANTS = [Cfg()]
if args.experiments:
    VARIANTS += [Cfg(no_dma=True), Cfg(no_ds=True), Cfg(no_barrier=True), Cfg(no_dma=True, no_ds=True), Cfg(no_mfma=True),
                 Cfg(mfma32=True, dma_first=4), Cfg(mfma32=True, no_dma=True), Cfg(mfma32=True, no_dma=True, no_ds=True), Cfg(mfma32=True, adv=0, dma_first=4)]
with open(args.out, "w") as f:
    f.write("// GENERATED by scripts/gen_w4_loop.py%s -- do not edit.\n" % (" --experiments" if args.experiments else ""))
    for vi, c in enumerate(VARIANTS):
        cfg = c
        lines = build()
        check_scc(lines)
        f.write("// variant %d: %s (%d instructions)\n" % (vi, c.desc, len(lines)))
        f.write("#define W4_LOOP_ASM_%d \\\n" % vi)
        for ln in lines:
            f.write('  "%s\\n\\t" \\\n' % ln)
        f.write('  ""\n')
    cfg = Cfg(swap=True)
    lines = build()
    check_scc(lines)
    f.write("// shipped schedule, transposed accumulator tiles (gemm_nt_w4_kernel<0, true>): %d instructions\n" % len(lines))
    f.write("#define W4S_LOOP_ASM \\\n")
    for ln in lines:
        f.write('  "%s\\n\\t" \\\n' % ln)
    f.write('  ""\n')
    cfg = Cfg(conv=True)
    lines = build()
    check_scc(lines)
    f.write("// conv mode (k = 3, stride 1): %d instructions\n" % len(lines))
    f.write("#define W4C_LOOP_ASM \\\n")
    for ln in lines:
        f.write('  "%s\\n\\t" \\\n' % ln)
    f.write('  ""\n')
    cfg = Cfg(conv=True, swap=True)
    lines = build()
    check_scc(lines)
    f.write("// conv mode, transposed accumulator tiles (gemm_nt_w4c_kernel<true>): %d instructions\n" % len(lines))
    f.write("#define W4CS_LOOP_ASM \\\n")
    for ln in lines:
        f.write('  "%s\\n\\t" \\\n' % ln)
    f.write('  ""\n')
    # 256 x 128 tiles (gemm_nt_w4h_kernel): 32 MFMAs per half-step -> one fragment read after every 2nd, one piece after every 3rd
    # (transposed accumulator tiles: gemm_nt_w4h_kernel's epilogue stores rows straight out of the AGPRs)
    for name, c in (("W4H_LOOP_ASM", Cfg(half=True, swap=True, ds_every=2, dma_every=3)), ("W4HC_LOOP_ASM", Cfg(half=True, swap=True, conv=True, ds_every=2, dma_every=3)),
                    ("W4HT_LOOP_ASM", Cfg(half=True, swap=True, conv=True, tapil=True, ds_every=2, dma_every=3))):
        cfg = c
        lines = build()
        check_scc(lines)
        f.write("// %s: %d instructions\n" % (c.desc, len(lines)))
        f.write("#define %s \\\n" % name)
        for ln in lines:
            f.write('  "%s\\n\\t" \\\n' % ln)
        f.write('  ""\n')
    cfg = Cfg(half=True, swap=True, conv=True, ds_every=2, dma_every=3)
    lines = build_halo()
    check_scc(lines)
    f.write("// W4HX_LOOP_ASM: k = 3 convolutions, a channel block staged once for its three taps (%d instructions)\n" % len(lines))
    f.write("#define W4HX_LOOP_ASM \\\n")
    for ln in lines:
        f.write('  "%s\\n\\t" \\\n' % ln)
    f.write('  ""\n')
    # ---- split-K exchange of gemm_nt_w4h_kernel (accumulators a[0:127], 32 groups of 4; a lane's group g of split q lives at
    # base + ((q*32 + g)*256 + tid)*16 bytes: 4 KB per wave instruction, consecutive groups 4096 bytes apart, consecutive splits too).
    # PUBLISH: the 32 groups straight out of the AGPRs (write-through), ONE wait.  GATHER (the last arriver): a[0:127] = 0, then
    # split after split is added in split order; a split comes in as two halves of 16 loads into v[128:191] / v[192:255], the next
    # half in flight while the current one is added (the first version waited for 8 loads at a time: 16 dependent ~2.5 us round
    # trips = the 40 us that made the in-launch split lose).
    def bump(lo):
        return ["s_add_u32 s%d, s%d, 0x1000" % (lo, lo), "s_addc_u32 s%d, s%d, 0" % (lo + 1, lo + 1)]
    pub = ["s_mov_b64 s[80:81], %[base]"]
    for g in range(32):
        pub.append("global_store_dwordx4 %%[off], a[%d:%d], s[80:81] sc1" % (4 * g, 4 * g + 3))
        pub += bump(80)
    pub.append("s_waitcnt vmcnt(0)")
    # CONFIRM (W4H_CONFIRM_ASM, run when GemmParams::ksplit carries DRN_XCHG_CONFIRM): the stores' completion is not their visibility
    # to the other XCDs (skinny_group_kernel, qdense.hip, says why): one returning agent-scope OR-with-zero per 64-byte request
    # (lanes 0, 4, 8, ...) of each of the 32 stores -- a read-modify-write of the same address is performed behind the store --
    # all in flight, one wait, before the workgroup counts itself in.
    conf = ["s_mov_b64 s[80:81], %[base]", "v_mov_b32 v113, 0", "s_mov_b32 s82, 0x11111111", "s_mov_b32 s83, 0x11111111",
            "s_mov_b64 exec, s[82:83]"]
    for g in range(32):
        conf.append("global_atomic_or v%d, %%[off], v113, s[80:81] sc0 sc1" % (128 + g))
        conf += bump(80)
    conf += ["s_mov_b64 exec, -1", "s_waitcnt vmcnt(0)"]
    # READ-BACK (W4H_READBACK_ASM, DRN_XCHG_READBACK): the cheaper confirmation -- an sc1 LOAD of the first dword of every 64-byte
    # request instead of the read-modify-write (no second write, no atomic unit)
    rdb = ["s_mov_b64 s[80:81], %[base]", "s_mov_b32 s82, 0x11111111", "s_mov_b32 s83, 0x11111111", "s_mov_b64 exec, s[82:83]"]
    for g in range(32):
        rdb.append("global_load_dword v%d, %%[off], s[80:81] sc1" % (128 + g))
        rdb += bump(80)
    rdb += ["s_mov_b64 exec, -1", "s_waitcnt vmcnt(0)"]
    def loads(v0):
        out = []
        for g in range(16):
            out.append("global_load_dwordx4 v[%d:%d], %%[off], s[82:83] sc1" % (v0 + 4 * g, v0 + 4 * g + 3))
            out += bump(82)
        return out
    def add(v0, a0):
        out = []
        for b in range(0, 64, 8):                         # 8 at a time through v[113:120]: read, add, write back
            out += ["v_accvgpr_read_b32 v%d, a%d" % (113 + i, a0 + b + i) for i in range(8)]
            out += ["v_add_f32 v%d, v%d, v%d" % (113 + i, 113 + i, v0 + b + i) for i in range(8)]
            out += ["v_accvgpr_write_b32 a%d, v%d" % (a0 + b + i, 113 + i) for i in range(8)]
        return out
    gat = ["s_mov_b64 s[82:83], %[base]", "s_mov_b32 s84, %[ks]", "v_mov_b32 v113, 0"]
    gat += ["v_accvgpr_write_b32 a%d, v113" % r for r in range(128)]
    gat += loads(128)
    gat.append("1:")
    gat += loads(192)
    gat.append("s_waitcnt vmcnt(16)")
    gat += add(128, 0)
    gat += ["s_sub_u32 s84, s84, 1", "s_cmp_eq_u32 s84, 0", "s_cbranch_scc1 2f"]
    gat += loads(128)
    gat.append("s_waitcnt vmcnt(16)")
    gat += add(192, 64)
    gat.append("s_branch 1b")
    gat.append("2:")
    gat.append("s_waitcnt vmcnt(0)")
    gat += add(192, 64)
    gat.append("s_nop 4")
    for name, lines in (("W4H_PUBLISH_ASM", pub), ("W4H_CONFIRM_ASM", conf), ("W4H_READBACK_ASM", rdb), ("W4H_GATHER_ASM", gat)):
        f.write("// %s: %d instructions\n" % (name, len(lines)))
        f.write("#define %s \\\n" % name)
        for ln in lines:
            f.write('  "%s\\n\\t" \\\n' % ln)
        f.write('  ""\n')
    f.write("#define W4H_XCHG_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in ["memory", "scc", "s80", "s81", "s82", "s83", "s84"] + ["v%d" % i for i in range(113, 121)] + ["v%d" % i for i in range(128, 256)]))
    f.write("#define W4_VARIANTS %d\n" % len(VARIANTS))
    f.write("#define W4_LOOP_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob))
    f.write("#define W4H_LOOP_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob + ["s100"]))
    f.write("#define W4HX_LOOP_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob + ["s100", "s101"]))
    f.write("#define W4HT_LOOP_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob + ["s100", "s101"] + ["v%d" % i for i in range(97, 113)]))
print("wrote %s: %d variant(s)" % (args.out, len(VARIANTS)))
