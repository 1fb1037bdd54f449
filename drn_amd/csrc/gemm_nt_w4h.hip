// The hand-scheduled 4-wave loop of gemm_nt_w4.hip on 256 x 128 output tiles, for the bf16 launches whose N (512 for every FPN /
// head convolution of DRN: model/FPN.py:54-69, model/fcos.py:58-62) gives too few 256 x 256 tiles to fill 256 CUs -- the three
// pyramid levels of a grouped launch make 112 -- and which on 128 x 128 tiles of eight 64 x 32 waves are bound by the LDS, not by
// the matrix pipe: 96 KB of fragment reads + 32 KB of LDS-DMA writes per 128 x 128 x 64 MACs = 1024 LDS cycles per K-step and
// workgroup against 512 of MFMA (measured: 35 / 55 us for the 24- / 48-K-step launches = two workgroups per CU x K-steps x 1024
// cycles + ~8 us).  Here a workgroup is 4 waves = ONE per SIMD, a wave owns 128 x 64 outputs (8 x 4 MFMA tiles, 128 accumulator
// AGPRs): 24 KB of fragment reads per 128 x 64 x 64 MACs, 1152 LDS cycles per K-step and CU against 1024 of MFMA, and 224 tiles for
// the same launch.  Ring: three K-steps as [A 32 KB | B 16 KB] pairs (144 KB); schedule, staging and waits as in the 256 x 256 loop
// (scripts/gen_w4_loop.py, `Geo`).
//
// Two kernels: plain products (1x1 convolutions / Linear, forward or data gradient: taps = 1, stride 1) and k = 3 / stride 1 / pad 1
// convolutions (forward: mode 0, data gradient: mode 1) -- the A side of the latter is gemm_nt_w4c_kernel's (buffer descriptor, tap
// shifts, zero rows at the sequence edges).  Same MFMA instruction and K order per output element, same `+ bias`, `* gate` and rounding:
// the OUTPUT is bit-identical to the 128 x 128 kernel these launches ran on.  Round 6: the MFMAs take the weight fragment first, so the
// accumulator tiles come out transposed and the epilogue needs no transposition (see "epilogue" below); the per-slab BatchNorm
// statistics are the same (sum, M2) pairs summed in this kernel's own fixed order (they agree with the 128 x 128 kernel's to fp32
// rounding, not bit for bit).
#ifdef DRN_NT_PHASES
#define DRN_NT_PHASES_NAME drn_debug_nt_phases_w4h      // (this translation unit's own stamp table: scripts/experiments/w4_phases.py)
#endif
#include "gemm_nt_kernel.h"
#include "w4_epilogue.h"
#include "gemm_nt_w4_loop.inc"
#ifdef DRN_NT_PHASES
extern "C" int drn_debug_epi_cyc(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nt_epi_cyc), (size_t)n * 8); }
#endif

// ---- split-K inside the launch (conv0's forward: 64 tiles x 204 K-steps): workgroup row y owns K-steps [k_lo, k_hi), publishes its
// 128 accumulator registers per lane as 32 write-through 16-byte pieces straight out of the AGPRs (lane-contiguous: 4 KB per wave
// instruction), takes a ticket, and the split that arrives LAST at a tile re-reads all partials -- its own included, in split order,
// so the sum does not depend on who was last -- into the AGPRs and runs the normal epilogue.  Same protocol as
// conv_gemm_nt_kernel's split (gemm_nt_kernel.h); the two statements are W4H_PUBLISH_ASM / W4H_GATHER_ASM (gen_w4_loop.py).
#define W4H_TAPIL 0x10000      // flag in GemmParams::ksplit (this kernel only)
#define W4H_HALO 0x100000      // flag in GemmParams::ksplit: the K loop walks (channel block, tap) and stages a channel block ONCE for its three
                               // taps (W4HX_LOOP_ASM, gen_w4_loop.py: rows -1 .. 256 of the tile as four 66-row blocks, a third of the A traffic)
constexpr int W4H_LDS = 3 * (32768 + 16384), W4HX_A_RING = 2 * 36864, W4HX_LDS = W4HX_A_RING + 5 * 16384;
template <bool CONV>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4h_kernel(const GemmParams P_arg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  NtHeader P;
  GemmProb pr;
  int g, tm, tn;
  NT_PHASE(0);
  nt_fetch(P_arg, P, pr, g, blockIdx.x);
  nt_locate<256>(P, pr, tm, tn);
  const int m0 = tm * 256, n0 = tn * 128;
  const int wr = w >> 1, wc = w & 1;

  // ---- staging.  A: piece p = 8*w + i = rows 8*p .. 8*p + 7 of the tile's 256; B: piece p = 4*w + i of its 128 rows; lane l carries
  // the 16-byte chunk that belongs at position l & 7 of row 8*p + (l >> 3), i.e. chunk (l & 7) ^ ((row >> 1) & 7)
  unsigned voa[8], vob[4], mask_first = 0u, mask_last = 0u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned row = (unsigned)((w * 8 + i) * 8 + (l >> 3));
    const unsigned chunk = (unsigned)((l & 7) ^ ((((i & 1) << 2) + (l >> 4)) & 7)) * 16u;
    voa[i] = row * (unsigned)pr.lda * 2u + chunk;
    if constexpr (CONV) {
      const int t = (m0 + (int)row) % pr.Lout;
      mask_first |= (t == 0 ? 1u : 0u) << i;
      mask_last |= (t == pr.Lout - 1 ? 1u : 0u) << i;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned row = (unsigned)((w * 4 + i) * 8 + (l >> 3));
    const unsigned chunk = (unsigned)((l & 7) ^ ((((i & 1) << 2) + (l >> 4)) & 7)) * 16u;
    vob[i] = row * (unsigned)pr.ldb * 2u + chunk;
  }
  // ---- fragments of k-slice ks: lane l reads chunk 4*ks + (l >> 4) of row l & 15 of a 16-row block
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned lrow = lds0 + (unsigned)(l & 15) * 128u, swz = (unsigned)(l >> 1) & 7u;
  const unsigned pc0 = (((unsigned)(l >> 4)) ^ swz) * 16u, pc1 = ((4u + (unsigned)(l >> 4)) ^ swz) * 16u;
  const unsigned la0 = lrow + (unsigned)wr * 16384u + pc0, la1 = lrow + (unsigned)wr * 16384u + pc1;
  const unsigned lb0 = lrow + (unsigned)wc * 8192u + pc0, lb1 = lrow + (unsigned)wc * 8192u + pc1;
  const unsigned lw = lds0 + (unsigned)w * 8192u, lwb = lds0 + (unsigned)w * 4096u;
  // split K (ksplit > 1): workgroup row y owns K-steps [k_lo, k_hi).  W4H_TAPIL (conv launches with a wide input): the K loop walks
  // (channel block, tap) instead of (tap, channel block) -- W4HT_LOOP_ASM -- and a split starts at a whole channel block
  const bool halo = CONV && (P.ksplit & W4H_HALO) != 0;
  const bool tapil = CONV && (P.ksplit & W4H_TAPIL) != 0;
  P.ksplit &= W4H_TAPIL - 1;       // (the exchange flags DRN_XCHG_* were taken out by nt_fetch)
  const int ksteps_all = pr.K / 64;
  int kt_per = (ksteps_all + P.ksplit - 1) / P.ksplit;
  if (tapil || halo) kt_per = (kt_per + 2) / 3 * 3;
  const int k_lo = (int)blockIdx.y * kt_per, k_hi = min(ksteps_all, k_lo + kt_per);
  const char* sb = (const char*)pr.B + (long)n0 * pr.ldb * 2 + (long)k_lo * 128;
  const int trips = (k_hi - k_lo) - 2;
  NT_PHASE(1);

  if constexpr (CONV) {
    const unsigned long long abase = (unsigned long long)(uintptr_t)pr.A + (unsigned long long)((long)(m0 - 1) * pr.lda * 2);
    const unsigned d0 = (unsigned)abase, d1 = (unsigned)(abase >> 32) & 0xffffu, d2 = 0x7fffffffu, d3 = 0x00020000u;
    const unsigned lda2 = (unsigned)pr.lda * 2u;
    // forward (mode 0): tap k reads row m + k - 1 -> shifts 0, 1, 2 rows, zeros for the first row at tap 0 and the last at tap 2;
    // data gradient (mode 1): tap k reads row m + 1 - k -> 2, 1, 0 and the masks change places
    const bool fwd = pr.mode == 0;
    const unsigned sh0 = fwd ? 0u : 2u * lda2, sh1 = lda2, sh2 = fwd ? 2u * lda2 : 0u;
    const unsigned ma = fwd ? mask_first : mask_last, mc = fwd ? mask_last : mask_first;
    const int per = pr.Cin / 64;                      // K-steps per tap
    if (halo) {
      // a channel block staged once for its three taps.  Wave w stages block w = positions 0 .. 71 = tile rows 64 w - 1 .. 64 w + 70 as nine
      // pieces; positions 66 .. 71 are never read, the halo positions 0 / 65 come in as zeros (lane offset out of range) when their row
      // belongs to another sequence.  Offsets relative to the descriptor base = tile row -1.
      const bool seq_first = (m0 + 64 * w) % pr.Lout == 0, seq_last = (m0 + 64 * w + 64) % pr.Lout == 0;
      unsigned vh[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int pos = 8 * k + (l >> 3);
        const unsigned chunk = (unsigned)((l & 7) ^ ((((k & 1) << 2) + (l >> 4)) & 7)) * 16u;
        const bool ok = pos <= 65 && !(pos == 0 && seq_first) && !(pos == 65 && seq_last);
        vh[k] = ok ? (unsigned)(64 * w + pos) * lda2 + chunk : 0x80000000u;
      }
      // fragments: tap t of output row r reads position (r % 64) + t of block r / 64; the swizzle follows the position
      const unsigned r16 = (unsigned)(l & 15);
      unsigned lat[3][2];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          lat[t][ks] = lds0 + (unsigned)wr * (2u * 9216u) + (r16 + t) * 128u + (((4u * ks + (unsigned)(l >> 4)) ^ (((r16 + t) >> 1) & 7u)) * 16u);
      // forward: weight tap k multiplies row m + k - 1 = position offset k; data gradient: row m + 1 - k = position offset 2 - k
      const int t0 = fwd ? 0 : 2, t2 = fwd ? 2 : 0;
      const unsigned lbx0 = lb0 + (unsigned)W4HX_A_RING, lbx1 = lb1 + (unsigned)W4HX_A_RING;           // the B ring sits behind the A slots
      const unsigned lwa = lds0 + (unsigned)w * 9216u, lwbx = lds0 + (unsigned)W4HX_A_RING + (unsigned)w * 4096u;
      const int c0 = (k_lo / 3) * 128, dstep = pr.Cin * 2, dwrap = 128 - 2 * dstep;
      const char* sbt = (const char*)pr.B + (long)n0 * pr.ldb * 2 + c0;
      const int ncb = (k_hi - k_lo) / 3;
      asm volatile(W4HX_LOOP_ASM
                   :
                   : [sb] "s"(sbt), [cnt] "s"(ncb - 2), [lw] "s"(lwa), [lwb] "s"(lwbx), [d0] "s"(d0), [d1] "s"(d1), [d2] "s"(d2), [d3] "s"(d3),
                     [c0] "s"(c0), [dstep] "s"(dstep), [dwrap] "s"(dwrap),
                     [voa0] "v"(vh[0]), [voa1] "v"(vh[1]), [voa2] "v"(vh[2]), [voa3] "v"(vh[3]), [voa4] "v"(vh[4]), [voa5] "v"(vh[5]),
                     [voa6] "v"(vh[6]), [voa7] "v"(vh[7]), [voa8] "v"(vh[8]),
                     [vob0] "v"(vob[0]), [vob1] "v"(vob[1]), [vob2] "v"(vob[2]), [vob3] "v"(vob[3]),
                     [la00] "v"(lat[t0][0]), [la01] "v"(lat[t0][1]), [la10] "v"(lat[1][0]), [la11] "v"(lat[1][1]),
                     [la20] "v"(lat[t2][0]), [la21] "v"(lat[t2][1]), [lb0] "v"(lbx0), [lb1] "v"(lbx1)
                   : W4HX_LOOP_CLOBBERS);
    } else if (tapil) {
      const int c0 = (k_lo / 3) * 128, dstep = pr.Cin * 2, dwrap = 128 - 2 * dstep;
      const char* sbt = (const char*)pr.B + (long)n0 * pr.ldb * 2 + c0;
      asm volatile(W4HT_LOOP_ASM
                   :
                   : [sb] "s"(sbt), [cnt] "s"(trips), [lw] "s"(lw), [lwb] "s"(lwb), [d0] "s"(d0), [d1] "s"(d1), [d2] "s"(d2), [d3] "s"(d3),
                     [c0] "s"(c0), [dstep] "s"(dstep), [dwrap] "s"(dwrap), [sh0] "s"(sh0), [sh1] "s"(sh1), [sh2] "s"(sh2),
                     [voa0] "v"(voa[0]), [voa1] "v"(voa[1]), [voa2] "v"(voa[2]), [voa3] "v"(voa[3]), [voa4] "v"(voa[4]), [voa5] "v"(voa[5]),
                     [voa6] "v"(voa[6]), [voa7] "v"(voa[7]), [ma] "v"(ma), [mc] "v"(mc),
                     [vob0] "v"(vob[0]), [vob1] "v"(vob[1]), [vob2] "v"(vob[2]), [vob3] "v"(vob[3]),
                     [la0] "v"(la0), [la1] "v"(la1), [lb0] "v"(lb0), [lb1] "v"(lb1)
                   : W4HT_LOOP_CLOBBERS);
    } else {
    const int tap = k_lo / per, c0 = (k_lo - tap * per) * 128, left = per - (k_lo - tap * per);
    asm volatile(W4HC_LOOP_ASM
                 :
                 : [sb] "s"(sb), [cnt] "s"(trips), [lw] "s"(lw), [lwb] "s"(lwb), [d0] "s"(d0), [d1] "s"(d1), [d2] "s"(d2), [d3] "s"(d3),
                   [c0] "s"(c0), [left] "s"(left), [per] "s"(per), [tap] "s"(tap), [sh0] "s"(sh0), [sh1] "s"(sh1), [sh2] "s"(sh2),
                   [voa0] "v"(voa[0]), [voa1] "v"(voa[1]), [voa2] "v"(voa[2]), [voa3] "v"(voa[3]), [voa4] "v"(voa[4]), [voa5] "v"(voa[5]),
                   [voa6] "v"(voa[6]), [voa7] "v"(voa[7]), [ma] "v"(ma), [mc] "v"(mc),
                   [vob0] "v"(vob[0]), [vob1] "v"(vob[1]), [vob2] "v"(vob[2]), [vob3] "v"(vob[3]),
                   [la0] "v"(la0), [la1] "v"(la1), [lb0] "v"(lb0), [lb1] "v"(lb1)
                 : W4H_LOOP_CLOBBERS);
    }
  } else {
    const char* sa = (const char*)pr.A + (long)m0 * pr.lda * 2 + (long)k_lo * 128;
    asm volatile(W4H_LOOP_ASM
                 :
                 : [sa] "s"(sa), [sb] "s"(sb), [cnt] "s"(trips), [lw] "s"(lw), [lwb] "s"(lwb),
                   [voa0] "v"(voa[0]), [voa1] "v"(voa[1]), [voa2] "v"(voa[2]), [voa3] "v"(voa[3]), [voa4] "v"(voa[4]), [voa5] "v"(voa[5]),
                   [voa6] "v"(voa[6]), [voa7] "v"(voa[7]),
                   [vob0] "v"(vob[0]), [vob1] "v"(vob[1]), [vob2] "v"(vob[2]), [vob3] "v"(vob[3]),
                   [la0] "v"(la0), [la1] "v"(la1), [lb0] "v"(lb0), [lb1] "v"(lb1)
                 : W4H_LOOP_CLOBBERS);
  }
  __syncthreads();        // everybody is past its last fragment read: the LDS becomes the epilogue's staging patches
  nt_globalize(pr);       // (after the loop statement: nothing of it is alive across the loop)
  nt_globalize(P);
  NT_PHASE(2);

  if (P.ksplit > 1) {
    const int tile_id = blockIdx.x, ks = P.ksplit;
    const char* tile_base = (const char*)P.ws + (long)tile_id * ks * (32 * 256 * 16);      // (wave-uniform: scalar registers)
    const char* slab = tile_base + (long)blockIdx.y * (32 * 256 * 16);
    const unsigned lane_off = (unsigned)tid * 16u;
    asm volatile(W4H_PUBLISH_ASM : : [base] "s"(slab), [off] "v"(lane_off) : W4H_XCHG_CLOBBERS);
    if (P_arg.ksplit & DRN_XCHG_CONFIRM) asm volatile(W4H_CONFIRM_ASM : : [base] "s"(slab), [off] "v"(lane_off) : W4H_XCHG_CLOBBERS);
    if (P_arg.ksplit & DRN_XCHG_READBACK) asm volatile(W4H_READBACK_ASM : : [base] "s"(slab), [off] "v"(lane_off) : W4H_XCHG_CLOBBERS);
    __syncthreads();
    int& s_last = *(int*)smem;
    if (tid == 0) {
      const int prev = __hip_atomic_fetch_add(P.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = prev == ks - 1;
      if (prev == ks - 1) __hip_atomic_store(P.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
    }
    __syncthreads();
    if (!s_last) return;
    asm volatile(W4H_GATHER_ASM : : [base] "s"(tile_base), [ks] "s"(ks), [off] "v"(lane_off) : W4H_XCHG_CLOBBERS);
    __syncthreads();
  }

  NT_PHASE(3);
  w4h_epilogue<4>(pr, smem + w * (2 * 32 * (4 * 32 + 16)), m0 + wr * 128, n0 + wc * 64, tm * 2 + wr);
#ifdef DRN_NT_PHASES
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the stamp that follows = the stores have left the wave)
#endif
  NT_PHASE(4);
}

// Can every problem of this launch run on gemm_nt_w4h_kernel<conv>?  All problems of a launch must be of one kind.
bool drn_nt_w4h_eligible(const DrnGemmDesc* d, int ngroups, int dtype, bool* conv_out) {
  if (dtype != DRN_BF16) return false;
  bool conv = false;
  for (int g = 0; g < ngroups; ++g) {
    const DrnGemmDesc& s = d[g];
    const bool plain = s.taps == 1 && s.stride == 1 && s.pad == 0 && s.Lout == s.Lsrc;          // (mode 0 / 1 coincide: one tap, no shift)
    const bool k3 = s.taps == 3 && s.stride == 1 && s.pad == 1 && s.Lout == s.Lsrc && (s.mode == 0 || s.mode == 1) && s.M % s.Lout == 0;
    if (!plain && !k3) return false;
    if (g == 0) conv = k3;
    else if (conv != k3) return false;
    if (s.M % 256 || s.N % 128 || s.Cin % 64 || s.taps * s.Cin < 128 || s.out_f32 || s.accumulate) return false;
    if ((long)s.lda >= (1L << 22) || (long)s.ldb >= (1L << 22)) return false;            // 32-bit staging offsets
    if (s.ldc % 8 || ((uintptr_t)s.C & 15) || (s.C2 && (s.ldc2 % 8 || ((uintptr_t)s.C2 & 15)))) return false;
    if (s.stats && (s.gate || s.C2)) return false;       // (statistics of gated outputs: no caller; nt_epilogue orders them differently)
  }
  *conv_out = conv;
  return true;
}

int drn_nt_w4h_launch(const GemmParams& P, int total, bool conv, hipStream_t stream, int ksplit) {
  static bool attr_set = false;
  constexpr int LDS = W4H_LDS;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4h_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4h_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W4HX_LDS);
    attr_set = true;
  }
  if (conv && drn_tuning(DRN_TUNE_W4H_HALO) > 0) {
    // every sequence a multiple of 64 rows (a staging block never straddles two), whole channel blocks per split, at least two of them
    bool ok = true;
    for (int g = 0; g < P.ngroups; ++g) ok = ok && P.p[g].Lout % 64 == 0 && P.p[g].Lout == P.p[g].Lsrc && P.p[g].Cin % 64 == 0 && P.p[g].M % 256 == 0;
    const int ncb_all = P.p[0].Cin / 64, per = cdiv(ncb_all, ksplit);
    for (int g = 1; g < P.ngroups; ++g) ok = ok && P.p[g].Cin == P.p[0].Cin;
    ok = ok && per >= 2 && ncb_all - (ksplit - 1) * per >= 2;
    if (ok) {
      GemmParams Q = P;
      Q.ksplit = P.ksplit | W4H_HALO;
      gemm_nt_w4h_kernel<true><<<dim3(total, ksplit), 256, W4HX_LDS, stream>>>(Q);
      return drn_launch_status("drn_gemm_nt");
    }
  }
  if (conv && ksplit > 1 && drn_tuning(DRN_TUNE_W4H_TAPIL) > 0 && P.p[0].Cin >= drn_tuning(DRN_TUNE_W4H_TAPIL)) {
    // split conv launches over a wide input (conv0's forward: 4352 channels): taps interleaved, splits of whole channel blocks --
    // if every split still gets its share
    const int ksteps = P.p[0].K / 64, per = (cdiv(ksteps, ksplit) + 2) / 3 * 3;
    if ((ksplit - 1) * per < ksteps) {
      GemmParams Q = P;
      Q.ksplit = P.ksplit | W4H_TAPIL;        // (P.ksplit = ksplit + the DRN_XCHG_CONFIRM flag)
      gemm_nt_w4h_kernel<true><<<dim3(total, ksplit), 256, LDS, stream>>>(Q);
      return drn_launch_status("drn_gemm_nt");
    }
  }
  if (conv) gemm_nt_w4h_kernel<true><<<dim3(total, ksplit), 256, LDS, stream>>>(P);
  else gemm_nt_w4h_kernel<false><<<dim3(total, ksplit), 256, LDS, stream>>>(P);
  return drn_launch_status("drn_gemm_nt");
}
