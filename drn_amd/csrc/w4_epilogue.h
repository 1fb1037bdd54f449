// Epilogue of the 4-wave hand-scheduled GEMM kernels whose loops run the MFMAs with the operands exchanged (gemm_nt_w4h_kernel,
// gemm_nt_w4c_kernel<true>; scripts/gen_w4_loop.py `swap`): transposed accumulator tiles -> bf16 rows + per-slab BatchNorm statistics.
#pragma once
#include "gemm_nt_kernel.h"

template <int R>
__device__ __forceinline__ float w4h_acc_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(R));
  return v;
}
// ---- epilogue.  The loops run their MFMAs with the operands exchanged (gen_w4_loop.py, `swap`): accumulator tile (mi, ni) of the wave's
// 8 x NI grid = a[(mi*NI + ni)*4 .. +3] holds, in lane l, C[16 mi + (l & 15)][16 ni + 4 (l >> 4) + r], r = 0..3 -- FOUR CONSECUTIVE COLUMNS
// OF ONE ROW.  A lane's four values are 8 contiguous bytes of a bf16 row: they go into the wave's LDS patch with ONE ds_write_b64 and no
// transposition, and a column's 128 rows -- one BatchNorm slab, ALL of them inside this wave -- are summed per lane over mi and then
// over the 16 lanes of a DPP row: no cross-wave exchange, no workgroup barrier.  (Until round 6 the tiles came out column-per-lane and
// went through quad transposes -- 3 DPP moves, 2 byte permutes, 3 selects per tile -- and the statistics through LDS and three
// barriers: measured with -DDRN_NT_PHASES, scripts/experiments/w4_phases.py, the epilogue of a 256 x 128 tile took 7-9 us -- 4 us of
// it with the global stores compiled OUT -- behind K loops of 6-23 us; one wave per SIMD hides no latency.  Now 4.0-4.4 us.)
// Same products, same K order, same `+ bias`, `* gate`, rounding: the output bits are those of the 128 x 128 kernel.  The slab
// statistics are the same (sum, M2) pairs in a different -- fixed -- summation order.
// the lane number from the hardware (v_mbcnt), not from threadIdx: a value derived from threadIdx BEFORE the loop statement would have to
// survive it in v0..v112 -- the statement owns the rest -- and with the staging offsets there already, the compiler spilled to scratch
__device__ __forceinline__ int w4h_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
template <int CTRL>
__device__ __forceinline__ float w4h_dpp(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (l & 15), every lane gets it: xor 1, xor 2 (quad permutes), then the half-row and the row mirrored
// (the groups already agree, so a mirror brings in the other group's value)
__device__ __forceinline__ float w4h_row16_sum(float v) {
  v += w4h_dpp<0xB1>(v);       // quad_perm [1,0,3,2]
  v += w4h_dpp<0x4E>(v);       // quad_perm [2,3,0,1]
  v += w4h_dpp<0x141>(v);      // row_half_mirror
  v += w4h_dpp<0x140>(v);      // row_mirror
  return v;
}
__device__ __forceinline__ unsigned w4h_pack_bf16(const float a, const float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t p = {(bf16_t)a, (bf16_t)b};
  return __builtin_bit_cast(unsigned, p);
}
// accumulator tiles (MI, ni), ni = NI0 .. NI - 1, of grid row MI into the lane's place in the patch row(s).  DUAL: the value before the gate
// goes to the first patch (the pre-gate copy C2), the gated value to the second, from ONE read of the accumulators.
template <int NI, int NIT, int NOFF, int MI, int NI0, bool GATED, bool DUAL>
__device__ __forceinline__ void w4h_row_tiles(const float (&g)[NI][4], char* wrow, char* wrow2, const float (&bias4)[NI][4]) {
  if constexpr (NI0 < NI) {
    constexpr int R = (MI * NIT + NOFF + NI0) * 4;
    const float x0 = w4h_acc_read<R>(), x1 = w4h_acc_read<R + 1>(), x2 = w4h_acc_read<R + 2>(), x3 = w4h_acc_read<R + 3>();
    float v0 = x0 + bias4[NI0][0], v1 = x1 + bias4[NI0][1], v2 = x2 + bias4[NI0][2], v3 = x3 + bias4[NI0][3];
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    if constexpr (DUAL) *(u32x2_t*)(wrow + NI0 * 32) = (u32x2_t){w4h_pack_bf16(v0, v1), w4h_pack_bf16(v2, v3)};
    if constexpr (GATED) { v0 *= g[NI0][0]; v1 *= g[NI0][1]; v2 *= g[NI0][2]; v3 *= g[NI0][3]; }
    *(u32x2_t*)((DUAL ? wrow2 : wrow) + NI0 * 32) = (u32x2_t){w4h_pack_bf16(v0, v1), w4h_pack_bf16(v2, v3)};
    w4h_row_tiles<NI, NIT, NOFF, MI, NI0 + 1, GATED, DUAL>(g, wrow, wrow2, bias4);
  }
}
struct W4hOut {
  bf16_t* C;
  bf16_t* C2;          // pre-gate copy or NULL
  const float* gate;   // or NULL
  long ldc, ldc2, ldg;
  int Lout;
  bool gate_uniform;   // the wave's 128 rows belong to ONE sequence: the gate row is loaded once for the tile
};
// One 32-row chunk (grid rows 2 CH, 2 CH + 1) of the wave's tile: + bias, (* gate), round, and -- the lane holding 4 consecutive columns of
// a row -- ONE ds_write_b64 per accumulator tile into the wave's private LDS patch; then whole 16-byte pieces of whole rows go out
// (LPR lanes per row: every wave instruction writes 64 / LPR complete row segments), and the next chunk's arithmetic runs while
// they drain.  Measured (s_memtime of one wave, -DDRN_NT_PHASES, 128 x 64 per wave): the 8 bytes per lane stored straight from the
// registers -- 16 rows x 32 bytes per instruction, 16 partial lines to the memory pipeline -- 7200 clocks for the tile; these chunks
// 4300; ONE patch for the whole tile and one wait 5100 (all 224 workgroups then store at once: ~225 clocks per 1 KB instruction is
// the chip's write rate, not the wave's).  GATED / DUAL: prop_fc's forward writes the pre-gate value AND the gated one; as two passes
// over the accumulators (what the old epilogue did too) the step was 10 us slower than with the old layout, as one pass with two
// patches it is the faster one (scripts/experiments/ab_lib.sh).
template <int NI, int NIT, int NOFF, int CH, bool GATED, bool DUAL>
__device__ __forceinline__ void w4h_store_chunk(const W4hOut& O, char* wbuf, const int mrow0, const int ncol0, const float (&bias4)[NI][4],
                                                float (&g)[NI][4]) {
  constexpr int PITCH = NI * 32 + 16, LPR = NI * 2, RPI = 64 / LPR, PATCH = 32 * PITCH;
  const int l = w4h_lane(), rho = l & 15, q = l >> 4;
  char* wrow = wbuf + rho * PITCH + q * 8;
#define W4H_ROW(MI2) do { \
    if (GATED && !O.gate_uniform) { \
      const float* gp = O.gate + (long)((mrow0 + CH * 32 + MI2 * 16 + rho) / O.Lout) * O.ldg + ncol0 + 4 * q; \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) _Pragma("unroll") for (int r = 0; r < 4; ++r) g[ni][r] = gp[ni * 16 + r]; \
    } \
    w4h_row_tiles<NI, NIT, NOFF, 2 * CH + MI2, 0, GATED, DUAL>(g, wrow + MI2 * 16 * PITCH, wrow + PATCH + MI2 * 16 * PITCH, bias4); } while (0)
  W4H_ROW(0); W4H_ROW(1);
#undef W4H_ROW
  wave_lds_sync();
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int rl = it * RPI + l / LPR, cv = l % LPR;
    if constexpr (DUAL) {
      const uint4 pre = *(const uint4*)(wbuf + rl * PITCH + cv * 16);
      *(uint4*)(O.C2 + ((long)(mrow0 + CH * 32 + rl) * O.ldc2 + ncol0 + cv * 8)) = pre;
    }
    const uint4 raw = *(const uint4*)(wbuf + (DUAL ? PATCH : 0) + rl * PITCH + cv * 16);
    *(uint4*)(O.C + ((long)(mrow0 + CH * 32 + rl) * O.ldc + ncol0 + cv * 8)) = raw;
  }
  wave_lds_sync();
}
template <int NI, int NIT, int NOFF, bool GATED, bool DUAL>
__device__ __forceinline__ void w4h_store_tile(const W4hOut& O, char* wbuf, const int mrow0, const int ncol0, const float (&bias4)[NI][4]) {
  float g[NI][4];
  if (GATED && O.gate_uniform) {
    const int l = w4h_lane();
    const float* gp = O.gate + (long)(mrow0 / O.Lout) * O.ldg + ncol0 + 4 * (l >> 4);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) g[ni][r] = gp[ni * 16 + r];
  }
  w4h_store_chunk<NI, NIT, NOFF, 0, GATED, DUAL>(O, wbuf, mrow0, ncol0, bias4, g);
  w4h_store_chunk<NI, NIT, NOFF, 1, GATED, DUAL>(O, wbuf, mrow0, ncol0, bias4, g);
  w4h_store_chunk<NI, NIT, NOFF, 2, GATED, DUAL>(O, wbuf, mrow0, ncol0, bias4, g);
  w4h_store_chunk<NI, NIT, NOFF, 3, GATED, DUAL>(O, wbuf, mrow0, ncol0, bias4, g);
}
// first pass of the statistics: sum over mi of the raw accumulators per column slot.  (Passes of their own, each reading the AGPRs
// again: folded into the store pass the compiler postponed the additions and parked the values -- in AGPRs it believes free, i.e. in
// accumulator tiles not stored yet.  tests/test_kernel_resources_cpu.py now refuses any compiler-made AGPR access in these kernels.)
template <int NI, int NIT, int NOFF, int I, int I1>
__device__ __forceinline__ void w4h_sum_tiles(float (&cs)[NI][4]) {
  if constexpr (I < I1) {
    constexpr int ni = I % NI, R = ((I / NI) * NIT + NOFF + ni) * 4;
    cs[ni][0] += w4h_acc_read<R>(); cs[ni][1] += w4h_acc_read<R + 1>(); cs[ni][2] += w4h_acc_read<R + 2>(); cs[ni][3] += w4h_acc_read<R + 3>();
    w4h_sum_tiles<NI, NIT, NOFF, I + 1, I1>(cs);
  }
}
// second pass of the statistics: sum over mi of (x - mean)^2 per column slot
template <int NI, int NIT, int NOFF, int I, int I1>
__device__ __forceinline__ void w4h_sq_tiles(const float (&mean)[NI][4], float (&q)[NI][4]) {
  if constexpr (I < I1) {
    constexpr int ni = I % NI, R = ((I / NI) * NIT + NOFF + ni) * 4;
    const float d0 = w4h_acc_read<R>() - mean[ni][0], d1 = w4h_acc_read<R + 1>() - mean[ni][1];
    const float d2 = w4h_acc_read<R + 2>() - mean[ni][2], d3 = w4h_acc_read<R + 3>() - mean[ni][3];
    q[ni][0] = fmaf(d0, d0, q[ni][0]); q[ni][1] = fmaf(d1, d1, q[ni][1]); q[ni][2] = fmaf(d2, d2, q[ni][2]); q[ni][3] = fmaf(d3, d3, q[ni][3]);
    w4h_sq_tiles<NI, NIT, NOFF, I + 1, I1>(mean, q);
  }
}
// The whole epilogue of 128 rows x (16 NI) columns of a wave's transposed accumulator tiles: tile (mi, NOFF + ni) of a grid with NIT tiles per
// row, ni < NI.  gemm_nt_w4h_kernel: NI = NIT = 4; gemm_nt_w4c_kernel<true> (8 tiles per row) runs it twice with NI = 4, NOFF = 0 / 4 --
// as ONE 8-tile-wide pass the register allocator ran out and parked values in AGPRs (tests/test_kernel_resources_cpu.py).
// mrow0 / ncol0 = first row / column of the sub-tile; slab = the wave's 128-row slab.
template <int NI, int NIT = NI, int NOFF = 0>
__device__ __forceinline__ void w4h_epilogue(const GemmProb& pr, char* wbuf, const int mrow0, const int ncol0, const int slab) {
  const int l = w4h_lane();
  const int ncolq = ncol0 + 4 * (l >> 4);
  int Lout = pr.Lout;
  asm volatile("" : "+s"(Lout));      // (opaque: the reciprocal the staging code derived from Lout before the loop must not be kept alive for the gate rows)
  W4hOut O{(bf16_t*)pr.C, (bf16_t*)pr.C2, pr.gate, (long)pr.ldc, (long)pr.ldc2, (long)pr.ldg, Lout, mrow0 / Lout == (mrow0 + 127) / Lout};
  float bias4[NI][4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[ni][r] = pr.bias ? pr.bias[ncolq + ni * 16 + r] : 0.f;
  EPI_STAMP(0);
  if (!pr.stats) {          // (wave-uniform: descriptor fields; the eligibility rules keep statistics and gate / C2 apart)
    // (a pre-gate copy without a gate has no caller: it goes out as "gated by nothing" through the dual path's first patch only)
    if (O.C2 && O.gate) w4h_store_tile<NI, NIT, NOFF, true, true>(O, wbuf, mrow0, ncol0, bias4);
    else if (O.gate) w4h_store_tile<NI, NIT, NOFF, true, false>(O, wbuf, mrow0, ncol0, bias4);
    else {
      if (O.C2) {
        W4hOut P2 = O;
        P2.C = O.C2; P2.ldc = O.ldc2;
        w4h_store_tile<NI, NIT, NOFF, false, false>(P2, wbuf, mrow0, ncol0, bias4);
      }
      w4h_store_tile<NI, NIT, NOFF, false, false>(O, wbuf, mrow0, ncol0, bias4);
    }
    return;
  }
  // (sum, M2) of the slab's 128 rows per column, nn.BatchNorm1d's training statistics in the form drn_bn_train_apply merges
  // (include/drn_hip.h); the eligibility rule (M % 256 == 0) makes every slab whole.  Order: column sums first (a pass over the
  // AGPRs of its own), then the tile goes out chunk by chunk with a quarter of the centred-squares pass behind each chunk's stores --
  // the store pass is bound by the chip's write rate (every workgroup of the launch stores at this moment), the arithmetic hides in it.
  float mean[NI][4], q[NI][4];
  {
    float cs[NI][4];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[ni][r] = 0.f;
    w4h_sum_tiles<NI, NIT, NOFF, 0, 8 * NI>(cs);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cs[ni][r] = w4h_row16_sum(cs[ni][r]);
        mean[ni][r] = cs[ni][r] * 0.0078125f;
        q[ni][r] = 0.f;
      }
    if ((l & 15) == 0) {      // one lane per column quadruple writes; the four l >> 4 groups cover a tile's 16 columns
      float* st = pr.stats + (long)slab * 2 * pr.N + ncolq;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) *(float4*)(st + ni * 16) = make_float4(cs[ni][0], cs[ni][1], cs[ni][2], cs[ni][3]);
    }
  }
  EPI_STAMP(1);
  float gdummy[NI][4];
  w4h_store_chunk<NI, NIT, NOFF, 0, false, false>(O, wbuf, mrow0, ncol0, bias4, gdummy);
  w4h_sq_tiles<NI, NIT, NOFF, 0, 2 * NI>(mean, q);
  w4h_store_chunk<NI, NIT, NOFF, 1, false, false>(O, wbuf, mrow0, ncol0, bias4, gdummy);
  w4h_sq_tiles<NI, NIT, NOFF, 2 * NI, 4 * NI>(mean, q);
  w4h_store_chunk<NI, NIT, NOFF, 2, false, false>(O, wbuf, mrow0, ncol0, bias4, gdummy);
  w4h_sq_tiles<NI, NIT, NOFF, 4 * NI, 6 * NI>(mean, q);
  w4h_store_chunk<NI, NIT, NOFF, 3, false, false>(O, wbuf, mrow0, ncol0, bias4, gdummy);
  w4h_sq_tiles<NI, NIT, NOFF, 6 * NI, 8 * NI>(mean, q);
  EPI_STAMP(3);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) q[ni][r] = w4h_row16_sum(q[ni][r]);
  if ((l & 15) == 0) {
    float* st = pr.stats + (long)slab * 2 * pr.N + pr.N + ncolq;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) *(float4*)(st + ni * 16) = make_float4(q[ni][0], q[ni][1], q[ni][2], q[ni][3]);
  }
  EPI_STAMP(4);
#ifdef DRN_NT_PHASES
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  EPI_STAMP(5);
}


// fp32 destination (weight gradients of prop_fc: gemm_nt_w4_kernel<0, true>): the lane's four consecutive columns of a row are 16
// contiguous bytes -- stored straight from the registers, + bias; no accumulate here (the launcher keeps those on the other layout).
// NIT tiles per grid row; columns [16 I0, 16 I0 + COLS) of the wave's tile in passes of four tiles per grid row.
template <int NIT, int I, int I1>
__device__ __forceinline__ void w4s_f32_tiles(char* crow, const unsigned rstep, const float (&bias4)[NIT][4]) {
  if constexpr (I < I1) {
    constexpr int mi = I / NIT, ni = I % NIT, R = I * 4;
    const float4 v = make_float4(w4h_acc_read<R>() + bias4[ni][0], w4h_acc_read<R + 1>() + bias4[ni][1], w4h_acc_read<R + 2>() + bias4[ni][2],
                                 w4h_acc_read<R + 3>() + bias4[ni][3]);
    *(float4*)(crow + ((unsigned)mi * rstep + ni * 64u)) = v;
    w4s_f32_tiles<NIT, I + 1, I1>(crow, rstep, bias4);
  }
}
template <int NIT, int NOFF_UNUSED, int COLS_UNUSED>
__device__ __forceinline__ void w4s_store_f32(const GemmProb& pr, const int mrow0, const int ncol0) {
  const int l = w4h_lane();
  const int ncolq = ncol0 + 4 * (l >> 4);
  float bias4[NIT][4];
#pragma unroll
  for (int ni = 0; ni < NIT; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[ni][r] = pr.bias ? pr.bias[ncolq + ni * 16 + r] : 0.f;
  char* crow = (char*)pr.C + ((long)(mrow0 + (l & 15)) * pr.ldc + ncolq) * 4;
  w4s_f32_tiles<NIT, 0, 8 * NIT>(crow, 64u * (unsigned)pr.ldc, bias4);          // (16 rows x ldc x 4 bytes per grid row)
}
