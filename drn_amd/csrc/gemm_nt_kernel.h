// Grouped implicit-GEMM (NT) on CDNA4 MFMA: Conv1d fwd / dgrad and Linear fwd / dgrad.
//
// C[M][N] = A'(M x K) * B[N][K]^T, K = taps*Cin, A' = im2col view of a channels-last
// tensor (see include/drn_hip.h).  Replaces the cuDNN/cuBLAS calls behind
// nn.Conv1d (model/basic_blocks.py:9-18, model/fcos.py:33-69) and nn.Linear
// (model/main_model.py:33,59) of the reference.
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4
// MFMA 16x16 tiles), K-step = 128 bytes per row (64 bf16 / 32 f32).  Both operands
// are staged HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip) into a
// double-buffered, XOR-swizzled image: LDS is written lane-linear, so the swizzle is
// applied to the per-lane SOURCE address and again on the ds_read_b128 (rule 21 of
// the CDNA guide).  Out-of-range rows/taps/columns read a 16-byte zero page, so
// padding, stride-2 gradients and ragged edges need no branches in the MFMA loop.
//   bf16: v_mfma_f32_16x16x32_bf16 (8 bf16 = 16 B per lane per operand)
//   f32 : 4 x v_mfma_f32_16x16x4_f32 per 16-B fragment (exact fp32, parity mode);
//         the k-permutation this implies is applied identically to A and B.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "bn_merge.h"
#include "../../include/drn_hip.h"



struct GemmProb {
  const void* A;
  const void* B;
  void* C;
  void* C2;
  const float* bias;
  const float* gate;
  float* stats;
  int M, N, K;
  int Cin, taps, stride, pad, mode;
  int Lout, Lsrc;
  int lda, ldb, ldc, ldg, ldc2;
  int accumulate;
  int out_f32;      // C is fp32 [M][ldc] whatever T is (weight gradients computed as an NT product of transposed operands)
  int tiles_n, tile_start;
  float* sumsq;     // gemm_nt_w4_kernel, out_f32: [workgroups] sum of the squares of each workgroup's outputs (DrnGemmDesc::sumsq)
  // gemm_nt_w4c_kernel, data gradient followed by the input stage's gate backward (DrnGemmDesc::gb_*): the product is not stored as
  // C but as dct[c][m] = bf16(product) * gate[m / Lout][c], with dgate[seq][c] = sum_t product * act, dsum[seq][c] = sum_t of dct's values
  const void* gb_act;
  void* gb_dct;
  float* gb_dgate;
  float* gb_dsum;
  int gb_ld_act, gb_ldt;
};
struct GemmParams {
  int ngroups;
  int xcd_swizzle;
  int ksplit;       // > 1: gridDim.y splits of the K loop; partial tiles are exchanged through `ws` and summed by the
                    // LAST-ARRIVING split of each tile, which then runs the epilogue (single group only)
  int nblocks;      // gridDim.x (the XCD-aware tile order needs it; read from here it arrives with the one descriptor fetch)
  float* ws;        // [tiles][ksplit][TM*TN] fp32, accumulator-native order (one 16-byte piece per thread and MFMA tile)
  int* counters;    // [tiles] arrival counters, zero on entry, re-armed by the last arriver
  GemmProb p[DRN_MAX_GROUPS];
};

// conv -> BatchNorm(train) -> ReLU in ONE launch (drn_conv_bn_train, gemm_nt_bn.hip): per group, what bn_train_apply_kernel
// would need; the raw conv output stays GemmProb::C (backward reads it), the statistics GemmProb::stats.
struct BnFuse {
  void* out;             // [M][ld_out] normalised (+ReLU, + upsample chain) output
  void* gated;           // or NULL: out * gate[seq] (GemmProb::gate / ldg; query gating of the NEXT layer, model/backbone.py:28-30)
  float* ss;             // [2][N] scale, shift (backward recomputes the ReLU mask from them)
  float* save;           // [2][N] mean, invstd
  const float* gamma;
  const float* beta;
  const float* cbias;    // or NULL: conv bias (shifts running_mean only)
  float* rm;             // running_mean / running_var or NULL
  float* rv;
  float momentum, eps;
  int ld_out, ld_gated;
  int slabs;             // ceil(M / 128)
  int up_group;          // chain (FPN laterals, model/FPN.py:63-68): out += nearest_x2(out of group up_group), or -1
  int rs_owner;          // 1: this group's first-row workgroups update the running statistics ...
  int rs_mask;           // ... of itself and then of these later groups (bit h), which share the BatchNorm module, in order
};
struct GemmParamsBn : GemmParams {
  // GemmProb::stats of every group points into the TAGGED statistics workspace (64-bit {value, tag} pairs, bn_merge.h);
  // GemmParams::counters carries the address of the launch generation word (fetched with the descriptor header)
  BnFuse bn[DRN_MAX_GROUPS];
  int bn_relu;
  int bn_chain;          // 1: some group has up_group >= 0 (raw outputs are exchanged between workgroups)
};

constexpr bool getenv_free_scalar_w = false;   // flip to try the scalar wave index on the 8-wave tile as well
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// Ordering point for LDS traffic that stays inside ONE wavefront (a per-wave staging patch): the LDS unit executes a
// wave's DS instructions in issue order, so only the compiler must be kept from reordering -- no workgroup barrier.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// -DDRN_NT_PHASES: cycle stamps (s_memtime) of workgroup 0 / thread 0 inside the epilogue, rows 4000.. of the phase table
#ifdef DRN_NT_PHASES
static __device__ long long g_nt_epi_cyc[64];
#define EPI_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_nt_epi_cyc[(i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define EPI_STAMP(i) do { } while (0)
#endif

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16x8 frag;
  // rows [mi0, mi0+MH) of the MI x NI grid of 16x16 MFMA tiles
  template <int MI, int NI, int MH>
  static __device__ __forceinline__ void part(const frag (&a)[MI], const frag (&b)[NI], f32x4 (&acc)[MI][NI], int mi0) {
#pragma unroll
    for (int mi = 0; mi < MH; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi0 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi0 + mi], b[ni], acc[mi0 + mi][ni], 0, 0, 0);
  }
};
template <> struct Mma<float> {
  typedef f32x4 frag;
  template <int MI, int NI, int MH>
  static __device__ __forceinline__ void part(const frag (&a)[MI], const frag (&b)[NI], f32x4 (&acc)[MI][NI], int mi0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < MH; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi0 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi0 + mi][j], b[ni][j], acc[mi0 + mi][ni], 0, 0, 0);
  }
};

// The launch descriptor (GemmParams, 576 bytes of kernel arguments) is fetched by the LANES in one memory trip -- lane i takes
// dword i of the header and of each group's GemmProb -- and handed to the scalar unit with v_readlane.  Left to the compiler the
// fields arrive through a chain of ~8 DEPENDENT scalar loads (group lookup, then fields as they are first used, each behind its own
// s_waitcnt), and the scalar cache starts every kernel cold: 1.6-2.0 us before a workgroup issued its first operand load
// (-DDRN_NT_PHASES), in every one of the ~25 launches of a step.
constexpr int NT_HDR_DW = (int)offsetof(GemmParams, p) / 4;
constexpr int NT_PROB_DW = (int)sizeof(GemmProb) / 4;
static_assert(NT_HDR_DW <= 64 && NT_PROB_DW <= 64 && sizeof(GemmProb) % 4 == 0 && offsetof(GemmParams, p) % 4 == 0, "one lane per dword");
struct NtHeader {
  int ngroups, xcd_swizzle, ksplit, nblocks;
  int confirm;      // DRN_XCHG_CONFIRM taken out of GemmParams::ksplit
  float* ws;
  int* counters;
};
__device__ __forceinline__ unsigned nt_rl(unsigned v, int lane) { return (unsigned)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ void nt_fetch(const GemmParams& P, NtHeader& H, GemmProb& pr, int& g_out, int bid_for_group) {
  const unsigned* kp = (const unsigned*)&P;
  const int l = threadIdx.x & 63;
  const unsigned hv = kp[l < NT_HDR_DW ? l : 0];
  unsigned gv[DRN_MAX_GROUPS];
#pragma unroll
  for (int i = 0; i < DRN_MAX_GROUPS; ++i) gv[i] = kp[NT_HDR_DW + i * NT_PROB_DW + (l < NT_PROB_DW ? l : 0)];
  __builtin_amdgcn_sched_barrier(0);            // all five loads are in flight before the first v_readlane waits for one
  H.ngroups = (int)nt_rl(hv, (int)offsetof(GemmParams, ngroups) / 4);
  H.xcd_swizzle = (int)nt_rl(hv, (int)offsetof(GemmParams, xcd_swizzle) / 4);
  H.ksplit = (int)nt_rl(hv, (int)offsetof(GemmParams, ksplit) / 4);
  H.confirm = H.ksplit & (DRN_XCHG_CONFIRM | DRN_XCHG_READBACK);
  H.ksplit &= ~(DRN_XCHG_CONFIRM | DRN_XCHG_READBACK);        // (gemm_nt_w4h_kernel's own flag, W4H_TAPIL, stays)
  H.nblocks = (int)nt_rl(hv, (int)offsetof(GemmParams, nblocks) / 4);
  H.ws = (float*)(((unsigned long long)nt_rl(hv, (int)offsetof(GemmParams, ws) / 4 + 1) << 32) | nt_rl(hv, (int)offsetof(GemmParams, ws) / 4));
  H.counters = (int*)(((unsigned long long)nt_rl(hv, (int)offsetof(GemmParams, counters) / 4 + 1) << 32) | nt_rl(hv, (int)offsetof(GemmParams, counters) / 4));
  int bid = bid_for_group;
  if (H.xcd_swizzle) {
    const int nb = H.nblocks, q = nb >> 3, r = nb & 7, xcd = bid & 7, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  int g = 0;
  unsigned sel = gv[0];
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < H.ngroups && bid >= (int)nt_rl(gv[i], (int)offsetof(GemmProb, tile_start) / 4)) {
      g = i;
      sel = gv[i];
    }
  unsigned* q = (unsigned*)&pr;
#pragma unroll
  for (int i = 0; i < NT_PROB_DW; ++i) q[i] = nt_rl(sel, i);
  g_out = g;
}
// The addresses nt_fetch put together from lanes (and the ones a kernel reads from its argument block by a computed index) are generic
// pointers to the compiler: every access through them would be a FLAT instruction (common.h, as_global).  The 4-wave kernels call this
// AFTER their loop statement -- the loop takes raw addresses, and nothing more should be alive across it.
__device__ __forceinline__ void nt_globalize(GemmProb& pr) {
  pr.A = as_global(pr.A); pr.B = as_global(pr.B); pr.C = as_global(pr.C); pr.C2 = as_global(pr.C2);
  pr.bias = as_global(pr.bias); pr.gate = as_global(pr.gate); pr.stats = as_global(pr.stats); pr.sumsq = as_global(pr.sumsq);
  pr.gb_act = as_global(pr.gb_act); pr.gb_dct = as_global(pr.gb_dct); pr.gb_dgate = as_global(pr.gb_dgate); pr.gb_dsum = as_global(pr.gb_dsum);
}
__device__ __forceinline__ void nt_globalize(NtHeader& H) { H.ws = as_global(H.ws); H.counters = as_global(H.counters); }

// Workgroup -> (tile row, tile column) inside its group.
template <int TM>
__device__ __forceinline__ void nt_locate(const NtHeader& H, const GemmProb& pr, int& tm_out, int& tn_out) {
  // XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (speed only, never correctness), so give each
  // XCD a CONTIGUOUS run of logical tiles (same A row-panels, all B column-panels) instead of every 8th one -- the A panel
  // of a tile row is then fetched into one L2 instead of eight.  Bijective for any grid size.
  int bid = blockIdx.x;
  if (H.xcd_swizzle) {
    const int nb = H.nblocks, q = nb >> 3, r = nb & 7, xcd = bid & 7, j = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int t_local = bid - pr.tile_start;
  int tm = t_local / pr.tiles_n, tn = t_local - tm * pr.tiles_n;
  if (H.xcd_swizzle & 2) {
    // grouped order: walk 8 tile rows down one tile column before moving to the next column, so the ~32 tiles an XCD runs
    // at a time form an 8 x 4 block (each A panel shared by 4 workgroups, each B panel by 8) instead of 2 x 16
    const int tiles_m = (pr.M + TM - 1) / TM;
    const int per_group = 8 * pr.tiles_n;
    const int gid = t_local / per_group;
    const int first_m = gid * 8;
    const int gsm = min(tiles_m - first_m, 8);
    const int r = t_local - gid * per_group;
    tm = first_m + r % gsm;
    tn = r / gsm;
  }
  tm_out = tm;
  tn_out = tn;
}

// BatchNorm statistics of a tile's raw fp32 accumulators (see the comment inside).  COHERENT: write-through stores, for readers
// in OTHER workgroups of the same launch (nt_epilogue_bn).
template <int MODE>
__device__ __forceinline__ void nt_st_stat(float* stats, long idx, float v, unsigned tag) {
  if constexpr (MODE == BN_ST_TAGGED)      // one 8-byte write-through store: value and tag become visible together
    __hip_atomic_store((unsigned long long*)stats + idx, bn_tag_pack(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else stats[idx] = v;
}
template <int WM, int WN, int MI, int NI, int MODE>
__device__ __forceinline__ void nt_bn_stats(f32x4 (&acc)[MI][NI], float* shs, float* __restrict__ stats, const int M, const int N,
                                            const int m0, const int n0, const int tm, const unsigned tag = 0u) {
  constexpr int NW = WM * WN, TM = WM * MI * 16, TN = WN * NI * 16;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int wr = w / WN, wc = w % WN;
    // BatchNorm statistics of the raw fp32 accumulators per 128-row slab, in the numerically robust (count, sum, M2)
    // form: stats[(slab*2 + 0)*N + n] = sum_rows x, stats[(slab*2 + 1)*N + n] = sum_rows (x - slab mean)^2.
    // bn_finalize_kernel merges the slabs with Chan's parallel-variance formula (no E[x^2]-E[x]^2 cancellation).
    constexpr int WROWS = MI * 16;            // rows owned by one wave row: 64 or 128
    constexpr int WPS = 128 / WROWS;          // wave rows per 128-row slab
    constexpr int SLABS = TM / 128;
    // shs: [WM][TN] partial sums, then [SLABS][TN] slab means
    float* shm = shs + WM * TN;
    // pass 1: column sums (rows >= M hold exact zeros)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float sm = 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) sm += acc[mi][ni][r];
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      if (l < 16) shs[wr * TN + wc * (NI * 16) + ni * 16 + l] = sm;
    }
    __syncthreads();
    for (int i = tid; i < SLABS * TN; i += 64 * NW) {
      const int n = i % TN, slab = i / TN;
      const int grow = tm * SLABS + slab;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < WPS; ++k) v += shs[(slab * WPS + k) * TN + n];
      const int rows = min(128, M - grow * 128);
      shm[i] = rows > 0 ? v / (float)rows : 0.f;
      if (n0 + n < N && rows > 0) nt_st_stat<MODE>(stats, ((long)grow * 2 + 0) * N + n0 + n, v, tag);
    }
    __syncthreads();
    // pass 2: centred sums of squares (only rows < M)
    const int slab_w = (wr * WROWS) / 128;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const float mean = shm[slab_w * TN + wc * (NI * 16) + ni * 16 + (l & 15)];
      float q = 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wr * WROWS + mi * 16 + (l >> 4) * 4 + r;
          const float dlt = acc[mi][ni][r] - mean;
          q += m < M ? dlt * dlt : 0.f;
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (l < 16) shs[wr * TN + wc * (NI * 16) + ni * 16 + l] = q;
    }
    __syncthreads();
    for (int i = tid; i < SLABS * TN; i += 64 * NW) {
      const int n = i % TN, slab = i / TN;
      const int grow = tm * SLABS + slab;
      if (n0 + n < N && grow * 128 < M) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < WPS; ++k) v += shs[(slab * WPS + k) * TN + n];
        nt_st_stat<MODE>(stats, ((long)grow * 2 + 1) * N + n0 + n, v, tag);
      }
    }
}

// The two store paths of nt_epilogue below, one 32-row chunk at a time, for kernels that do not hold their accumulators in a C++
// array (gemm_nt_w4.hip reads them out of fixed AGPRs chunk by chunk).  SAME statements as the loop bodies of nt_epilogue -- kept
// apart because calling these from nt_epilogue cost the general bf16 kernels registers (128x128 tile 118 -> 141 VGPRs = one
// workgroup per CU instead of two); tests/test_gemm_gpu.py holds the two kernels to equal bits.
// One 32-row chunk (two rows of a wave's grid of 16x16 MFMA tiles) of the fp32-destination epilogue: a2[mi2][ni] = the wave's
// accumulators of grid rows 2*ch + mi2.  wbuf: this wave's private LDS patch; mrow0 / ncol0: first row / column of the chunk.
template <int NI>
__device__ __forceinline__ void nt_epi_chunk_f32(const GemmProb& pr, const f32x4 (&a2)[2][NI], char* wbuf, const int mrow0,
                                                 const int ncol0, const bool vec4) {
  constexpr int WCOLS = NI * 16, PITCHF = WCOLS * 4 + 16, LPRF = WCOLS / 4, RPIF = 64 / LPRF;
  const int l = threadIdx.x & 63, M = pr.M, N = pr.N;
  float* __restrict__ Cf = (float*)pr.C;
#pragma unroll
  for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        *((float*)(wbuf + (mi2 * 16 + (l >> 4) * 4 + r) * PITCHF) + ni * 16 + (l & 15)) = a2[mi2][ni][r];
  wave_lds_sync();   // the patch is private to this wave: LDS executes a wave's accesses in order
#pragma unroll
  for (int it = 0; it < 32 / RPIF; ++it) {  // LPRF lanes per row, RPIF rows per instruction
    const int rl = it * RPIF + l / LPRF, cv = l % LPRF;
    const int m = mrow0 + rl, n = ncol0 + cv * 4;
    if (m < M && n < N) {
      f32x4 v = *(const f32x4*)(wbuf + rl * PITCHF + cv * 16);
      float* g = Cf + ((long)m * pr.ldc + n);
      if (pr.bias)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += n + k < N ? pr.bias[n + k] : 0.f;
      if (vec4 && n + 4 <= N && !pr.accumulate) {
        *(f32x4*)g = v;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (n + k < N) g[k] = pr.accumulate ? g[k] + v[k] : v[k];
      }
    }
  }
  wave_lds_sync();   // the patch is private to this wave: LDS executes a wave's accesses in order
}

// One 32-row chunk of the coalesced dtype-destination epilogue (bias, optional pre-gate copy C2, gate), see nt_epilogue.
template <typename T, int NI>
__device__ __forceinline__ void nt_epi_chunk_vec(const GemmProb& pr, const f32x4 (&a2)[2][NI], char* wbuf, const int mrow0,
                                                 const int ncol0_, const float (&bias_v)[NI]) {
  constexpr int WCOLS = NI * 16, PITCH = WCOLS * (int)sizeof(T) + 16, LPR = WCOLS * (int)sizeof(T) / 16, RPI = 64 / LPR;
  constexpr int VEC = 16 / (int)sizeof(T);
  const int l = threadIdx.x & 63, M = pr.M, N = pr.N;
  T* __restrict__ Cg = (T*)pr.C;
  T* __restrict__ C2g = (T*)pr.C2;
  const int npass = C2g ? 2 : 1;
  if constexpr (std::is_same<T, bf16_t>::value) {
    // Fast path (bf16, the chunk's 32 rows x WCOLS columns all inside the matrix and -- if gated -- inside ONE sequence, no
    // accumulate): the generic path below costs one ds_write_b16, one row -> sequence division and one gate load PER ELEMENT
    // behind per-element masks (the 256x256 tile's epilogue took 6.9 us, mostly this).  Here the gate row is loaded once
    // per chunk, and the 4 rows x 4 columns a quad of lanes holds are transposed inside the quad (packed bf16 pairs: 3 DPP
    // moves + 2 byte permutes + 3 selects) so that every lane owns 4 consecutive columns of one row: one ds_write_b64 where
    // there were four ds_write_b16.  Same conversions, same values, same 16-byte global stores.
    const int ncol0 = ncol0_;
    const int sq0 = mrow0 / pr.Lout;
    const bool fast_w = mrow0 + 32 <= M && ncol0 + WCOLS <= N && !pr.accumulate &&
                        (!pr.gate || (mrow0 + 31) / pr.Lout == sq0);
    if (__builtin_amdgcn_readfirstlane((int)fast_w)) {
      float gv[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) gv[ni] = pr.gate ? pr.gate[(long)sq0 * pr.ldg + ncol0 + ni * 16 + (l & 15)] : 1.f;
      const bool j0 = l & 1, j1 = l & 2;
      const unsigned sel1 = j0 ? 0x03020706u : 0x05040100u;
      for (int pass = 0; pass < npass; ++pass) {
        const bool gated = pr.gate && pass == npass - 1;
        T* dst = (C2g && pass == 0) ? C2g : Cg;
        const int ldd = (C2g && pass == 0) ? pr.ldc2 : pr.ldc;
#pragma unroll
        for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = a2[mi2][ni][r] + bias_v[ni];
              if (gated) v[r] *= gv[ni];
            }
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t b01 = {(bf16_t)v[0], (bf16_t)v[1]}, b23 = {(bf16_t)v[2], (bf16_t)v[3]};
            const unsigned p0 = __builtin_bit_cast(unsigned, b01), p1 = __builtin_bit_cast(unsigned, b23);   // (row r, row r+1) of this lane's column
            const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p0, 0xB1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
            const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p1, 0xB1, 0xf, 0xf, false);
            // even lane: (own.lo, partner.lo) = row 2*r1, columns (j, j+1); odd lane: (partner.hi, own.hi) = row 2*r1 + 1, columns (j-1, j)
            const unsigned q0 = __builtin_amdgcn_perm(r0, p0, sel1), q1 = __builtin_amdgcn_perm(r1, p1, sel1);
            const unsigned snd = j1 ? q0 : q1;
            const unsigned rcv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
            const unsigned o0 = j1 ? rcv : q0, o1 = j1 ? q1 : rcv;         // row (l & 3), columns 0-1 and 2-3 of the quad's four
            const int rl = mi2 * 16 + (l >> 4) * 4 + (l & 3);
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            *(u32x2_t*)(wbuf + rl * PITCH + (ni * 16 + ((l & 15) >> 2) * 4) * 2) = (u32x2_t){o0, o1};
          }
        EPI_STAMP(10);
        wave_lds_sync();
        EPI_STAMP(11);
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
          const int rl = it * RPI + l / LPR, cv = l % LPR;
          const uint4 raw = *(const uint4*)(wbuf + rl * PITCH + cv * 16);
#ifdef DRN_EPI_NOSTORE      // (timing experiment, scripts/experiments/w4_phases.py: the epilogue without its global stores)
          asm volatile("" : : "v"(raw.x), "v"(raw.y), "v"(raw.z), "v"(raw.w));
#else
          *(uint4*)(dst + ((long)(mrow0 + rl) * ldd + ncol0 + cv * VEC)) = raw;
#endif
        }
        EPI_STAMP(12);
        wave_lds_sync();
        EPI_STAMP(13);
      }
      return;
    }
  }
  for (int pass = 0; pass < npass; ++pass) {
    const bool gated = pr.gate && pass == npass - 1;
    T* dst = (C2g && pass == 0) ? C2g : Cg;
    const int ldd = (C2g && pass == 0) ? pr.ldc2 : pr.ldc;
#pragma unroll
    for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = mi2 * 16 + (l >> 4) * 4 + r;
        const int m = mrow0 + rl;
        const float* grow = (gated && m < M) ? pr.gate + (long)(m / pr.Lout) * pr.ldg : nullptr;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          float v = a2[mi2][ni][r] + bias_v[ni];
          if (grow) {
            const int n = ncol0_ + ni * 16 + (l & 15);
            v *= n < N ? grow[n] : 0.f;
          }
          DT<T>::st((T*)(wbuf + rl * PITCH) + ni * 16 + (l & 15), v);
        }
      }
    wave_lds_sync();   // the patch is private to this wave: LDS executes a wave's accesses in order
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int rl = it * RPI + l / LPR, cv = l % LPR;
      const int m = mrow0 + rl, n = ncol0_ + cv * VEC;
      if (m < M && n < N) {
        const uint4 raw = *(const uint4*)(wbuf + rl * PITCH + cv * 16);
        T* g = dst + ((long)m * ldd + n);
        const bool acc_c = pr.accumulate && dst == Cg;
        if (n + VEC <= N && !acc_c) {
          *(uint4*)g = raw;
        } else {
          const T* e = (const T*)&raw;
#pragma unroll
          for (int k = 0; k < VEC; ++k)
            if (n + k < N) DT<T>::st(g + k, acc_c ? DT<T>::ld(e + k) + DT<T>::ld(g + k) : DT<T>::ld(e + k));
        }
      }
    }
    wave_lds_sync();   // the patch is private to this wave: LDS executes a wave's accesses in order
  }
}

// Epilogue shared by the NT kernels.  acc[mi][ni][r]: m = wr*MI*16 + mi*16 + (l>>4)*4 + r, n = wc*NI*16 + ni*16 + (l&15).
// Enter after a workgroup barrier that follows the last LDS read of the main loop (it reuses `smem`).
template <typename T, int WM, int WN, int MI, int NI>
__device__ __forceinline__ void nt_epilogue(const NtHeader& P, const GemmProb& pr, f32x4 (&acc)[MI][NI], char* smem,
                                            const int m0, const int n0, const int tm) {
  constexpr int NW = WM * WN, TM = WM * MI * 16, TN = WN * NI * 16;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int wr = w / WN, wc = w % WN;
  const int M = pr.M, N = pr.N;
  (void)TM; (void)NW;
  if (pr.out_f32) {
    // fp32 destination (a weight gradient): each wave transposes its slab (NI*16 columns) through a private LDS patch, 32 rows
    // at a time, and writes 16-byte row segments.  Only bias / accumulate apply here.
    float* __restrict__ Cf = (float*)pr.C;
    constexpr int WCOLS = NI * 16;           // columns owned by one wave
    constexpr int PITCHF = WCOLS * 4 + 16;
    constexpr int LPRF = WCOLS / 4;          // lanes per staged row in the 16-byte read-back
    constexpr int RPIF = 64 / LPRF;          // rows per read-back instruction
    char* wbuf = smem + w * (32 * PITCHF);
    const bool vec4 = (pr.ldc % 4 == 0) && (((uintptr_t)Cf & 15) == 0);
#pragma unroll
    for (int ch = 0; ch < MI / 2; ++ch) {
      const int mrow0 = m0 + wr * (MI * 16) + ch * 32;
#pragma unroll
      for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            *((float*)(wbuf + (mi2 * 16 + (l >> 4) * 4 + r) * PITCHF) + ni * 16 + (l & 15)) = acc[ch * 2 + mi2][ni][r];
      wave_lds_sync();   // the patch is private to this wave: LDS executes a wave's accesses in order
#pragma unroll
      for (int it = 0; it < 32 / RPIF; ++it) {  // LPRF lanes per row, RPIF rows per instruction
        const int rl = it * RPIF + l / LPRF, cv = l % LPRF;
        const int m = mrow0 + rl, n = n0 + wc * (NI * 16) + cv * 4;
        if (m < M && n < N) {
          f32x4 v = *(const f32x4*)(wbuf + rl * PITCHF + cv * 16);
          float* g = Cf + ((long)m * pr.ldc + n);
          if (pr.bias)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += n + k < N ? pr.bias[n + k] : 0.f;
          if (vec4 && n + 4 <= N && !pr.accumulate) {
            *(f32x4*)g = v;
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (n + k < N) g[k] = pr.accumulate ? g[k] + v[k] : v[k];
          }
        }
      }
      wave_lds_sync();   // the patch is private to this wave: LDS executes a wave's accesses in order
    }
    return;
  }
  auto bn_stats = [&](float* shs) { nt_bn_stats<WM, WN, MI, NI, BN_ST_PLAIN>(acc, shs, pr.stats, M, N, m0, n0, tm); };

  T* __restrict__ Cg = (T*)pr.C;
  T* __restrict__ C2g = (T*)pr.C2;
  constexpr int VEC = 16 / (int)sizeof(T);
  const bool vec_ok = (pr.ldc % VEC == 0) && (((uintptr_t)Cg & 15) == 0) &&
                      (!C2g || ((pr.ldc2 % VEC == 0) && (((uintptr_t)C2g & 15) == 0)));
  const bool stats_first = !vec_ok;
  if (stats_first && pr.stats) {
    bn_stats((float*)smem);
    __syncthreads();
  }
  if (vec_ok) {
    // Coalesced epilogue: every wave transposes its slab (NI*16 columns) through a private LDS patch, 32 rows at a time, and
    // writes it out as 16-byte row segments (8 store instructions per wave and 64x64 bf16 tile instead of 64 two-byte
    // ones -- the narrow stores were ~8 us of issue-bound tail per launch).
    constexpr int WCOLS = NI * 16;                           // columns owned by one wave
    constexpr int PITCH = WCOLS * (int)sizeof(T) + 16;       // bytes per staged row (+16: conflict-free column writes)
    constexpr int LPR = WCOLS * (int)sizeof(T) / 16;         // lanes per staged row in the 16-byte read-back
    constexpr int RPI = 64 / LPR;                            // rows per read-back instruction
    char* wbuf = smem + w * (32 * PITCH);
    float bias_v[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wc * (NI * 16) + ni * 16 + (l & 15);
      bias_v[ni] = (pr.bias && n < N) ? pr.bias[n] : 0.f;
    }
    const int npass = C2g ? 2 : 1;
#pragma unroll
    for (int ch = 0; ch < MI / 2; ++ch) {
      const int mrow0 = m0 + wr * (MI * 16) + ch * 32;
      if constexpr (std::is_same<T, bf16_t>::value) {
        // Fast path (bf16, the chunk's 32 rows x WCOLS columns all inside the matrix and -- if gated -- inside ONE sequence, no
        // accumulate): the generic path below costs one ds_write_b16, one row -> sequence division and one gate load PER ELEMENT
        // behind per-element masks (the 256x256 tile's epilogue took 6.9 us, mostly this).  Here the gate row is loaded once
        // per chunk, and the 4 rows x 4 columns a quad of lanes holds are transposed inside the quad (packed bf16 pairs: 3 DPP
        // moves + 2 byte permutes + 3 selects) so that every lane owns 4 consecutive columns of one row: one ds_write_b64 where
        // there were four ds_write_b16.  Same conversions, same values, same 16-byte global stores.
        const int ncol0 = n0 + wc * WCOLS;
        const int sq0 = mrow0 / pr.Lout;
        const bool fast_w = mrow0 + 32 <= M && ncol0 + WCOLS <= N && !pr.accumulate &&
                            (!pr.gate || (mrow0 + 31) / pr.Lout == sq0);
        if (__builtin_amdgcn_readfirstlane((int)fast_w)) {
          float gv[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) gv[ni] = pr.gate ? pr.gate[(long)sq0 * pr.ldg + ncol0 + ni * 16 + (l & 15)] : 1.f;
          const bool j0 = l & 1, j1 = l & 2;
          const unsigned sel1 = j0 ? 0x03020706u : 0x05040100u;
          for (int pass = 0; pass < npass; ++pass) {
            const bool gated = pr.gate && pass == npass - 1;
            T* dst = (C2g && pass == 0) ? C2g : Cg;
            const int ldd = (C2g && pass == 0) ? pr.ldc2 : pr.ldc;
#pragma unroll
            for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  v[r] = acc[ch * 2 + mi2][ni][r] + bias_v[ni];
                  if (gated) v[r] *= gv[ni];
                }
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                const bf16x2_t b01 = {(bf16_t)v[0], (bf16_t)v[1]}, b23 = {(bf16_t)v[2], (bf16_t)v[3]};
                const unsigned p0 = __builtin_bit_cast(unsigned, b01), p1 = __builtin_bit_cast(unsigned, b23);   // (row r, row r+1) of this lane's column
                const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p0, 0xB1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
                const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p1, 0xB1, 0xf, 0xf, false);
                // even lane: (own.lo, partner.lo) = row 2*r1, columns (j, j+1); odd lane: (partner.hi, own.hi) = row 2*r1 + 1, columns (j-1, j)
                const unsigned q0 = __builtin_amdgcn_perm(r0, p0, sel1), q1 = __builtin_amdgcn_perm(r1, p1, sel1);
                const unsigned snd = j1 ? q0 : q1;
                const unsigned rcv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
                const unsigned o0 = j1 ? rcv : q0, o1 = j1 ? q1 : rcv;         // row (l & 3), columns 0-1 and 2-3 of the quad's four
                const int rl = mi2 * 16 + (l >> 4) * 4 + (l & 3);
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                *(u32x2_t*)(wbuf + rl * PITCH + (ni * 16 + ((l & 15) >> 2) * 4) * 2) = (u32x2_t){o0, o1};
              }
            wave_lds_sync();
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
              const int rl = it * RPI + l / LPR, cv = l % LPR;
              const uint4 raw = *(const uint4*)(wbuf + rl * PITCH + cv * 16);
              *(uint4*)(dst + ((long)(mrow0 + rl) * ldd + ncol0 + cv * VEC)) = raw;
            }
            wave_lds_sync();
          }
          continue;
        }
      }
      for (int pass = 0; pass < npass; ++pass) {
        const bool gated = pr.gate && pass == npass - 1;
        T* dst = (C2g && pass == 0) ? C2g : Cg;
        const int ldd = (C2g && pass == 0) ? pr.ldc2 : pr.ldc;
#pragma unroll
        for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = mi2 * 16 + (l >> 4) * 4 + r;
            const int m = mrow0 + rl;
            const float* grow = (gated && m < M) ? pr.gate + (long)(m / pr.Lout) * pr.ldg : nullptr;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              float v = acc[ch * 2 + mi2][ni][r] + bias_v[ni];
              if (grow) {
                const int n = n0 + wc * (NI * 16) + ni * 16 + (l & 15);
                v *= n < N ? grow[n] : 0.f;
              }
              DT<T>::st((T*)(wbuf + rl * PITCH) + ni * 16 + (l & 15), v);
            }
          }
        wave_lds_sync();   // the patch is private to this wave: LDS executes a wave's accesses in order
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
          const int rl = it * RPI + l / LPR, cv = l % LPR;
          const int m = mrow0 + rl, n = n0 + wc * (NI * 16) + cv * VEC;
          if (m < M && n < N) {
            const uint4 raw = *(const uint4*)(wbuf + rl * PITCH + cv * 16);
            T* g = dst + ((long)m * ldd + n);
            const bool acc_c = pr.accumulate && dst == Cg;
            if (n + VEC <= N && !acc_c) {
              *(uint4*)g = raw;
            } else {
              const T* e = (const T*)&raw;
#pragma unroll
              for (int k = 0; k < VEC; ++k)
                if (n + k < N) DT<T>::st(g + k, acc_c ? DT<T>::ld(e + k) + DT<T>::ld(g + k) : DT<T>::ld(e + k));
            }
          }
        }
        wave_lds_sync();   // the patch is private to this wave: LDS executes a wave's accesses in order
      }
    }
    // BatchNorm statistics AFTER the stores have been issued: the tile's write burst (the whole chip stores at once: 3-6 us at the
    // HBM write rate) drains while the three-barrier statistics pass runs, instead of starting behind it.  The statistics use the
    // LDS beyond the per-wave store patches.
    if (pr.stats && !stats_first) bn_stats((float*)(smem + NW * (32 * PITCH)));
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wr * (MI * 16) + mi * 16 + (l >> 4) * 4 + r;
      if (m >= M) continue;
      const float* grow = pr.gate ? pr.gate + (long)(m / pr.Lout) * pr.ldg : nullptr;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wc * (NI * 16) + ni * 16 + (l & 15);
        if (n >= N) continue;
        float v = acc[mi][ni][r];
        if (pr.bias) v += pr.bias[n];
        const long off = (long)m * pr.ldc + n;
        if (C2g) DT<T>::st(C2g + ((long)m * pr.ldc2 + n), v);
        if (grow) v *= grow[n];
        if (pr.accumulate) v += DT<T>::ld(Cg + off);
        DT<T>::st(Cg + off, v);
      }
    }
  }
}

// Optional per-wave timeline (build with -DDRN_NT_TRACE, scripts/experiments/nt_trace.py): workgroup 0 stamps s_memtime at
// seven points of every K-step into P.ws (pass a buffer through drn_gemm_nt_splitk with ksplit = 1).
// -DDRN_NT_PHASES (scripts/experiments/nt_phases.py): wall_clock64() (100 MHz) per workgroup at entry / staging state ready /
// first K-tile landed / K loop done / exit, read back with drn_debug_nt_phases() -- the fixed part of a small launch.
#ifdef DRN_NT_PHASES
static __device__ long long g_nt_phases[4096 * 8];
// (slots 5 / 6: s_memtime -- shader cycles -- at phases 1 / 2, the two ends of the K loop: cycles / wall time = the clock the loop ran at,
//  scripts/experiments/gemm_clock.py)
#define NT_PHASE(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) { g_nt_phases[blockIdx.x * 8 + (i)] = wall_clock64(); \
    if ((i) == 1 || (i) == 2) g_nt_phases[blockIdx.x * 8 + 4 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } } while (0)
#ifndef DRN_NT_PHASES_NAME
#define DRN_NT_PHASES_NAME drn_debug_nt_phases
#endif
extern "C" int DRN_NT_PHASES_NAME(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nt_phases), (size_t)n * 8); }
#else
#define NT_PHASE(i) do { } while (0)
#endif
#ifdef DRN_NT_TRACE
#define NT_STAMP(slot) do { if (P.ws && P.ksplit == 1 && blockIdx.x == 0 && l == 0 && kt - kt_lo < 64) \
    ((long long*)P.ws)[((w * 64) + (kt - kt_lo)) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define NT_STAMP(slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// conv -> BatchNorm1d (training) -> ReLU in ONE launch (model/basic_blocks.py:9-31; per-level-call statistics,
// model/fcos.py:93-102).  Before: the GEMM wrote the raw tile, bn_train_apply_kernel re-read it, normalised and wrote it again
// -- 7 extra launches and ~160 MB of re-reads per forward of FPN + heads.  Here a workgroup
//   1. publishes its per-128-row-slab (sum, M2) statistics as TAGGED 64-bit pairs {value, launch generation} with one
//      write-through store each (bn_merge.h), stages its tile through the per-wave LDS patches, stores the RAW rows (backward
//      needs them) and KEEPS the 16-byte row segments it stored in registers (exactly the bits the stand-alone pass would read);
//   2. merges the statistics of its tile column with the SAME routine, in the same order, as bn_train_apply_kernel -- polling the
//      pairs themselves until they carry this launch's tag: no arrival counter (the first version's per-column counter cost
//      15-20 us per launch: ~112 device-scope read-modify-writes on one address queue up at the memory side), no flag to clear,
//      one store -> load hand-off.  The launch is co-resident by construction (the launcher checks grid <= occupancy x CUs), so
//      this is a wait for the slowest neighbour, not a dependency on unscheduled work; bounded: after 2 s a reader raises the
//      watchdog counter and carries on.  Every workgroup of a column computes bit-identical scale / shift;
//   3. normalises (+ReLU) its registers and stores the output once; `gated` = out * gate[seq] (the next layer's query gating)
//      and the FPN top-down chain out_l += nearest_x2(out_{l+1}) (recomputed from the coarser levels' raw rows and statistics,
//      rounding where the stand-alone passes round) ride along;
//   4. first-row workgroups write scale/shift + (mean, invstd) and update the running statistics, groups that share a
//      BatchNorm module in group order; workgroup 0 finally advances the generation word -- after it has seen a tagged pair of
//      EVERY tile of the launch, i.e. after every workgroup has read the old value.
static __device__ int g_bn_fuse_timeouts;   // (one copy per translation unit; gemm_nt_bn.hip owns the live one)

__device__ __forceinline__ uint4 nt_ld16_coherent(const void* p) {
  const unsigned* q = (const unsigned*)p;      // four agent-scope dword loads (global_load_dword sc1), all in flight together
  uint4 v;
  v.x = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.z = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.w = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
__device__ __forceinline__ void nt_st16_coherent(void* p, const uint4 v) {
  // write-through 16-byte store.  ONE asm statement with its wait states: the compiler does not know this is a store and may
  // otherwise overwrite the data registers inside the hazard window of a > 64-bit VMEM store (see the split-K exchange).
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t r = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(r) : "memory");
}

template <typename T> struct NtSeg;
template <> struct NtSeg<bf16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void cvt(const uint4& r, float (&x)[8]) {
    const bf16x8 t = __builtin_bit_cast(bf16x8, r);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (float)t[k];
  }
  static __device__ __forceinline__ uint4 pack(const float (&x)[8]) {
    bf16x8 t;
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = (bf16_t)x[k];
    return __builtin_bit_cast(uint4, t);
  }
  static __device__ __forceinline__ float round(float v) { return (float)(bf16_t)v; }
};
template <> struct NtSeg<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void cvt(const uint4& r, float (&x)[4]) {
    const f32x4 t = __builtin_bit_cast(f32x4, r);
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = t[k];
  }
  static __device__ __forceinline__ uint4 pack(const float (&x)[4]) {
    f32x4 t;
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = x[k];
    return __builtin_bit_cast(uint4, t);
  }
  static __device__ __forceinline__ float round(float v) { return v; }
};

// gen: the launch generation this workgroup read at its start (GemmParams::counters points at the word).
template <typename T, int WM, int WN, int MI, int NI, bool CHAIN>
__device__ __forceinline__ void nt_epilogue_bn(const GemmParamsBn& K, int* gen_word, const unsigned gen, const GemmProb& pr, const int g,
                                               f32x4 (&acc)[MI][NI], char* smem, const int m0, const int n0, const int tm, const int tn) {
  constexpr int NW = WM * WN, TM = WM * MI * 16, TN = WN * NI * 16;
  constexpr int VEC = NtSeg<T>::VEC;
  constexpr int WCOLS = NI * 16;                           // columns owned by one wave
  constexpr int PITCH = WCOLS * (int)sizeof(T) + 16;       // bytes per staged row (+16: conflict-free column writes)
  constexpr int LPR = WCOLS * (int)sizeof(T) / 16;         // lanes per staged row in the 16-byte read-back
  constexpr int RPI = 64 / LPR;                            // rows per read-back instruction
  constexpr int NIT = 32 / RPI, NCH = MI / 2;              // row segments per lane: NCH x NIT
  constexpr int MAXLEV = CHAIN ? 3 : 1;                   // levels a workgroup normalises: its own + the coarser ones of the chain
  static_assert(NW == 8 && TN % 128 == 0, "the statistics merge runs 128 channels x 4 slab lanes on 512 threads");
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: patch addresses, tile offsets stay in SGPRs
  const int wr = w / WN, wc = w % WN;
  const int M = pr.M, N = pr.N, L = pr.Lout;
  T* __restrict__ rawg = (T*)pr.C;
  const BnFuse& F = K.bn[g];
  const BnTagged tg{gen + 1u, &g_bn_fuse_timeouts};
  // The kept row segments: registers -- or, for the bf16 128x128 tile (two workgroups per CU, 128 registers per lane, where 16
  // more live registers across the statistics merge meant scratch and with it no guaranteed occupancy), the LDS patches
  // themselves, sized for the whole 64-row wave tile and read a second time after the merge.
  constexpr bool SEG_LDS = sizeof(T) == 2 && MI * NI == 8;
  constexpr int PROWS = SEG_LDS ? 64 : 32;                 // rows per wave patch
  // LDS: per-wave store patches | statistics scratch | merge lanes | scale / shift per chain level
  char* wbuf = smem + w * (PROWS * PITCH);
  float* st_scratch = (float*)(smem + NW * (PROWS * PITCH));
  double (*shd)[128] = (double (*)[128])(smem + NW * (PROWS * PITCH) + (WM + TM / 128) * TN * 4);
  float* s_sc = (float*)((char*)shd + 4 * 128 * 8);       // [MAXLEV][TN]
  float* s_sh = s_sc + MAXLEV * TN;

  const int cv = l % LPR;
  const int ncol = n0 + wc * WCOLS + cv * VEC;             // this lane's first column (N % TN == 0: always inside)
  uint4 seg[SEG_LDS ? 1 : NCH][SEG_LDS ? 1 : NIT];
  // raw tile -> LDS patches -> 16-byte row segments: stored, and kept
  auto stage_and_store = [&]() {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int mrow0 = m0 + wr * (MI * 16) + ch * 32;
      char* wbuf = smem + w * (PROWS * PITCH) + (SEG_LDS ? ch * 32 * PITCH : 0);
      if constexpr (std::is_same<T, bf16_t>::value) {
        // quad transposition of nt_epilogue's fast path: a lane ends up with 4 consecutive columns of one row (one ds_write_b64)
        const bool j0 = l & 1, j1 = l & 2;
        const unsigned sel1 = j0 ? 0x03020706u : 0x05040100u;
#pragma unroll
        for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const f32x4 v = acc[ch * 2 + mi2][ni];
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t b01 = {(bf16_t)v[0], (bf16_t)v[1]}, b23 = {(bf16_t)v[2], (bf16_t)v[3]};
            const unsigned p0 = __builtin_bit_cast(unsigned, b01), p1 = __builtin_bit_cast(unsigned, b23);
            const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p0, 0xB1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
            const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p1, 0xB1, 0xf, 0xf, false);
            const unsigned q0 = __builtin_amdgcn_perm(r0, p0, sel1), q1 = __builtin_amdgcn_perm(r1, p1, sel1);
            const unsigned snd = j1 ? q0 : q1;
            const unsigned rcv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)snd, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
            const unsigned o0 = j1 ? rcv : q0, o1 = j1 ? q1 : rcv;
            const int rl = mi2 * 16 + (l >> 4) * 4 + (l & 3);
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            *(u32x2_t*)(wbuf + rl * PITCH + (ni * 16 + ((l & 15) >> 2) * 4) * 2) = (u32x2_t){o0, o1};
          }
      } else {
#pragma unroll
        for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              *((float*)(wbuf + (mi2 * 16 + (l >> 4) * 4 + r) * PITCH) + ni * 16 + (l & 15)) = acc[ch * 2 + mi2][ni][r];
      }
      wave_lds_sync();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int rl = it * RPI + l / LPR;
        const uint4 sv = *(const uint4*)(wbuf + rl * PITCH + cv * 16);
        if constexpr (!SEG_LDS) seg[ch][it] = sv;
        const int m = mrow0 + rl;
        if (m < M) {
          T* dst = rawg + ((long)m * pr.ldc + ncol);
          if constexpr (CHAIN) nt_st16_coherent(dst, sv);       // other workgroups of this launch read it (the top-down chain)
          else *(uint4*)dst = sv;
        }
      }
      wave_lds_sync();
    }
  };
  // ---- 1. publish.  Plain launches: the statistics first (the accumulators die while the tile is staged, and the tagged pairs
  // get the longest head start).  Chain launches: a reader that sees this tile's pairs also reads its RAW rows, so those go
  // out first and have reached the coherence point (vmcnt(0) + barrier) before the first pair is stored.
  if constexpr (CHAIN) {
    stage_and_store();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    nt_bn_stats<WM, WN, MI, NI, BN_ST_TAGGED>(acc, st_scratch, pr.stats, M, N, m0, n0, tm, tg.want);
  } else if constexpr (sizeof(T) == 4 && MI * NI == 8) {
    // (exact-f32 128x128 tile: in this order the 128-register variant stays out of scratch -- a kernel with scratch cannot
    // count on full occupancy, which the column wait needs)
    stage_and_store();
    nt_bn_stats<WM, WN, MI, NI, BN_ST_TAGGED>(acc, st_scratch, pr.stats, M, N, m0, n0, tm, tg.want);
  } else {
    nt_bn_stats<WM, WN, MI, NI, BN_ST_TAGGED>(acc, st_scratch, pr.stats, M, N, m0, n0, tm, tg.want);
    stage_and_store();
  }

  NT_PHASE(4);
  // ---- 2. merge: own group, then the coarser levels of the chain (each merge polls its pairs until they carry this tag)
  const int ci = tid & 127;
  int lev_group[MAXLEV];
  int nlev = 0;
  {
    int h = g;
#pragma unroll
    for (int lev = 0; lev < MAXLEV; ++lev) {      // (static indices only: the array stays in registers)
      lev_group[lev] = h;
      if (h >= 0) {
        nlev = lev + 1;
        h = K.bn[h].up_group;
      }
    }
  }
  if (w == 0) {                                   // one wave waits (quietly) for the column's slabs of every level it needs
#pragma unroll
    for (int lev = 0; lev < MAXLEV; ++lev)
      if (lev < nlev) bn_wait_slabs(K.p[lev_group[lev]].stats, K.bn[lev_group[lev]].slabs, N, n0, tg);
    if (tm == 0 && F.rs_owner && F.rs_mask)
      for (int h = g + 1; h < K.ngroups; ++h)
        if ((F.rs_mask >> h) & 1) bn_wait_slabs(K.p[h].stats, K.bn[h].slabs, N, n0, tg);
  }
  __syncthreads();
  NT_PHASE(5);
#pragma unroll
  for (int lev = 0; lev < MAXLEV; ++lev) {
    if (lev >= nlev) break;
    const int h = lev_group[lev];
    const BnFuse& Fh = K.bn[h];
    const GemmProb& ph = K.p[h];
#pragma unroll
    for (int half = 0; half < TN / 128; ++half) {
      const int cbase = n0 + half * 128;
      double mean, var;
      bn_merge_cols<128, BN_ST_TAGGED, (sizeof(T) == 4 ? 4 : 8)>(ph.stats, Fh.slabs, ph.M, N, cbase, shd, mean, var, tg);
      if (tid < 128) {
        const int c = cbase + ci;
        float sc, sh, invstd;
        bn_scale_shift(mean, var, Fh.eps, Fh.gamma[c], Fh.beta[c], sc, sh, invstd);
        s_sc[lev * TN + half * 128 + ci] = sc;
        s_sh[lev * TN + half * 128 + ci] = sh;
        if (lev == 0 && tm == 0) {                        // ---- 4. once per channel and group
          Fh.ss[c] = sc;
          Fh.ss[N + c] = sh;
          Fh.save[c] = (float)mean;
          Fh.save[N + c] = invstd;
          if (Fh.rs_owner) bn_running_update(mean, var, ph.M, Fh.momentum, Fh.cbias ? Fh.cbias[c] : 0.f, Fh.rm, Fh.rv, c);
        }
      }
    }
  }
  __syncthreads();

  NT_PHASE(6);
  // ---- 3. normalise the kept row segments (scale / shift of the lane's columns are read from LDS where they are used: the
  // registers they would occupy across the whole loop are what pushed the 128-register variants into scratch)
  const float* my_sc = s_sc + wc * WCOLS + cv * VEC;
  const float* my_sh = s_sh + wc * WCOLS + cv * VEC;
  const bool relu = K.bn_relu != 0;
  T* __restrict__ outg = (T*)F.out;
  T* __restrict__ gatedg = (T*)F.gated;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int mrow0 = m0 + wr * (MI * 16) + ch * 32;
    uint4 upseg[NIT][MAXLEV > 1 ? MAXLEV - 1 : 1];
    int sq[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = min(mrow0 + it * RPI + l / LPR, M - 1);
      sq[it] = m / L;
      if (nlev > 1) {
        const int t = m - sq[it] * L;
#pragma unroll
        for (int lev = 1; lev < MAXLEV; ++lev)
          if (lev < nlev) {
            const GemmProb& ph = K.p[lev_group[lev]];
            upseg[it][lev - 1] = nt_ld16_coherent((const T*)ph.C + ((long)sq[it] * (L >> lev) + (t >> lev)) * ph.ldc + ncol);
          }
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = mrow0 + it * RPI + l / LPR;
      float x[VEC];
      uint4 sv;
      if constexpr (SEG_LDS) sv = *(const uint4*)(wbuf + (ch * 32 + it * RPI + l / LPR) * PITCH + cv * 16);
      else sv = seg[ch][it];
      NtSeg<T>::cvt(sv, x);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float y = fmaf(x[k], my_sc[k], my_sh[k]);
        x[k] = relu ? fmaxf(y, 0.f) : y;
      }
      if (nlev > 1) {
        float u[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) u[k] = 0.f;
#pragma unroll
        for (int lev = MAXLEV - 1; lev >= 1; --lev)
          if (lev < nlev) {
            float xv[VEC];
            NtSeg<T>::cvt(upseg[it][lev - 1], xv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
              const float y = fmaf(xv[k], my_sc[lev * TN + k], my_sh[lev * TN + k]);
              const float v = (relu ? fmaxf(y, 0.f) : y) + (lev + 1 < nlev ? u[k] : 0.f);
              u[k] = NtSeg<T>::round(v);                   // (the stand-alone pass stores out_{lev} in T and re-reads it)
            }
          }
#pragma unroll
        for (int k = 0; k < VEC; ++k) x[k] += u[k];
      }
      if (m < M) {
        *(uint4*)(outg + ((long)m * F.ld_out + ncol)) = NtSeg<T>::pack(x);
        if (gatedg) {
          const float* gp = pr.gate + (long)sq[it] * pr.ldg + ncol;
#pragma unroll
          for (int k = 0; k < VEC; ++k) x[k] *= gp[k];
          *(uint4*)(gatedg + ((long)m * F.ld_gated + ncol)) = NtSeg<T>::pack(x);
        }
      }
    }
  }

  NT_PHASE(7);
  // ---- 4b. running statistics of the later groups that share this group's BatchNorm module, in group order
  if (tm == 0 && F.rs_owner && F.rs_mask) {
    for (int h = g + 1; h < K.ngroups; ++h) {
      if (!((F.rs_mask >> h) & 1)) continue;
      const BnFuse& Fh = K.bn[h];
      const GemmProb& ph = K.p[h];
#pragma unroll
      for (int half = 0; half < TN / 128; ++half) {
        const int cbase = n0 + half * 128;
        double mean, var;
        bn_merge_cols<128, BN_ST_TAGGED, (sizeof(T) == 4 ? 4 : 8)>(ph.stats, Fh.slabs, ph.M, N, cbase, shd, mean, var, tg);
        if (tid < 128) bn_running_update(mean, var, ph.M, Fh.momentum, Fh.cbias ? Fh.cbias[cbase + ci] : 0.f, Fh.rm, Fh.rv, cbase + ci);
      }
    }
  }

  // ---- 4c. workgroup 0 advances the generation once EVERY tile of the launch has published under the old one (a tile's pairs
  // are written after its workgroup read the word): one pair per (group, slab, 128-channel block) is looked at
  if (blockIdx.x == 0) {
    const int cblocks = N >> 7;
    for (int h = 0; h < K.ngroups; ++h) {
      const unsigned long long* base = (const unsigned long long*)K.p[h].stats;
      const int cnt = K.bn[h].slabs * cblocks;
      const long long t0 = wall_clock64();
      for (int i = tid; i < cnt; i += 64 * NW) {
        const int slab = i / cblocks, cb = i - slab * cblocks;
        const unsigned long long* q = base + ((long)slab * 2 + 1) * N + cb * 128;
        while ((unsigned)(bn_ld_pair(q) >> 32) != tg.want)
          if (bn_wait_expired(t0, tg.timeouts)) break;
      }
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(gen_word, (int)(gen + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// STAGES-deep LDS ring (STAGES x 32 KB).  Iteration kt: wait until tile kt's global_load_lds have landed with a COUNTED
// vmcnt (the STAGES-2 younger tiles stay in flight across the barrier), one raw s_barrier, issue tile kt+STAGES-1 into the
// slot everybody just finished reading, then MFMA on tile kt.  Past-the-end tiles read the zero page so the count is uniform.
// FAST (every group has Cin % BK == 0, so a K-tile never straddles two taps): the tap and channel offset of a tile are
// wave-uniform scalars advanced incrementally, and each thread keeps 4+4 precomputed row pointers -- ~10 VALU per
// global_load_lds instead of a per-lane integer division and 64-bit multiply.  The generic path keeps those.
// Tile shape: WM x WN waves, each owning MI x NI MFMA tiles of 16x16 -> TM = WM*MI*16 rows, TN = WN*NI*16 columns.
//   <2,2,4,4>: 128x128, 4 waves, 32 KB/stage (2 workgroups per CU at 2 stages)    -- general purpose
//   <2,4,4,2>: 128x128, 8 waves (2 per SIMD)                                       -- launches of <= 256 tiles (one
//              workgroup per CU): with 4 waves each wave spends ~800 cycles per K-step just ISSUING its 8 global_load_lds
//              (per-wave timeline: 2300 cycles per K-step for 512 cycles of MFMA); 8 waves halve that and overlap it
//   <2,4,8,4>: 256x256, 8 waves (2 per SIMD), 64 KB/stage, 2 stages               -- large GEMMs: half the operand
//              traffic and half the global_load_lds / ds_read per MFMA
template <typename T, int STAGES, bool FAST, int WM, int WN, int MI, int NI, bool BNF = false, bool CHAIN = false>
__global__ __launch_bounds__(64 * WM * WN, (BNF && MI * NI == 8 && STAGES == 2) ? 4 : (WM * WN == 8 ? 2 : (STAGES <= 2 ? 2 : 1))) void conv_gemm_nt_kernel(
    const std::conditional_t<BNF, GemmParamsBn, GemmParams> P_arg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CH = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int BK = 8 * CH;               // elements per K-step (128 bytes)
  constexpr int NW = WM * WN, TM = WM * MI * 16, TN = WN * NI * 16;
  constexpr int PA = TM / (8 * NW), PB = TN / (8 * NW);      // 1-KB staging pieces per wave and operand
  constexpr int PMAX = PA > PB ? PA : PB;
  constexpr int A_BYTES = TM * 128, STAGE_B = (TM + TN) * 128;
  static_assert(PA * 8 * NW == TM && PB * 8 * NW == TN && PA == PB && (PMAX == 4 || PMAX == 2), "square tiles: 4 or 2 staging pieces per wave and operand");
  // wave index as a scalar for the 4-wave tiles (LDS-DMA bases / M0 stay in SGPRs: +5-10 % on the pyramid-level GEMMs);
  // the 8-wave 256x256 tile measured 3 % slower with it, so it keeps the per-lane value
  const int tid = threadIdx.x, l = tid & 63;
  const int w = (WM * WN == 8 && !getenv_free_scalar_w) ? (tid >> 6) : __builtin_amdgcn_readfirstlane(tid >> 6);
  NT_PHASE(0);

  NtHeader P;
  GemmProb pr;
  int g, tm, tn;
  nt_fetch(P_arg, P, pr, g, blockIdx.x);
  nt_globalize(pr);
  nt_globalize(P);
  // conv -> BN -> ReLU launches: the launch generation (tag of this launch's statistics pairs), requested now, needed in the epilogue
  unsigned bn_gen_v = 0;
  if constexpr (BNF) bn_gen_v = (unsigned)__hip_atomic_load(P.counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  nt_locate<TM>(P, pr, tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const int M = pr.M, N = pr.N, K = pr.K, Cin = pr.Cin, taps = pr.taps;
  const int stride = pr.stride, pad = pr.pad, mode = pr.mode, Lsrc = pr.Lsrc;
  const T* __restrict__ Ag = (const T*)pr.A;
  const T* __restrict__ Bg = (const T*)pr.B;
  const T* zero = (const T*)g_zero_page;

  // ---- per-thread staging state: 4 A rows + 4 B rows (one 16-byte chunk each per K-step)
  const int pch = l & 7;
  const int div = mode ? stride : 1;
  const long lda = pr.lda;
  const int sh = div == 2 ? 1 : 0;
  const int tsgn = mode ? -1 : 1;
  int a_s[PA];           // mode 0: t*stride - pad ; mode 1: t + pad ; hugely negative when the row is out of range
  const T* pA[PA];       // A + (seq*Lsrc)*lda + lane chunk offset
  const T* pB[PB];       // B + n*ldb + lane chunk offset
  bool okb[PB];
  int a_base[PA];        // generic path
  long b_off[PB];
#pragma unroll
  for (int i = 0; i < PMAX; ++i) {
    const int row = (w * PMAX + i) * 8 + (l >> 3);
    const int m = m0 + row;
    const int coff = (pch ^ (((i & 1) << 2) + (l >> 4))) * CH;
    int seq = 0, t = -(1 << 28);
    if (m < M) {
      seq = m / pr.Lout;
      t = m - seq * pr.Lout;
    }
    a_s[i] = m < M ? (mode ? t + pad : t * stride - pad) : -(1 << 28);
    a_base[i] = seq * Lsrc;
    pA[i] = Ag + ((long)seq * Lsrc * lda + coff);
    const int n = n0 + row;
    okb[i] = n < N;
    b_off[i] = n < N ? (long)n * pr.ldb : -1;
    pB[i] = Bg + ((long)(n < N ? n : 0) * pr.ldb + coff);
  }

  // tiles are staged strictly in order; these advance by one tile per stage() call.  Split-K: this workgroup owns
  // K-tiles [kt_lo, nkt) of the problem.
  const int nkt_all = (K + BK - 1) / BK;
  const int kt_per = (nkt_all + P.ksplit - 1) / P.ksplit;
  const int kt_lo = (int)blockIdx.y * kt_per;
  const int nkt = min(nkt_all, kt_lo + kt_per);
  int s_kt = kt_lo, s_tap = 0, s_c0 = 0;
  if (FAST && kt_lo > 0) {
    s_tap = (kt_lo * BK) / Cin;
    s_c0 = kt_lo * BK - s_tap * Cin;
  }

  // Every thread issues exactly 8 global_load_lds per tile (the counted vmcnt below depends on it); masked lanes and
  // past-the-end tiles read the zero page.
  // piece(buf, i): the A and B loads of staging row-group i (2 of the 8 global_load_lds of a tile); advance(): next tile.
  auto piece = [&](int buf, int i) {
    char* As = smem + buf * STAGE_B;
    char* Bs = As + A_BYTES;
    if constexpr (FAST) {
      // one branch-free form for both modes (mode 0: sh = 0, div = 1): no control flow inside the MFMA stream
      const bool kin = s_kt < nkt;
      const long koff = (long)s_kt * BK;
      const int num = a_s[i] + tsgn * s_tap;
      const int st = num >> sh;
      const bool ok = kin & (num >= 0) & ((num & (div - 1)) == 0) & (st < Lsrc);
      const T* cand = pA[i] + ((long)st * lda + s_c0);
      const T* src = ok ? cand : zero;
      glds16(src, As + (w * PMAX + i) * 1024);
      const T* bcand = pB[i] + koff;
      const T* bsrc = (kin & okb[i]) ? bcand : zero;
      glds16(bsrc, Bs + (w * PMAX + i) * 1024);
    } else {
      const int h = i & 1;
      const int c = pch ^ ((h << 2) + (l >> 4));
      const int kk = s_kt * BK + c * CH;
      const int tp = taps == 1 ? 0 : kk / Cin;
      const int cc = kk - tp * Cin;
      const bool kin = kk < K;
      const int num = mode ? a_s[i] - tp : a_s[i] + tp;
      const int st = div == 1 ? num : (div == 2 ? num >> 1 : num / div);
      const bool ok = kin & (num >= 0) & (st * div == num) & (st < Lsrc);
      const long aoff = (long)(a_base[i] + st) * lda + cc;
      const T* src = ok ? Ag + aoff : zero;
      glds16(src, As + (w * PMAX + i) * 1024);
      const bool okb2 = kin & (b_off[i] >= 0);
      const T* bsrc = okb2 ? Bg + (b_off[i] + kk) : zero;
      glds16(bsrc, Bs + (w * PMAX + i) * 1024);
    }
  };
  auto advance = [&]() {
    if constexpr (FAST) {
      s_c0 += BK;
      if (s_c0 >= Cin) {
        s_c0 -= Cin;
        ++s_tap;
      }
    }
    ++s_kt;
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PMAX; ++i) piece(buf, i);
    advance();
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = w / WN, wc = w % WN;
  const int swz = (l >> 1) & 7;

  NT_PHASE(1);
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st) stage(st);
  // (wave-uniform: into an SGPR now -- the first tiles' loads are in flight behind it -- instead of a VGPR carried through the loop)
  const unsigned bn_gen = BNF ? (unsigned)__builtin_amdgcn_readfirstlane((int)bn_gen_v) : 0u;
  int cur = 0;
  for (int kt = kt_lo; kt < nkt; ++kt) {
    // each thread issues 8 loads per tile; tiles kt+1 .. kt+STAGES-2 may still be in flight
    NT_STAMP(0);
    if constexpr (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (STAGES == 4 && PMAX == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if constexpr (STAGES == 4 && PMAX == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    NT_STAMP(1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    NT_STAMP(2);
    if (kt == kt_lo) NT_PHASE(2);
    int nxt = cur + STAGES - 1;
    if (nxt >= STAGES) nxt -= STAGES;
    const char* As = smem + cur * STAGE_B;
    const char* Bs = As + A_BYTES;
    // The 8 loads of tile kt+STAGES-1 are issued in 4 pairs BETWEEN the MFMA groups of tile kt, so their issue cost
    // (~100 cycles each) overlaps the matrix pipe instead of preceding it.
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int pc = ((ks * 4 + (l >> 4)) ^ swz) * 16;
      typename Mma<T>::frag a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        a[mi] = *(const typename Mma<T>::frag*)(As + (wr * (MI * 16) + mi * 16 + (l & 15)) * 128 + pc);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        b[ni] = *(const typename Mma<T>::frag*)(Bs + (wc * (NI * 16) + ni * 16 + (l & 15)) * 128 + pc);
      if constexpr (PMAX == 4) piece(nxt, ks * 2); else piece(nxt, ks);    // 2-piece waves: one pair of loads per k-slice
      __builtin_amdgcn_sched_barrier(0);
#ifdef DRN_NT_TRACE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      NT_STAMP(3 + ks * 2);                    // fragments of this k-slice arrived (first piece issued)
      __builtin_amdgcn_sched_barrier(0);
#endif
      Mma<T>::template part<MI, NI, MI / 2>(a, b, acc, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PMAX == 4) piece(nxt, ks * 2 + 1);
      __builtin_amdgcn_sched_barrier(0);
      Mma<T>::template part<MI, NI, MI / 2>(a, b, acc, MI / 2);
      __builtin_amdgcn_sched_barrier(0);
      NT_STAMP(4 + ks * 2);                    // this k-slice's MFMAs and both pieces issued
    }
    advance();
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  NT_PHASE(3);

  if constexpr (BNF) {                   // conv -> BN -> ReLU in this launch (never split)
    nt_epilogue_bn<T, WM, WN, MI, NI, CHAIN>(P_arg, P.counters, bn_gen, pr, g, acc, smem, m0, n0, tm, tn);
    return;
  } else if constexpr (MI * NI == 8) {   // (the 128x128 tiles: split launches always use them)
    if (P.ksplit > 1) {
      // Split-K without a second launch: every split publishes its partial tile with write-through (sc1) 16-byte stores in
      // accumulator-native order (lane-contiguous: 1 KB per wave instruction), counts itself in, and the split that arrives
      // LAST re-reads all partials -- its own included, always in split order, so the sum does not depend on who was last --
      // and carries on into the epilogue.  Same exchange protocol as skinny_group_kernel (qdense.hip).
      constexpr int NT = 64 * NW;
      const int tile_id = blockIdx.x, ks = P.ksplit;
      f32x4* slab = (f32x4*)P.ws + ((long)tile_id * ks + blockIdx.y) * (MI * NI * NT) + tid;
      // ONE asm statement for the eight stores and their drain.  As separate statements the compiler recycled a store's
      // data registers for the next store's address straight after issuing it -- it cannot know the statement is a store, so
      // its hazard recogniser did not keep the wait states a > 64-bit VMEM store needs before its data VGPRs are
      // overwritten: rare corrupted partial tiles (scripts/stress_splitk.py).
      asm volatile(
          "global_store_dwordx4 %0, %8, off sc1\n\t"
          "global_store_dwordx4 %1, %9, off sc1\n\t"
          "global_store_dwordx4 %2, %10, off sc1\n\t"
          "global_store_dwordx4 %3, %11, off sc1\n\t"
          "global_store_dwordx4 %4, %12, off sc1\n\t"
          "global_store_dwordx4 %5, %13, off sc1\n\t"
          "global_store_dwordx4 %6, %14, off sc1\n\t"
          "global_store_dwordx4 %7, %15, off sc1\n\t"
          "s_waitcnt vmcnt(0)"
          :
          : "v"(slab), "v"(slab + NT), "v"(slab + 2 * NT), "v"(slab + 3 * NT), "v"(slab + 4 * NT), "v"(slab + 5 * NT),
            "v"(slab + 6 * NT), "v"(slab + 7 * NT), "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]), "v"(acc[2][0]),
            "v"(acc[2][1]), "v"(acc[3][0]), "v"(acc[3][1])
          : "memory");
      // DRN_XCHG_CONFIRM (launches that another queue's kernel may run beside; skinny_group_kernel, qdense.hip, says why: a
      // write-through store's completion is not its visibility to the other XCDs): one returning agent-scope OR-with-zero per
      // 64-byte request of every store (lanes 0, 4, 8, ...: four lanes share a request) -- a read-modify-write of the same address
      // is performed behind the store -- before the workgroup counts itself in.  Issued right behind the stores or after their
      // wait costs the same ~6 us per launch: it is the atomics' rate, not their latency.
      if ((P.confirm & DRN_XCHG_READBACK) && (tid & 3) == 0) {          // the cheaper variant: sc1 loads instead of read-modify-writes
        unsigned b0, b1, b2, b3, b4, b5, b6, b7;
        asm volatile(
            "global_load_dword %0, %8, off sc1\n\t"
            "global_load_dword %1, %9, off sc1\n\t"
            "global_load_dword %2, %10, off sc1\n\t"
            "global_load_dword %3, %11, off sc1\n\t"
            "global_load_dword %4, %12, off sc1\n\t"
            "global_load_dword %5, %13, off sc1\n\t"
            "global_load_dword %6, %14, off sc1\n\t"
            "global_load_dword %7, %15, off sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "=&v"(b7)
            : "v"(slab), "v"(slab + NT), "v"(slab + 2 * NT), "v"(slab + 3 * NT), "v"(slab + 4 * NT), "v"(slab + 5 * NT),
              "v"(slab + 6 * NT), "v"(slab + 7 * NT)
            : "memory");
      }
      if ((P.confirm & DRN_XCHG_CONFIRM) && (tid & 3) == 0) {
        unsigned b0, b1, b2, b3, b4, b5, b6, b7;
        asm volatile(
            "global_atomic_or %0, %8, %16, off sc0 sc1\n\t"
            "global_atomic_or %1, %9, %16, off sc0 sc1\n\t"
            "global_atomic_or %2, %10, %16, off sc0 sc1\n\t"
            "global_atomic_or %3, %11, %16, off sc0 sc1\n\t"
            "global_atomic_or %4, %12, %16, off sc0 sc1\n\t"
            "global_atomic_or %5, %13, %16, off sc0 sc1\n\t"
            "global_atomic_or %6, %14, %16, off sc0 sc1\n\t"
            "global_atomic_or %7, %15, %16, off sc0 sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "=&v"(b7)
            : "v"(slab), "v"(slab + NT), "v"(slab + 2 * NT), "v"(slab + 3 * NT), "v"(slab + 4 * NT), "v"(slab + 5 * NT),
              "v"(slab + 6 * NT), "v"(slab + 7 * NT), "v"(0u)
            : "memory");
      }
      static_assert(MI == 4 && NI == 2, "the exchange is written for the 8-wave 128x128 tile");
      __syncthreads();
      int& s_last = *(int*)smem;            // (the ring is idle: everybody is past the main loop's last LDS read)
      if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(P.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == ks - 1;
        if (prev == ks - 1) __hip_atomic_store(P.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
      }
      __syncthreads();
      if (!s_last) return;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4* src = (const f32x4*)P.ws + (long)tile_id * ks * (MI * NI * NT) + tid;
      for (int q = 0; q < ks; ++q, src += MI * NI * NT) {
        // ONE asm statement for the eight loads AND their wait: with a separate wait the compiler, which takes an asm output
        // as ready when its statement ends, may move a loaded value to another register before the data has landed
        // (seen as rare garbage tiles).  Early-clobber outputs: no output may share registers with a later load's address.
        f32x4 p0, p1, p2, p3, p4, p5, p6, p7;
        asm volatile(
            "global_load_dwordx4 %0, %8, off sc1\n\t"
            "global_load_dwordx4 %1, %9, off sc1\n\t"
            "global_load_dwordx4 %2, %10, off sc1\n\t"
            "global_load_dwordx4 %3, %11, off sc1\n\t"
            "global_load_dwordx4 %4, %12, off sc1\n\t"
            "global_load_dwordx4 %5, %13, off sc1\n\t"
            "global_load_dwordx4 %6, %14, off sc1\n\t"
            "global_load_dwordx4 %7, %15, off sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7)
            : "v"(src), "v"(src + NT), "v"(src + 2 * NT), "v"(src + 3 * NT), "v"(src + 4 * NT), "v"(src + 5 * NT), "v"(src + 6 * NT),
              "v"(src + 7 * NT)
            : "memory");
        const f32x4 part[8] = {p0, p1, p2, p3, p4, p5, p6, p7};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += part[mi * NI + ni];
      }
      __syncthreads();
    }
  }
  nt_epilogue<T, WM, WN, MI, NI>(P, pr, acc, smem, m0, n0, tm);
  NT_PHASE(4);
}
