// Dense layers of the query side (model/language_module.py:13-23,38-63; model/main_model.py:36-50), fp32 throughout.
//
// Every product on this side has a "batch" dimension of 32 clips (or 32 clips x <= 8 words for the LSTM input projection):
//   forward / input gradients   Y[M][N]  = X[M][K] * W[N][K]^T (+ bias)(ReLU)(* mask)      M <= 64 rows per problem
//   weight / bias gradients     dW[N][K] = sum_m dY[m][N] * X[m][K],  db[N] = sum_m dY[m][N]  M <= a few hundred rows
// i.e. weight-streaming problems with a tiny reduction or a tiny row count: a library GEMM puts 16-32 workgroups on each and
// takes 5-25 us, and there are ~25 of them per step.  Here they are GROUPED (one launch serves up to DRN_QD_MAX problems)
// and run on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) with operands straight from global memory (everything is L2
// resident).  Long-K problems are split over workgroups; the last-arriving workgroup of an output tile adds the partial
// tiles in a fixed order (deterministic) and applies the epilogue -- no second launch.
#include "common.h"
#include "../../include/drn_hip.h"

#define QD_THREADS 256

// ---------------------------------------------------------------------------------------------------------------------
// grouped skinny NT product
// ---------------------------------------------------------------------------------------------------------------------
struct SkProb {
  const float* X;      // fp32, or (x_bf16) bf16 rows -- then the weights are rounded to bf16 in registers and the product runs on
                       // v_mfma_f32_16x16x32_bf16 (the bf16 model's LSTM input gradient reads the gate gradients' bf16 copy: the X
                       // rows, re-read by every column tile, are most of this kernel's bytes)
  const float* W;
  const float* bias;
  const float* mask;   // optional [M][ldm]: output zeroed where mask <= 0 (ReLU backward of the layer in front)
  float* Y;
  float* part;         // [ksplit][M][N] partial tiles (ksplit > 1)
  int* counters;       // [ceil(N/16)] arrival counters of this problem (self-resetting)
  int ldx, ldy, ldm, M, N, K;
  int kslice, ksplit, relu;
  int blk0;            // first workgroup of this problem; a problem owns ceil(N/16) * ksplit workgroups
  int x_bf16;
};
struct SkGroupArgs {
  SkProb p[DRN_QD_MAX];
  int n;
};

// These are weight-streaming problems whose weights come from HBM (Adam rewrote them ~2 ms earlier): what matters is that
// the whole matrix is requested at once, i.e. >= ~256 workgroups with every operand of a wave's K range in flight.  So a
// long K is split over workgroups; the LAST-ARRIVING workgroup of a column tile adds the partial tiles in a fixed order
// (deterministic) and applies the epilogue -- no second launch.  The partial tiles are exchanged with agent-scope relaxed
// atomic stores / loads (write-through / L2-bypassing accesses): ordinary stores + __threadfence() cost 25 us per launch
// here, because a device-scope release writes back the whole L2 of the XCD.  Every store is confirmed by a returning atomic
// on its own address before the workgroup counts itself in (see below: the store's completion alone is not enough).
// Optional per-workgroup timeline (build with -DDRN_QD_TRACE, scripts/experiments/qd_trace.py): wall_clock64() (100 MHz) at
// entry / after the K loop / after the cross-wave sum / at exit, per workgroup, read back with drn_debug_qd_trace().
// What it showed (round 3, gate projections 32 x 1024 -> 4864, 304 workgroups): 2.5 us per 64-wide K step with 0.4 us of MFMAs
// in it, whatever the prefetch depth (KU = 2 .. 16 measured equal) -- a workgroup pulls 48 KB per step through its L1 at
// ~24 GB/s.  The L2s start every kernel cold (cross-XCD coherence is kept by write-back + invalidate at kernel boundaries), so
// "hot" operands come from the Infinity Cache, and every workgroup re-reads all of X.  A 64-column x K-slice variant that
// stages X and W once per workgroup in LDS with full-line loads (scripts/experiments/qdense_wide_kernel.hip.txt, bit-compatible,
// tested) moved half the bytes and was SLOWER (gates 24 vs 15 us, LSTM input projection 34 vs 17): its 100 KB of LDS leave one
// workgroup per CU (two rounds), staging 96 KB took 5.5 us before the first MFMA, and the slices meet through a 3 us exchange.
#ifdef DRN_QD_TRACE
__device__ long long g_qd_trace[8192 * 4];
#define QD_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_qd_trace[blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
extern "C" int drn_debug_qd_trace(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qd_trace), (size_t)n * 8);
}
#else
#define QD_STAMP(i) do { } while (0)
#endif

template <int NBT, bool XB16>
__global__ __launch_bounds__(QD_THREADS) void skinny_group_kernel(const SkGroupArgs G) {
  __shared__ float red[4][NBT][64][4];
  __shared__ int is_last;
  QD_STAMP(0);
  int g = 0;
#pragma unroll
  for (int i = 1; i < DRN_QD_MAX; ++i)
    if (i < G.n && (int)blockIdx.x >= G.p[i].blk0) g = i;
  const SkProb& P = G.p[g];
  const int rel = blockIdx.x - P.blk0;
  const int ks = rel % P.ksplit, tile = rel / P.ksplit;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int n0 = tile * 16;
  const int row = l & 15, kc = (l >> 4) * 4;
  const int kq = P.kslice >> 2;                          // per wave, a multiple of 16
  const int kbeg = ks * P.kslice + w * kq;
  const int kend = min(kbeg + kq, P.K);
  f32x4 acc[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) acc[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool nok = n0 + row < P.N;
  const float* wrow = P.W + (long)(n0 + row) * P.K;
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int KU = 4;                                  // 16-wide (bf16 rows: 32-wide) K steps whose operands are all requested up front
  if constexpr (XB16) {
    const bf16_t* X16 = (const bf16_t*)P.X;
    const int kc8 = (l >> 4) * 8;
    for (int k0 = kbeg; k0 < kend; k0 += 32 * KU) {      // (kslice is a multiple of 128 here: a wave's range is a multiple of 32)
      bf16x8 a[KU][NBT];
      f32x4 blo[KU], bhi[KU];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int k = k0 + u * 32 + kc8;
        const bool kok = k < kend;                       // K % 8 == 0: an 8-element piece is in or out as a whole
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) {
          const int m = bt * 16 + row;
          a[u][bt] = *(const bf16x8*)(X16 + (long)(m < P.M ? m : 0) * P.ldx + (kok ? k : 0));       // (masked loads read a valid address)
          if (!(kok && m < P.M))
#pragma unroll
            for (int e = 0; e < 8; ++e) a[u][bt][e] = (bf16_t)0.f;
        }
        const float* wp = wrow + (kok ? k : 0);
        blo[u] = *(const f32x4*)(nok ? wp : P.W);
        bhi[u] = *(const f32x4*)(nok ? wp + 4 : P.W);
        if (!(kok && nok)) { blo[u] = zero4; bhi[u] = zero4; }
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (k0 + u * 32 >= kend) break;                  // wave-uniform
        bf16x8 bv;
#pragma unroll
        for (int e = 0; e < 4; ++e) { bv[e] = (bf16_t)blo[u][e]; bv[4 + e] = (bf16_t)bhi[u][e]; }
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) acc[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][bt], bv, acc[bt], 0, 0, 0);
      }
    }
  } else
  for (int k0 = kbeg; k0 < kend; k0 += 16 * KU) {
    f32x4 a[KU][NBT], b[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int k = k0 + u * 16 + kc;
      const bool kok = k < kend;                         // K % 4 == 0 and kc % 4 == 0: a quad is in or out as a whole
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) {
        const int m = bt * 16 + row;
        a[u][bt] = (kok && m < P.M) ? *(const f32x4*)(P.X + (long)m * P.ldx + k) : zero4;
      }
      b[u] = (kok && nok) ? *(const f32x4*)(wrow + k) : zero4;
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (k0 + u * 16 >= kend) break;                    // wave-uniform: no MFMAs on an all-zero (masked) K step
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][bt][e], b[u][e], acc[bt], 0, 0, 0);
    }
  }
  QD_STAMP(1);
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][bt][l][r] = acc[bt][r];
  __syncthreads();
  QD_STAMP(2);
  // D layout: m = bt*16 + (l>>4)*4 + r, n = n0 + (l&15); wave w finishes register r = w of every lane
  const int r = w, n = n0 + (l & 15);
  float v[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) v[bt] = red[0][bt][l][r] + red[1][bt][l][r] + red[2][bt][l][r] + red[3][bt][l][r];
  if (P.ksplit > 1) {
    // partial tiles are exchanged in the lanes' own order: a lane's NBT values (one per 16-row block) are ONE 4 / 8 / 16-byte
    // write-through store and one load per split, not NBT four-byte ones (a scalar sc1 access is a fabric transaction of its own)
    const int ntiles = (P.N + 15) >> 4;
    float* mine = P.part + (((long)ks * ntiles + tile) * QD_THREADS + threadIdx.x) * NBT;
    if constexpr (NBT == 4) {
      const f32x4 pv = (f32x4){v[0], v[1], v[2], v[3]};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(mine), "v"(pv) : "memory");
    } else if constexpr (NBT == 2) {
      const f32x2 pv = (f32x2){v[0], v[1]};
      asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(mine), "v"(pv) : "memory");
    } else {
      __hip_atomic_store(mine, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ... and CONFIRMED: s_waitcnt vmcnt(0) after a write-through store does not mean the data has reached the point the other
    // XCDs read from -- with another queue's bandwidth-bound kernel running beside this one the ticket (a different address, a
    // different channel) overtook a partial about once in 10^5 launches and the last arriver summed the previous launch's value
    // (the two-branch hipGraph step; scripts/experiments/forked_race_hunt.py: 12 events in 180 k replays, none in 240 k with this).
    // A returning agent-scope read-modify-write of the SAME address is performed behind the store at the serialisation point.
    {
      unsigned back;
      asm volatile("global_atomic_or %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(back) : "v"(mine), "v"(0u) : "memory");
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's write-through stores are complete ...
    __syncthreads();                                          // ... and so are the other waves' before thread 0 counts us in
    if (threadIdx.x == 0) {
      const int prev = __hip_atomic_fetch_add(P.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = prev == P.ksplit - 1;
      if (is_last) __hip_atomic_store(P.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
    }
    __syncthreads();
    if (!is_last) return;
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) v[bt] = 0.f;
    // fixed order: deterministic whoever arrives last.  ksplit is a power of two >= 2 (qd_ksplit): four (or the two) partial
    // tiles per trip are requested together -- one asm statement per trip with its own wait (see gemm_nt.hip on why)
    const long qstride = (long)ntiles * QD_THREADS * NBT;
    const float* src = P.part + ((long)tile * QD_THREADS + threadIdx.x) * NBT;
    if constexpr (NBT == 4) {
      for (int q = 0; q < P.ksplit; q += 4) {
        f32x4 p0, p1, p2 = (f32x4){0.f, 0.f, 0.f, 0.f}, p3 = p2;
        if (P.ksplit >= 4)
          asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                       "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
                       : "v"(src + q * qstride), "v"(src + (q + 1) * qstride), "v"(src + (q + 2) * qstride), "v"(src + (q + 3) * qstride)
                       : "memory");
        else
          asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(p0), "=&v"(p1)
                       : "v"(src + q * qstride), "v"(src + (q + 1) * qstride)
                       : "memory");
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) v[bt] = (((v[bt] + p0[bt]) + p1[bt]) + p2[bt]) + p3[bt];
      }
    } else if constexpr (NBT == 2) {
      for (int q = 0; q < P.ksplit; q += 4) {
        f32x2 p0, p1, p2 = (f32x2){0.f, 0.f}, p3 = p2;
        if (P.ksplit >= 4)
          asm volatile("global_load_dwordx2 %0, %4, off sc1\n\tglobal_load_dwordx2 %1, %5, off sc1\n\t"
                       "global_load_dwordx2 %2, %6, off sc1\n\tglobal_load_dwordx2 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
                       : "v"(src + q * qstride), "v"(src + (q + 1) * qstride), "v"(src + (q + 2) * qstride), "v"(src + (q + 3) * qstride)
                       : "memory");
        else
          asm volatile("global_load_dwordx2 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(p0), "=&v"(p1)
                       : "v"(src + q * qstride), "v"(src + (q + 1) * qstride)
                       : "memory");
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) v[bt] = (((v[bt] + p0[bt]) + p1[bt]) + p2[bt]) + p3[bt];
      }
    } else {
      for (int q = 0; q < P.ksplit; ++q) v[0] += __hip_atomic_load(src + q * qstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) {
    const int m = bt * 16 + (l >> 4) * 4 + r;
    if (m >= P.M || n >= P.N) continue;
    float o = v[bt];
    if (P.bias) o += P.bias[n];
    if (P.relu) o = fmaxf(o, 0.f);
    if (P.mask && !(P.mask[(long)m * P.ldm + n] > 0.f)) o = 0.f;
    P.Y[(long)m * P.ldy + n] = o;
  }
  QD_STAMP(3);
}

// K >= 1024 is cut into slices of >= 256 until ~256 workgroups work on the long-K problems of the launch together
// (`tiles_long` = their column tiles): fewer, longer slices when the group already has many tiles (less partial-tile traffic).
static int qd_ksplit(int tiles_long, int K) {
  if (K < 1024) return 1;
  int ks = 1;
  const int tgt = drn_tuning(DRN_TUNE_EXP0 + 4) > 0 ? drn_tuning(DRN_TUNE_EXP0 + 4) : 256;      // (exp4: experiment override)
  while (ks < 16 && tiles_long * ks < tgt && K / (2 * ks) >= 256) ks *= 2;
  return ks;
}
static int qd_tiles_long(const DrnSkinnyDesc* d, int n) {
  int t = 0;
  for (int i = 0; i < n; ++i)
    if (d[i].K >= 1024) t += cdiv(d[i].N, 16);
  return t;
}

extern "C" int64_t drn_skinny_group_ws_elems(const DrnSkinnyDesc* d, int n) {
  int64_t tot = 0;
  const int tl = qd_tiles_long(d, n);
  for (int i = 0; i < n; ++i) {
    const int ks = qd_ksplit(tl, d[i].K);
    if (ks > 1) tot += (int64_t)ks * cdiv(d[i].N, 16) * QD_THREADS * 4;      // lane-ordered partial tiles, up to 4 row blocks
  }
  return tot;
}

extern "C" int drn_skinny_group(const DrnSkinnyDesc* d, int n, float* ws, int32_t* counters, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(d && n >= 1 && n <= DRN_QD_MAX, "drn_skinny_group: 1..%d problems", DRN_QD_MAX);
  SkGroupArgs G;
  memset(&G, 0, sizeof(G));
  G.n = n;
  int blocks = 0, nbt = 1, cnt = 0;
  int64_t wsoff = 0;
  const int tiles_long = qd_tiles_long(d, n);
  for (int i = 0; i < n; ++i) {
    const DrnSkinnyDesc& s = d[i];
    DRN_CHECK_ARG(s.X && s.W && s.Y && s.M > 0 && s.M <= 64 && s.N > 0 && s.K > 0, "drn_skinny_group: problem %d: bad shape (M <= 64)", i);
    DRN_CHECK_ARG(s.K % 4 == 0 && (s.ldx % 4 == 0 || s.x_dtype == DRN_BF16) && (((uintptr_t)s.X | (uintptr_t)s.W) & 15) == 0,
                  "drn_skinny_group: problem %d: need K %% 4 == 0, ldx %% 4 == 0, 16-byte aligned X / W", i);
    SkProb& P = G.p[i];
    P.X = (const float*)s.X; P.W = s.W; P.bias = s.bias; P.mask = s.mask; P.Y = s.Y;
    P.x_bf16 = s.x_dtype == DRN_BF16;
    DRN_CHECK_ARG(s.x_dtype == DRN_F32 || s.x_dtype == DRN_BF16, "drn_skinny_group: problem %d: bad x_dtype", i);
    DRN_CHECK_ARG(s.x_dtype == d[0].x_dtype, "drn_skinny_group: the problems of a launch share one x_dtype");
    DRN_CHECK_ARG(s.x_dtype == DRN_F32 || (s.K % 8 == 0 && s.ldx % 8 == 0), "drn_skinny_group: problem %d: bf16 rows need K %% 8 == 0, ldx %% 8 == 0", i);
    P.ldx = s.ldx; P.ldy = s.ldy; P.ldm = s.ldm; P.M = s.M; P.N = s.N; P.K = s.K; P.relu = s.relu;
    P.ksplit = qd_ksplit(tiles_long, s.K);
    P.kslice = cdiv(cdiv(s.K, P.ksplit), 64) * 64;       // 4 waves x 16-wide steps
    if (P.x_bf16) P.kslice = cdiv(P.kslice, 128) * 128;  // ... x 32-wide steps
    P.blk0 = blocks;
    blocks += cdiv(s.N, 16) * P.ksplit;
    if (P.ksplit > 1) {
      DRN_CHECK_ARG(ws && counters, "drn_skinny_group: problem %d is K-split: workspace (drn_skinny_group_ws_elems) and counters required", i);
      P.part = ws + wsoff;
      wsoff += (int64_t)P.ksplit * cdiv(s.N, 16) * QD_THREADS * 4;
      P.counters = counters + cnt;
      cnt += cdiv(s.N, 16);
      DRN_CHECK_ARG(cnt <= DRN_QD_COUNTERS, "drn_skinny_group: more than %d K-split column tiles", DRN_QD_COUNTERS);
    }
    nbt = nbt > cdiv(s.M, 16) ? nbt : cdiv(s.M, 16);
  }
  hipStream_t st = (hipStream_t)stream;
  if (d[0].x_dtype == DRN_BF16) {
    if (nbt == 1) skinny_group_kernel<1, true><<<blocks, QD_THREADS, 0, st>>>(G);
    else if (nbt == 2) skinny_group_kernel<2, true><<<blocks, QD_THREADS, 0, st>>>(G);
    else skinny_group_kernel<4, true><<<blocks, QD_THREADS, 0, st>>>(G);
  } else if (nbt == 1) skinny_group_kernel<1, false><<<blocks, QD_THREADS, 0, st>>>(G);
  else if (nbt == 2) skinny_group_kernel<2, false><<<blocks, QD_THREADS, 0, st>>>(G);
  else skinny_group_kernel<4, false><<<blocks, QD_THREADS, 0, st>>>(G);
  return drn_launch_status("drn_skinny_group");
}

// ---------------------------------------------------------------------------------------------------------------------
// grouped weight / bias gradients of batch-sized layers: dW[N][K] = dY^T X over M rows, db[N] = column sums of dY
// ---------------------------------------------------------------------------------------------------------------------
struct OwProb {
  const float* dY;
  const float* X;
  float* dW;
  float* db;           // optional
  float* db2;          // optional second destination of the same sums (LSTM: b_ih and b_hh share their gradient)
  int ldy, ldx, ldw, M, N, K;
  int ntk;             // workgroup tiles along K (64 wide)
  int blk0;
};
struct OwGroupArgs {
  OwProb p[DRN_QD_MAX];
  int n;
};

// workgroup: 64 rows of dW (n) x 64 columns (k); wave w owns 16 of the rows as four 16x16 MFMA tiles.  The M rows are
// consumed 32 at a time: the 32 x 64 slabs of dY and X are fetched with coalesced 16-byte loads (256 B per row), parked in
// LDS (row pitch 80 floats: the four row groups of an MFMA operand read land on disjoint bank quarters) and the next
// slab's loads are already in flight while the current one feeds the MFMAs -- operand loads straight in MFMA layout (4 B per
// lane, every panel re-read by each wave) held this kernel to ~1.2 TB/s of L2 traffic.
// LOWP (the bf16 model): the staged fp32 slabs are rounded to bf16 as the MFMA operands are read and multiplied on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- 4 MFMAs per 32-row slab and wave instead of 32 exact-fp32 ones (the fp32
// kernel is MFMA-bound on the LSTM weights, M = clips x words = 256 rows: ~14 of its 37 us); bias gradients stay exact fp32 sums.
#define OW_MB 32
#define OW_PITCH 80
template <bool LOWP>
__global__ __launch_bounds__(QD_THREADS) void outer_wgrad_kernel(const OwGroupArgs G) {
  __shared__ __attribute__((aligned(16))) float sm[2][OW_MB][OW_PITCH];   // one block: the epilogue re-uses all of it
  float (*Ys)[OW_PITCH] = sm[0], (*Xs)[OW_PITCH] = sm[1];
  int g = 0;
#pragma unroll
  for (int i = 1; i < DRN_QD_MAX; ++i)
    if (i < G.n && (int)blockIdx.x >= G.p[i].blk0) g = i;
  const OwProb& P = G.p[g];
  const int rel = blockIdx.x - P.blk0;
  const int tk = rel % P.ntk, tn = rel / P.ntk;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int nb = tn * 64, k0 = tk * 64;
  const int n0 = nb + w * 16;
  const int li = l & 15, lq = l >> 4;                    // A: (i = n, kk = m) ; B: (kk = m, j = k)
  const bool nok = n0 + li < P.N;
  const bool want_db = (P.db != nullptr) && tk == 0;
  float bsum = 0.f;
  if (P.dW == nullptr) {                                 // column sums only
    if (!want_db) return;
    for (int m = lq; m < P.M; m += 4) bsum += nok ? P.dY[(long)m * P.ldy + n0 + li] : 0.f;
  } else {
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging map: thread t loads rows (t >> 4) and (t >> 4) + 16 of the slab, columns 4 * (t & 15) .. + 3 of both panels
    const int sr = threadIdx.x >> 4, sc = (threadIdx.x & 15) * 4;
    const bool yvec = (P.ldy % 4 == 0) && ((((uintptr_t)P.dY) & 15) == 0) && (nb + sc + 3 < P.N);
    const bool xvec = (P.ldx % 4 == 0) && ((((uintptr_t)P.X) & 15) == 0) && (k0 + sc + 3 < P.K);
    auto fetch = [&](const float* base, int ld, int m, int c, int cmax, bool vec) -> f32x4 {
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (m < P.M) {
        const float* p = base + (long)m * ld + c;
        if (vec) v = *(const f32x4*)p;
        else
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < cmax) v[e] = p[e];
      }
      return v;
    };
    f32x4 ry[2], rx[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ry[h] = fetch(P.dY, P.ldy, sr + 16 * h, nb + sc, P.N, yvec);
      rx[h] = fetch(P.X, P.ldx, sr + 16 * h, k0 + sc, P.K, xvec);
    }
    for (int m0 = 0; m0 < P.M; m0 += OW_MB) {
      __syncthreads();                                   // the previous slab's MFMA operand reads are done
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *(f32x4*)&Ys[sr + 16 * h][sc] = ry[h];
        *(f32x4*)&Xs[sr + 16 * h][sc] = rx[h];
      }
      __syncthreads();
      if (m0 + OW_MB < P.M) {                            // next slab in flight while this one is multiplied
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          ry[h] = fetch(P.dY, P.ldy, m0 + OW_MB + sr + 16 * h, nb + sc, P.N, yvec);
          rx[h] = fetch(P.X, P.ldx, m0 + OW_MB + sr + 16 * h, k0 + sc, P.K, xvec);
        }
      }
      if constexpr (LOWP) {
        // lane group lq takes rows lq, lq + 4, ..., lq + 28 of the slab as its 8 k values (any assignment of k to lanes is fine as
        // long as A and B agree; this one keeps the four groups on different LDS banks)
        bf16x8 a8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = Ys[e * 4 + lq][w * 16 + li];
          bsum += a;
          a8[e] = (bf16_t)a;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x8 b8;
#pragma unroll
          for (int e = 0; e < 8; ++e) b8[e] = (bf16_t)Xs[e * 4 + lq][j * 16 + li];
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[j], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int u = 0; u < OW_MB / 4; ++u) {
          const float a = Ys[u * 4 + lq][w * 16 + li];
          bsum += a;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Xs[u * 4 + lq][j * 16 + li], acc[j], 0, 0, 0);
        }
      }
    }
    // D: n = n0 + lq*4 + r, k = k0 + j*16 + li.  Each wave parks its 16 x 64 tile in LDS (the slabs are spent) and writes it
    // out as 16-byte pieces of whole 256-byte row segments: 4 store instructions per wave instead of 16 four-byte ones
    // that touched a 64-byte piece of four different rows each (the launch writes 26 MB: it is store-bound).
    __syncthreads();                                     // every wave is done reading the last slab
    float* wt = &sm[0][0][0] + w * (16 * 68);               // 4 waves x 16 rows x pitch 68 floats = 17 KB of the 20 KB
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) wt[(lq * 4 + r) * 68 + j * 16 + li] = acc[j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a wave reads back only what it wrote itself)
    const bool ovec = (P.ldw % 4 == 0) && ((((uintptr_t)P.dW) & 15) == 0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = it * 4 + (l >> 4), cc = (l & 15) * 4;
      const int n = n0 + rr, k = k0 + cc;
      if (n < P.N && k < P.K) {
        const f32x4 v = *(const f32x4*)(wt + rr * 68 + cc);
        float* dst = P.dW + (long)n * P.ldw + k;
        if (ovec && k + 3 < P.K) *(f32x4*)dst = v;
        else
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < P.K) dst[e] = v[e];
      }
    }
  }
  if (want_db) {
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lq == 0 && nok) {
      P.db[n0 + li] = bsum;
      if (P.db2) P.db2[n0 + li] = bsum;
    }
  }
}

extern "C" int drn_outer_wgrad(const DrnOuterDesc* d, int n, int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(d && n >= 1 && n <= DRN_QD_MAX, "drn_outer_wgrad: 1..%d problems", DRN_QD_MAX);
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "drn_outer_wgrad: bad dtype %d", dtype);
  OwGroupArgs G;
  memset(&G, 0, sizeof(G));
  G.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const DrnOuterDesc& s = d[i];
    DRN_CHECK_ARG(s.dY && s.M > 0 && s.N > 0 && ((s.dW && s.X && s.K > 0) || (!s.dW && s.db)), "drn_outer_wgrad: problem %d: bad args", i);
    OwProb& P = G.p[i];
    P.dY = s.dY; P.X = s.X; P.dW = s.dW; P.db = s.db; P.db2 = s.db2;
    P.ldy = s.ldy; P.ldx = s.ldx; P.ldw = s.ldw; P.M = s.M; P.N = s.N; P.K = s.K;
    P.ntk = s.dW ? cdiv(s.K, 64) : 1;
    P.blk0 = blocks;
    blocks += cdiv(s.N, 64) * P.ntk;
  }
  if (dtype == DRN_BF16) outer_wgrad_kernel<true><<<blocks, QD_THREADS, 0, (hipStream_t)stream>>>(G);
  else outer_wgrad_kernel<false><<<blocks, QD_THREADS, 0, (hipStream_t)stream>>>(G);
  return drn_launch_status("drn_outer_wgrad");
}
