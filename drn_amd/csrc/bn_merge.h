// Train-mode BatchNorm statistics merge, shared by the stand-alone pass (bn.hip: bn_train_apply_kernel, 64 channels x 4 slab
// lanes = 256 threads) and by the conv -> BN -> ReLU kernel (gemm_nt_kernel.h: nt_epilogue_bn, 128 channels x 4 slab lanes = 512
// threads): ONE copy of the arithmetic, contraction off, so that both paths -- and every workgroup of either -- produce the same
// bits (nn.BatchNorm1d in training, model/basic_blocks.py:23-26; per-level-call statistics, model/fcos.py:93-102).
#pragma once
#include "common.h"

// Where the slab statistics come from:
//   BN_ST_PLAIN     float stats[(slab*2 + which)*C + c] written by an EARLIER launch (drn_gemm_nt -> drn_bn_train_apply);
//   BN_ST_TAGGED    64-bit {float value, uint32 tag} pairs at the same indices, written by OTHER workgroups of the SAME launch
//                   with ONE 8-byte write-through store each (conv -> BN -> ReLU in one launch, gemm_nt_kernel.h).  A pair is
//                   valid when its tag equals `want`, the launch's generation: readers poll the data itself -- no counter, no
//                   read-modify-write, no flag that would need clearing -- so a hand-off costs one store and one load.
enum { BN_ST_PLAIN = 0, BN_ST_TAGGED = 2 };
struct BnTagged {
  unsigned want;        // tag of this launch
  int* timeouts;        // watchdog counter: a reader that saw no valid pair for 2 s gives up and counts itself here
};
__device__ __forceinline__ unsigned long long bn_tag_pack(float v, unsigned tag) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}
__device__ __forceinline__ unsigned long long bn_ld_pair(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // global_load_dwordx2 ... sc1
}
// (Reading the pairs through this XCD's L2 once bn_wait_slabs() has seen the writers -- ~14 workgroups per XCD merge the same
// column -- was tried: merge phase 4.7 -> 12 us.  The coherent polling loads leave their lines in L2 as they were when
// polled, later plain loads hit those and fail the tag check, and the coherent retry comes on top.)
__device__ __forceinline__ bool bn_wait_expired(const long long t0, int* timeouts) {
  __builtin_amdgcn_s_sleep(32);
  if (wall_clock64() - t0 <= 200000000LL) return false;                            // 2 s of the 100 MHz wall clock
  __hip_atomic_fetch_add(timeouts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// Quiet wait for ALL slabs of a tile column (BN_ST_TAGGED): executed by ONE wave of the workgroup (the others park at the
// barrier that follows), lane i looks at the M2 pair of channel `c_first` of slabs i, i + 64, ... and re-reads only while ITS
// slab is missing, sleeping ~1 us between looks -- a few loads per microsecond and workgroup.  (Letting all 512 threads spin
// on the 32 pairs each of them merges put ~30 TB/s of L2-bypassing loads on the fabric as soon as the first workgroups
// waited: the workgroups still computing starved behind them and launches ran into the 2 s watchdog.)  The merge that follows
// still verifies every pair it uses -- the other channels of a slab land within the same microsecond -- and re-reads if not.
__device__ __forceinline__ void bn_wait_slabs(const void* st_, const int slabs, const int C, const int c_first, const BnTagged tg) {
  const unsigned long long* p = (const unsigned long long*)st_ + c_first;
  const int lane = threadIdx.x & 63;
  const long long t0 = wall_clock64();
  for (int s0 = 0; s0 < slabs; s0 += 64) {
    const int s = s0 + lane;
    bool ready = s >= slabs;
    for (;;) {
      if (!ready) ready = (unsigned)(bn_ld_pair(p + ((long)s * 2 + 1) * C) >> 32) == tg.want;
      if (__builtin_amdgcn_ballot_w64(!ready) == 0) break;
      __builtin_amdgcn_s_sleep(12);
      if (wall_clock64() - t0 > 200000000LL) {                                          // 2 s of the 100 MHz wall clock
        if (!ready) __hip_atomic_fetch_add(tg.timeouts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
}

// (mean, biased variance) of channel cbase + (tid % CB) from the GEMM epilogue's per-128-row-slab (sum, M2) pairs, merged in
// double with the parallel-variance formula (Chan et al.).  4*CB threads = CB channels x 4 slab lanes; the lanes meet in LDS
// and are added in a fixed order.  Up to 64 slabs (8192 rows) a thread's pairs are loaded once, all in flight together, and
// kept in registers for the second pass; longer problems re-read them (L2 hits).  Ends with a workgroup barrier (shd is free).
template <int CB, int MODE, int TB = 8>
__device__ __forceinline__ void bn_merge_cols(const void* __restrict__ st_, const int tiles, const int Mrows, const int C,
                                              const int cbase, double (*shd)[CB], double& mean_out, double& var_out,
                                              const BnTagged tg = BnTagged{0u, nullptr}) {
#pragma clang fp contract(off)
  constexpr int KMAX = 16;
  const int ci = threadIdx.x % CB, j = threadIdx.x / CB;
  // wave-uniform base + 32-bit element offsets (host-checked to fit): scalar-base addressing, no 64-bit address per load
  const float* __restrict__ p = (const float*)st_ + cbase;                                 // BN_ST_PLAIN
  const unsigned long long* __restrict__ pt = (const unsigned long long*)st_ + cbase;      // BN_ST_TAGGED
  auto at = [&](int k, int which) -> unsigned { return (unsigned)(k * 2 + which) * (unsigned)C + (unsigned)ci; };
  auto ld = [&](int k, int which) -> float {       // a value known to be valid
    if constexpr (MODE == BN_ST_TAGGED) return __uint_as_float((unsigned)bn_ld_pair(pt + at(k, which)));
    else return p[at(k, which)];
  };
  const bool cached = tiles <= 4 * KMAX;
  float c0[KMAX], c1[KMAX];
  double s = 0.0;
  if (cached) {
    if constexpr (MODE == BN_ST_TAGGED) {
      // batches of TB slabs (2*TB pairs in flight per thread: the landing registers of all 32 would push the 128-register
      // variants into scratch, and a kernel with scratch cannot count on full occupancy); later batches exist only when the
      // problem has that many slabs
      const long long t0 = wall_clock64();
#pragma unroll
      for (int hb = 0; hb < KMAX / TB; ++hb) {
        if (hb > 0 && tiles <= 4 * TB * hb) {
#pragma unroll
          for (int i = hb * TB; i < KMAX; ++i) c0[i] = c1[i] = 0.f;
          break;
        }
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int i = hb * TB; i < (hb + 1) * TB; ++i) {
            const int k = min(j + 4 * i, tiles - 1);
            const unsigned long long v0 = bn_ld_pair(pt + at(k, 0)), v1 = bn_ld_pair(pt + at(k, 1));
            ok &= (unsigned)(v0 >> 32) == tg.want && (unsigned)(v1 >> 32) == tg.want;
            c0[i] = __uint_as_float((unsigned)v0);
            c1[i] = __uint_as_float((unsigned)v1);
          }
          if (ok || bn_wait_expired(t0, tg.timeouts)) break;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < KMAX; ++i) {
        const int k = min(j + 4 * i, tiles - 1);        // clamped index, masked use: no branch around the loads
        c0[i] = ld(k, 0);
        c1[i] = ld(k, 1);
      }
    }
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
      if (j + 4 * i < tiles) s += (double)c0[i];
  } else {
    if constexpr (MODE == BN_ST_TAGGED) {              // wait for every pair of this thread first, then read as usual
      const long long t0 = wall_clock64();
      for (int k = j; k < tiles; k += 4)
        while ((unsigned)(bn_ld_pair(pt + at(k, 0)) >> 32) != tg.want || (unsigned)(bn_ld_pair(pt + at(k, 1)) >> 32) != tg.want)
          if (bn_wait_expired(t0, tg.timeouts)) break;
    }
    for (int k = j; k < tiles; k += 4) s += (double)ld(k, 0);
  }
  shd[j][ci] = s;
  __syncthreads();
  // (double-precision divisions are ~40-instruction sequences: every slab but the last has 128 rows -- an exact power-of-two
  // reciprocal -- and the two per-channel ones are multiplications by 1/M)
  const double inv_m = 1.0 / (double)Mrows;
  const int n_last = Mrows - (tiles - 1) * 128;
  const double inv_last = n_last == 128 ? 0.0078125 : 1.0 / (double)n_last;
  const double mean = ((shd[0][ci] + shd[1][ci]) + (shd[2][ci] + shd[3][ci])) * inv_m;
  double q = 0.0;
  if (cached) {
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      const int k = j + 4 * i;
      if (k < tiles) {
        const bool last = k == tiles - 1;
        const double d = (double)c0[i] * (last ? inv_last : 0.0078125) - mean;
        q += (double)c1[i] + (double)(last ? n_last : 128) * d * d;
      }
    }
  } else {
    for (int k = j; k < tiles; k += 4) {
      const bool last = k == tiles - 1;
      const double d = (double)ld(k, 0) * (last ? inv_last : 0.0078125) - mean;
      q += (double)ld(k, 1) + (double)(last ? n_last : 128) * d * d;
    }
  }
  __syncthreads();
  shd[j][ci] = q;
  __syncthreads();
  q = (shd[0][ci] + shd[1][ci]) + (shd[2][ci] + shd[3][ci]);
  double var = q * inv_m;
  if (var < 0.0) var = 0.0;
  __syncthreads();                       // shd is free again
  mean_out = mean;
  var_out = var;
}

// scale / shift of the normalisation and the inverse standard deviation backward needs, from the merged statistics.
__device__ __forceinline__ void bn_scale_shift(const double mean, const double var, const float eps, const float gamma,
                                               const float beta, float& sc, float& sh, float& invstd) {
#pragma clang fp contract(off)
  invstd = (float)(1.0 / sqrt(var + (double)eps));
  sc = gamma * invstd;
  sh = beta - (float)mean * sc;
}

// running statistics of nn.BatchNorm1d (momentum form; the conv bias that the GEMM leaves out shifts the mean only)
__device__ __forceinline__ void bn_running_update(const double mean, const double var, const int Mrows, const float momentum,
                                                  const float cb, float* running_mean, float* running_var, const int c) {
#pragma clang fp contract(off)
  const float unb = (float)(Mrows > 1 ? var * ((double)Mrows / (Mrows - 1)) : var);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * ((float)mean + cb);
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
}
