// Train-mode BatchNorm statistics merge, shared by the stand-alone pass (bn.hip: bn_train_apply_kernel, 64 channels x 4 slab
// lanes = 256 threads) and by the conv -> BN -> ReLU kernel (gemm_nt_kernel.h: nt_epilogue_bn, 128 channels x 4 slab lanes = 512
// threads): ONE copy of the arithmetic, contraction off, so that both paths -- and every workgroup of either -- produce the same
// bits (nn.BatchNorm1d in training, model/basic_blocks.py:23-26; per-level-call statistics, model/fcos.py:93-102).
#pragma once
#include "common.h"

// COHERENT: the slab statistics were written by OTHER workgroups of the SAME launch (write-through stores): read them with
// agent-scope relaxed atomic loads (global_load ... sc1), which do not hit in this XCD's non-coherent L2.
template <bool COHERENT>
__device__ __forceinline__ float bn_ld_stat(const float* p) {
  if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

// (mean, biased variance) of channel cbase + (tid % CB) from the GEMM epilogue's per-128-row-slab (sum, M2) pairs, merged in
// double with the parallel-variance formula (Chan et al.).  4*CB threads = CB channels x 4 slab lanes; the lanes meet in LDS
// and are added in a fixed order.  Up to 64 slabs (8192 rows) a thread's pairs are loaded once, all in flight together, and
// kept in registers for the second pass; longer problems re-read them (L2 hits).  Ends with a workgroup barrier (shd is free).
template <int CB, bool COHERENT>
__device__ __forceinline__ void bn_merge_cols(const float* __restrict__ st, const int tiles, const int Mrows, const int C,
                                              const int cbase, double (*shd)[CB], double& mean_out, double& var_out) {
#pragma clang fp contract(off)
  constexpr int KMAX = 16;
  const int ci = threadIdx.x % CB, j = threadIdx.x / CB;
  const float* __restrict__ p = st + cbase + ci;
  const bool cached = tiles <= 4 * KMAX;
  float c0[KMAX], c1[KMAX];
  double s = 0.0;
  if (cached) {
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      const int k = min(j + 4 * i, tiles - 1);        // clamped index, masked use: no branch around the loads
      c0[i] = bn_ld_stat<COHERENT>(p + ((long)k * 2 + 0) * C);
      c1[i] = bn_ld_stat<COHERENT>(p + ((long)k * 2 + 1) * C);
    }
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
      if (j + 4 * i < tiles) s += (double)c0[i];
  } else {
    for (int k = j; k < tiles; k += 4) s += (double)bn_ld_stat<COHERENT>(p + ((long)k * 2 + 0) * C);
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
      const double d = (double)bn_ld_stat<COHERENT>(p + ((long)k * 2 + 0) * C) * (last ? inv_last : 0.0078125) - mean;
      q += (double)bn_ld_stat<COHERENT>(p + ((long)k * 2 + 1) * C) + (double)(last ? n_last : 128) * d * d;
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
