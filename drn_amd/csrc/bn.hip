// Train/eval BatchNorm1d (+ReLU) around the implicit-GEMM convs, channels-last.
//
// Reference: nn.BatchNorm1d inside conv blocks (model/basic_blocks.py:23-26, model/fcos.py:34,38,60,66),
// momentum 0.1, eps 1e-5, biased batch variance for normalisation, unbiased for running_var.
// Forward statistics come from the GEMM epilogue's per-tile column sums (deterministic); this file
// finalises them (double accumulation), applies scale/shift(+ReLU) with the fused consumers'
// prologues -- query gating (model/backbone.py:28-30) and FPN nearest-x2 upsample-add
// (model/FPN.py:63-68) -- and implements the backward pass (reduce -> finalize -> apply).
// A conv bias in front of a train-mode BN cancels in the output; it only shifts running_mean, so
// the GEMM never adds it (its gradient is analytically zero).
#include "vec.h"
#include "bn_merge.h"
#include "../../include/drn_hip.h"

struct BnFinGroup {
  const float* stats;
  int tiles, M;
  float* scale_shift;
  float* save;
  const float* gamma;
  const float* beta;
  const float* conv_bias;
  float* running_mean;
  float* running_var;
  float momentum, eps;
};
struct BnFinParams {
  int ngroups;
  BnFinGroup g[DRN_MAX_GROUPS];
};

// 256 threads per group (16 channels x 16 lanes over the per-tile partial sums, LDS-combined in a fixed order), one such
// slice of the workgroup per group, so the groups of a launch run side by side.  Only the running-statistics update is
// sequential: the thread that owns a channel applies the groups IN ORDER, so groups that share one BatchNorm module (the
// head modules applied once per pyramid level) update it like the reference does, and groups with their own modules
// (FPN levels) simply carry different pointers.
__global__ __launch_bounds__(256 * DRN_MAX_GROUPS) void bn_finalize_kernel(const BnFinParams P, int C) {
  __shared__ double sh[DRN_MAX_GROUPS][16][17];
  __shared__ float s_mean[DRN_MAX_GROUPS][16], s_unb[DRN_MAX_GROUPS][16];
  const int g = threadIdx.x >> 8, t = threadIdx.x & 255;
  const int ci = t & 15, j = t >> 4;
  const int c = blockIdx.x * 16 + ci;
  const bool live = c < C;
  const BnFinGroup& G = P.g[g];                     // blockDim.x = 256 * ngroups
  // slab k holds (sum, M2 about the slab mean) of n_k = min(128, M - 128 k) rows; merge in double (Chan et al.)
  // Up to 128 slabs (16384 rows): a thread's <= 8 (sum, M2) pairs are loaded ONCE, all loads in flight together, and kept in
  // registers for the second pass (the kernel is a pure latency chain: two dependent trips to memory cost ~2 us of its 8).
  constexpr int KMAX = 8;
  const int tiles = G.tiles, Mrows = G.M;
  const float* __restrict__ st = G.stats;
  const bool cached = tiles <= 16 * KMAX;
  float c0[KMAX], c1[KMAX];
  double s = 0.0;
  if (live) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < KMAX; ++i) {
        const int k = j + 16 * i;
        c0[i] = k < tiles ? st[((long)k * 2 + 0) * C + c] : 0.f;
        c1[i] = k < tiles ? st[((long)k * 2 + 1) * C + c] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < KMAX; ++i)
        if (j + 16 * i < tiles) s += (double)c0[i];
    } else {
      for (int k = j; k < tiles; k += 16) s += (double)st[((long)k * 2 + 0) * C + c];
    }
  }
  sh[g][ci][j] = s;
  __syncthreads();
  double mean = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) mean += sh[g][ci][k];
  mean /= Mrows;
  double q = 0.0;
  if (live) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < KMAX; ++i) {
        const int k = j + 16 * i;
        if (k < tiles) {
          const int nt = min(128, Mrows - k * 128);
          const double d = (double)c0[i] / nt - mean;
          q += (double)c1[i] + nt * d * d;
        }
      }
    } else {
      for (int k = j; k < tiles; k += 16) {
        const int nt = min(128, Mrows - k * 128);
        const double d = (double)st[((long)k * 2 + 0) * C + c] / nt - mean;
        q += (double)st[((long)k * 2 + 1) * C + c] + nt * d * d;
      }
    }
  }
  __syncthreads();
  sh[g][ci][j] = q;
  __syncthreads();
  if (live && j == 0) {
    q = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) q += sh[g][ci][k];
    double var = q / G.M;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)G.eps));
    const float sc = G.gamma[c] * invstd;
    G.scale_shift[c] = sc;
    G.scale_shift[C + c] = G.beta[c] - (float)mean * sc;
    G.save[c] = (float)mean;
    G.save[C + c] = invstd;
    s_mean[g][ci] = (float)mean;
    s_unb[g][ci] = (float)(G.M > 1 ? var * ((double)G.M / (G.M - 1)) : var);
  }
  __syncthreads();
  if (threadIdx.x < 16 && live)
    for (int k = 0; k < P.ngroups; ++k) {
      const BnFinGroup& H = P.g[k];
      const float cb = H.conv_bias ? H.conv_bias[c] : 0.f;
      if (H.running_mean) H.running_mean[c] = (1.f - H.momentum) * H.running_mean[c] + H.momentum * (s_mean[k][ci] + cb);
      if (H.running_var) H.running_var[c] = (1.f - H.momentum) * H.running_var[c] + H.momentum * s_unb[k][ci];
    }
}

static int bn_finalize_launch(const DrnBnFinDesc* d, int n, int C, void* stream, const char* who) {
  DRN_CHECK_ARG(d && n >= 1 && n <= DRN_MAX_GROUPS && C > 0, "%s: bad args", who);
  BnFinParams P;
  P.ngroups = n;
  for (int g = 0; g < n; ++g) {
    DRN_CHECK_ARG(d[g].stats && d[g].scale_shift && d[g].save && d[g].gamma && d[g].beta && d[g].M > 0 && d[g].tiles > 0,
                  "%s: bad group %d", who, g);
    P.g[g].stats = d[g].stats; P.g[g].tiles = d[g].tiles; P.g[g].M = d[g].M;
    P.g[g].scale_shift = d[g].scale_shift; P.g[g].save = d[g].save;
    P.g[g].gamma = d[g].gamma; P.g[g].beta = d[g].beta; P.g[g].conv_bias = d[g].conv_bias;
    P.g[g].running_mean = d[g].running_mean; P.g[g].running_var = d[g].running_var;
    P.g[g].momentum = d[g].momentum; P.g[g].eps = d[g].eps;
  }
  bn_finalize_kernel<<<cdiv(C, 16), 256 * n, 0, (hipStream_t)stream>>>(P, C);
  return drn_launch_status(who);
}

extern "C" int drn_bn_finalize_multi(const DrnBnFinDesc* descs, int n, int C, void* stream) {
  drn_clear_status();
  return bn_finalize_launch(descs, n, C, stream, "drn_bn_finalize_multi");
}

extern "C" int drn_bn_finalize(const DrnBnGroup* groups, int ngroups, int C, const float* gamma, const float* beta,
                               const float* conv_bias, float* running_mean, float* running_var, float momentum, float eps,
                               void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(groups && ngroups >= 1 && ngroups <= DRN_MAX_GROUPS, "drn_bn_finalize: bad args");
  DrnBnFinDesc d[DRN_MAX_GROUPS];
  for (int g = 0; g < ngroups; ++g) {
    d[g].stats = groups[g].stats; d[g].tiles = groups[g].tiles; d[g].M = groups[g].M;
    d[g].scale_shift = groups[g].scale_shift; d[g].save = groups[g].save;
    d[g].gamma = gamma; d[g].beta = beta; d[g].conv_bias = conv_bias;
    d[g].running_mean = running_mean; d[g].running_var = running_var; d[g].momentum = momentum; d[g].eps = eps;
  }
  return bn_finalize_launch(d, ngroups, C, stream, "drn_bn_finalize");
}

// eval mode: scale/shift from the running statistics (conv bias folded into the shift)
__global__ void bn_eval_ss_kernel(int C, const float* gamma, const float* beta, const float* conv_bias, const float* rm, const float* rv,
                                  float eps, float* ss) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  ss[c] = sc;
  ss[C + c] = beta[c] + ((conv_bias ? conv_bias[c] : 0.f) - rm[c]) * sc;
}
extern "C" int drn_bn_eval_scale_shift(int C, const float* gamma, const float* beta, const float* conv_bias, const float* running_mean,
                                       const float* running_var, float eps, float* scale_shift, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(C > 0 && gamma && beta && running_mean && running_var && scale_shift, "drn_bn_eval_scale_shift: bad args");
  bn_eval_ss_kernel<<<cdiv(C, 128), 128, 0, (hipStream_t)stream>>>(C, gamma, beta, conv_bias, running_mean, running_var, eps, scale_shift);
  return drn_launch_status("drn_bn_eval_scale_shift");
}

// out = [relu](raw*scale + shift) [+ up[s, t/2]] ;  gated = out * gate[s]
// One launch covers up to DRN_MAX_GROUPS levels (same C): level l owns the blocks [blk0_l, blk0_{l+1}).  Per level the
// launch geometry keeps (level threads) % nvec == 0, so each thread owns one 16-byte channel vector for all its rows:
// scale/shift live in registers and the row loop has no integer division by nvec.
struct BnApplyLv {
  const void* raw;
  void* out;
  const void* up;
  void* gated;
  const float* ss;
  const float* gate;
  int ld_raw, ld_out, ld_up, ld_gated, ldg, M, L, blk0;
};
struct BnApplyParams {
  BnApplyLv lv[DRN_MAX_GROUPS];
  int n, C, relu, total_blocks;
};

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyParams P) {
  constexpr int N = V16<T>::N;
  int li = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.n && (int)blockIdx.x >= P.lv[i].blk0) li = i;
  const BnApplyLv& G = P.lv[li];
  const int nblk = (li + 1 < P.n ? P.lv[li + 1].blk0 : P.total_blocks) - G.blk0;
  const int C = P.C, nvec = C / N, relu = P.relu;
  const int gtid = (blockIdx.x - G.blk0) * 256 + threadIdx.x;
  const int v = gtid % nvec, c0 = v * N;
  const int rstride = (nblk * 256) / nvec;
  const T* __restrict__ raw = (const T*)G.raw;
  const T* __restrict__ up = (const T*)G.up;
  T* __restrict__ out = (T*)G.out;
  T* __restrict__ gated = (T*)G.gated;
  const int M = G.M, L = G.L;
  float sc[N], sh[N];
#pragma unroll
  for (int k = 0; k < N; ++k) { sc[k] = G.ss[c0 + k]; sh[k] = G.ss[C + c0 + k]; }
  for (int m = gtid / nvec; m < M; m += rstride) {
    float x[N];
    V16<T>::load(raw + (long)m * G.ld_raw + c0, x);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float y = fmaf(x[k], sc[k], sh[k]);
      x[k] = relu ? fmaxf(y, 0.f) : y;
    }
    const int s = m / L;
    if (up) {
      const int t = m - s * L;
      float u[N];
      V16<T>::load(up + ((long)s * (L >> 1) + (t >> 1)) * G.ld_up + c0, u);
#pragma unroll
      for (int k = 0; k < N; ++k) x[k] += u[k];
    }
    V16<T>::store(out + (long)m * G.ld_out + c0, x);
    if (gated) {
      const float* gp = G.gate + (long)s * G.ldg + c0;
#pragma unroll
      for (int k = 0; k < N; ++k) x[k] *= gp[k];
      V16<T>::store(gated + (long)m * G.ld_gated + c0, x);
    }
  }
}

// grid size with (blocks*256) % nvec == 0, ~8 rows per thread, capped
static int row_grid(int M, int nvec) {
  int unit = nvec;                       // blocks must be a multiple of nvec / gcd(nvec, 256)
  int a = nvec, b = 256;
  while (b) { int t = a % b; a = b; b = t; }
  unit = nvec / a;
  long want = ((long)M * nvec + 256 * 4 - 1) / (256 * 4);
  if (want < 1) want = 1;
  if (want > 4096) want = 4096;
  long blocks = (want + unit - 1) / unit * unit;
  return (int)blocks;
}

static int bn_apply_launch(const DrnBnApplyDesc* d, int n, int C, int relu, int dtype, void* stream, const char* who) {
  DRN_CHECK_ARG(d && n >= 1 && n <= DRN_MAX_GROUPS && C > 0, "%s: bad args", who);
  const int vn = dtype == DRN_BF16 ? 8 : 4;
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "%s: bad dtype %d", who, dtype);
  DRN_CHECK_ARG(C % vn == 0, "%s: C must be a 16-byte multiple", who);
  BnApplyParams P;
  memset(&P, 0, sizeof(P));
  P.n = n; P.C = C; P.relu = relu;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const DrnBnApplyDesc& s = d[i];
    DRN_CHECK_ARG(s.raw && s.scale_shift && s.out && s.M > 0 && s.L > 0 && s.M % s.L == 0, "%s: bad level %d", who, i);
    DRN_CHECK_ARG(!s.up || (s.L % 2 == 0), "%s: upsample-add needs an even sequence length", who);
    DRN_CHECK_ARG((s.gate != nullptr) == (s.gated != nullptr), "%s: gate and gated must come together", who);
    DRN_CHECK_ARG(s.ld_raw % vn == 0 && s.ld_out % vn == 0 && (!s.up || s.ld_up % vn == 0) && (!s.gated || s.ld_gated % vn == 0),
                  "%s: ld must be 16-byte multiples", who);
    BnApplyLv& G = P.lv[i];
    G.raw = s.raw; G.out = s.out; G.up = s.up; G.gated = s.gated; G.ss = s.scale_shift; G.gate = s.gate;
    G.ld_raw = s.ld_raw; G.ld_out = s.ld_out; G.ld_up = s.ld_up; G.ld_gated = s.ld_gated; G.ldg = s.ldg; G.M = s.M; G.L = s.L;
    G.blk0 = blocks;
    blocks += row_grid(s.M, C / vn);
  }
  P.total_blocks = blocks;
  if (dtype == DRN_BF16) bn_apply_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(P);
  else bn_apply_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(P);
  return drn_launch_status(who);
}

extern "C" int drn_bn_apply_multi(const DrnBnApplyDesc* descs, int n, int C, int relu, int dtype, void* stream) {
  drn_clear_status();
  return bn_apply_launch(descs, n, C, relu, dtype, stream, "drn_bn_apply_multi");
}

extern "C" int drn_bn_apply(const void* raw, int ld_raw, const float* scale_shift, void* out, int ld_out, int M, int C, int L,
                            const void* up, int ld_up, const float* gate, int ldg, void* gated, int ld_gated, int relu, int dtype,
                            void* stream) {
  drn_clear_status();
  DrnBnApplyDesc d;
  d.raw = raw; d.ld_raw = ld_raw; d.scale_shift = scale_shift; d.out = out; d.ld_out = ld_out; d.M = M; d.L = L;
  d.up = up; d.ld_up = ld_up; d.gate = gate; d.ldg = ldg; d.gated = gated; d.ld_gated = ld_gated;
  return bn_apply_launch(&d, 1, C, relu, dtype, stream, "drn_bn_apply");
}

// ---------------------------------------------------------------- train-mode forward in ONE launch
// The separate finalize launch (5-8 us: a C/16-workgroup kernel that is one memory round trip long) is gone: every workgroup
// of the apply pass owns a (row block x 64-channel tile) and merges the slab statistics of ITS 64 channels itself -- tiles x
// 64 x 2 floats out of L2, all loads of a thread in flight together -- before it touches its rows.  All workgroups of a
// channel tile run the same instructions on the same numbers, so they agree bit for bit.  What must happen once per channel
// and IN GROUP ORDER -- scale/shift + (mean, invstd) for backward, the running statistics of a BatchNorm module that several
// groups share (the head towers, applied once per pyramid level: model/fcos.py:93-102) -- is done by C/64 extra "updater"
// workgroups at the front of the grid, one per channel tile, which walk the groups one after the other.
struct BnTrainLv {
  const void* raw;
  void* out;
  const void* up;
  void* gated;
  const float* gate;
  const float* stats;
  float* ss;
  float* save;
  const float* gamma;
  const float* beta;
  const float* conv_bias;
  float* running_mean;
  float* running_var;
  float momentum, eps;
  int ld_raw, ld_out, ld_up, ld_gated, ldg, M, L, tiles, blk0, rows_wg;
  int chain_next;   // >= 0: `up` is the output of THAT level of this launch (the FPN top-down chain, model/FPN.py:63-68): the
                    // rows to add are recomputed from its raw rows and statistics here instead of waiting for its output
};
struct BnTrainParams {
  BnTrainLv lv[DRN_MAX_GROUPS];
  int n, C, relu, nupd, total_blocks;
};

// The statistics merge itself lives in bn_merge.h (shared, bit for bit, with the conv -> BN -> ReLU kernel of gemm_nt_bn.hip).
__device__ __forceinline__ void bn_merge64(const float* __restrict__ st, const int tiles, const int Mrows, const int C,
                                           const int cbase, double (*shd)[64], double& mean_out, double& var_out) {
  bn_merge_cols<64, BN_ST_PLAIN>(st, tiles, Mrows, C, cbase, shd, mean_out, var_out);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_train_apply_kernel(const BnTrainParams P) {
  constexpr int N = V16<T>::N, NV = 64 / N, RP = 256 / NV;      // 16-byte vectors per 64-channel row piece, rows per pass
  __shared__ double shd[4][64];
  __shared__ float s_sc[64], s_sh[64];
  const int C = P.C, tid = threadIdx.x;
  if ((int)blockIdx.x < P.nupd) {
    const int cbase = blockIdx.x * 64, c = cbase + (tid & 63);
    for (int g = 0; g < P.n; ++g) {
      const BnTrainLv& G = P.lv[g];
      double mean, var;
      bn_merge64(G.stats, G.tiles, G.M, C, cbase, shd, mean, var);
      if (tid < 64) {
        float sc, sh, invstd;
        bn_scale_shift(mean, var, G.eps, G.gamma[c], G.beta[c], sc, sh, invstd);
        G.ss[c] = sc;
        G.ss[C + c] = sh;
        G.save[c] = (float)mean;
        G.save[C + c] = invstd;
        bn_running_update(mean, var, G.M, G.momentum, G.conv_bias ? G.conv_bias[c] : 0.f, G.running_mean, G.running_var, c);
      }
    }
    return;
  }
  int li = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.n && (int)blockIdx.x >= P.lv[i].blk0) li = i;
  const BnTrainLv& G = P.lv[li];
  const int ctiles = C >> 6;
  const int b = blockIdx.x - G.blk0;
  const int rblk = b / ctiles, cbase = (b - rblk * ctiles) * 64;
  const int v = tid % NV, ry = tid / NV, c0 = cbase + v * N;
  const T* __restrict__ raw = (const T*)G.raw;
  const T* __restrict__ up = (const T*)G.up;
  T* __restrict__ out = (T*)G.out;
  T* __restrict__ gated = (T*)G.gated;
  const int M = G.M, L = G.L, relu = P.relu;
  const long ld_raw = G.ld_raw, ld_out = G.ld_out, ld_up = G.ld_up, ld_gated = G.ld_gated, ldg = G.ldg;
  const float* __restrict__ gate = G.gate;
  const int row0 = rblk * G.rows_wg, row1 = min(M, row0 + G.rows_wg);
  constexpr int U = 4;
  typename V16<T>::raw_t xr[U], ur[U], ur2[U];
  int sq[U];
  // top-down chain inside one launch: out_l = relu(bn_l(raw_l)) + nearest_x2(out_{l+1}), out_{l+1} = T(relu(bn(raw_{l+1})) +
  // nearest_x2(out_{l+2})), out_{l+2} = T(relu(bn(raw_{l+2}))) -- every intermediate rounded to T exactly where the one-launch-
  // per-level order stores and re-reads it, so the bits are the same and the three levels need no order between them
  const int h1 = G.chain_next, h2 = h1 >= 0 ? P.lv[h1].chain_next : -1;
  const T* __restrict__ raw1 = h1 >= 0 ? (const T*)P.lv[h1].raw : nullptr;
  const T* __restrict__ raw2 = h2 >= 0 ? (const T*)P.lv[h2].raw : nullptr;
  const long ld1 = h1 >= 0 ? P.lv[h1].ld_raw : 0, ld2 = h2 >= 0 ? P.lv[h2].ld_raw : 0;
  auto load_batch = [&](int mb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = min(mb + u * RP, M - 1);           // clamped index, masked store
      xr[u] = V16<T>::ldraw(raw + (long)m * ld_raw + c0);
      sq[u] = m / L;
      const int t = m - sq[u] * L;
      if (raw1) {
        ur[u] = V16<T>::ldraw(raw1 + ((long)sq[u] * (L >> 1) + (t >> 1)) * ld1 + c0);
        if (raw2) ur2[u] = V16<T>::ldraw(raw2 + ((long)sq[u] * (L >> 2) + (t >> 2)) * ld2 + c0);
      } else if (up) {
        ur[u] = V16<T>::ldraw(up + ((long)sq[u] * (L >> 1) + (t >> 1)) * ld_up + c0);
      }
    }
  };
  // the rows of the first trip are requested BEFORE the statistics merge: every workgroup of a launch reaches its merge at the
  // same time, and without this the memory system idles through it
  load_batch(row0 + ry);
  __shared__ float s_sc1[2][64], s_sh1[2][64];
  {
    double mean, var;
    bn_merge64(G.stats, G.tiles, G.M, C, cbase, shd, mean, var);
    if (tid < 64) {
      float sc, sh, invstd;
      bn_scale_shift(mean, var, G.eps, G.gamma[cbase + tid], G.beta[cbase + tid], sc, sh, invstd);
      s_sc[tid] = sc;
      s_sh[tid] = sh;
    }
    for (int lev = 0; lev < 2; ++lev) {
      const int h = lev == 0 ? h1 : h2;
      if (h < 0) break;
      const BnTrainLv& H = P.lv[h];
      bn_merge64(H.stats, H.tiles, H.M, C, cbase, shd, mean, var);
      if (tid < 64) {
        float sc, sh, invstd;
        bn_scale_shift(mean, var, H.eps, H.gamma[cbase + tid], H.beta[cbase + tid], sc, sh, invstd);
        s_sc1[lev][tid] = sc;
        s_sh1[lev][tid] = sh;
      }
    }
    __syncthreads();
  }
  float sc[N], sh[N], sc1[N], sh1[N], sc2[N], sh2[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    sc[k] = s_sc[v * N + k]; sh[k] = s_sh[v * N + k];
    sc1[k] = raw1 ? s_sc1[0][v * N + k] : 0.f; sh1[k] = raw1 ? s_sh1[0][v * N + k] : 0.f;
    sc2[k] = raw2 ? s_sc1[1][v * N + k] : 0.f; sh2[k] = raw2 ? s_sh1[1][v * N + k] : 0.f;
  }
  for (int mb = row0 + ry; mb < row1; mb += U * RP) {
    if (mb != row0 + ry) load_batch(mb);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mb + u * RP;
      float x[N];
      V16<T>::cvt(xr[u], x);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float y = fmaf(x[k], sc[k], sh[k]);
        x[k] = relu ? fmaxf(y, 0.f) : y;
      }
      if (raw1) {
        float u1[N];
        V16<T>::cvt(ur[u], u1);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const float y = fmaf(u1[k], sc1[k], sh1[k]);
          u1[k] = relu ? fmaxf(y, 0.f) : y;
        }
        if (raw2) {
          float u2[N];
          V16<T>::cvt(ur2[u], u2);
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const float y = fmaf(u2[k], sc2[k], sh2[k]);
            u1[k] += DT<T>::round(relu ? fmaxf(y, 0.f) : y);
          }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] += DT<T>::round(u1[k]);
      } else if (up) {
        float uu[N];
        V16<T>::cvt(ur[u], uu);
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] += uu[k];
      }
      if (m < row1) {
        V16<T>::store(out + (long)m * ld_out + c0, x);
        if (gated) {
          const float* gp = gate + (long)sq[u] * ldg + c0;
#pragma unroll
          for (int k = 0; k < N; ++k) x[k] *= gp[k];
          V16<T>::store(gated + (long)m * ld_gated + c0, x);
        }
      }
    }
  }
}

extern "C" int drn_bn_train_apply(const DrnBnTrainDesc* d, int n, int C, int relu, int dtype, void* stream) {
  drn_clear_status();
  const char* who = "drn_bn_train_apply";
  DRN_CHECK_ARG(d && n >= 1 && n <= DRN_MAX_GROUPS && C > 0 && C % 64 == 0, "%s: bad args (C must be a multiple of 64)", who);
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "%s: bad dtype %d", who, dtype);
  const int vn = dtype == DRN_BF16 ? 8 : 4;
  const int RP = 256 / (64 / vn);
  BnTrainParams P;
  memset(&P, 0, sizeof(P));
  P.n = n; P.C = C; P.relu = relu; P.nupd = C / 64;
  long m_all = 0;
  for (int i = 0; i < n; ++i) m_all += d[i].M;
  // four rows per thread = ONE trip per workgroup, requested before the statistics merge (more trips only beyond 4096 workgroups)
  const long wg4 = ((m_all + 4 * RP - 1) / (4 * RP)) * (C / 64);
  const int rows_wg = 4 * RP * (int)(wg4 > 4096 ? (wg4 + 4095) / 4096 : 1);
  int blocks = P.nupd;
  for (int i = 0; i < n; ++i) {
    const DrnBnTrainDesc& s = d[i];
    DRN_CHECK_ARG(s.raw && s.out && s.stats && s.scale_shift && s.save && s.gamma && s.beta && s.M > 0 && s.L > 0 && s.M % s.L == 0 &&
                  s.tiles == (s.M + 127) / 128, "%s: bad level %d", who, i);
    DRN_CHECK_ARG(!s.up || (s.L % 2 == 0), "%s: upsample-add needs an even sequence length", who);
    DRN_CHECK_ARG((s.gate != nullptr) == (s.gated != nullptr), "%s: gate and gated must come together", who);
    DRN_CHECK_ARG(s.ld_raw % vn == 0 && s.ld_out % vn == 0 && (!s.up || s.ld_up % vn == 0) && (!s.gated || (s.ld_gated % vn == 0 && s.ldg % 4 == 0)),
                  "%s: ld must be 16-byte multiples", who);
    BnTrainLv& G = P.lv[i];
    G.raw = s.raw; G.out = s.out; G.up = s.up; G.gated = s.gated; G.gate = s.gate;
    G.stats = s.stats; G.ss = s.scale_shift; G.save = s.save; G.gamma = s.gamma; G.beta = s.beta; G.conv_bias = s.conv_bias;
    G.running_mean = s.running_mean; G.running_var = s.running_var; G.momentum = s.momentum; G.eps = s.eps;
    G.ld_raw = s.ld_raw; G.ld_out = s.ld_out; G.ld_up = s.ld_up; G.ld_gated = s.ld_gated; G.ldg = s.ldg; G.M = s.M; G.L = s.L;
    G.tiles = s.tiles; G.blk0 = blocks; G.rows_wg = rows_wg;
    blocks += cdiv(s.M, rows_wg) * (C / 64);
    // `up` = the output of another level of THIS launch (half the length, same clips): recomputed in place, no order needed
    G.chain_next = -1;
    for (int j = 0; j < n && s.up; ++j)
      if (j != i && d[j].out == s.up) {
        // a buffer this launch writes, read with nothing ordering the write before the read, unless it can be recomputed in place
        DRN_CHECK_ARG(d[j].L * 2 == s.L && d[j].M * 2 == s.M,
                      "%s: level %d adds the output of level %d of the same launch, whose geometry (M=%d L=%d) is not half of its own "
                      "(M=%d L=%d): launch the levels in order", who, i, j, d[j].M, d[j].L, s.M, s.L);
        G.chain_next = j;
        G.up = nullptr;
      }
  }
  for (int i = 0; i < n; ++i) {
    int depth = 1;
    for (int h = P.lv[i].chain_next; h >= 0 && depth <= n; h = P.lv[h].chain_next) ++depth;
    DRN_CHECK_ARG(depth <= 3, "%s: an upsample chain of more than three levels inside one launch (launch the levels in order)", who);
    DRN_CHECK_ARG(P.lv[i].chain_next < 0 || P.lv[i].L % 4 == 0 || P.lv[P.lv[i].chain_next].chain_next < 0,
                  "%s: a three-level chain needs sequence lengths that are multiples of 4", who);
  }
  P.total_blocks = blocks;
  if (dtype == DRN_BF16) bn_train_apply_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(P);
  else bn_train_apply_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(P);
  return drn_launch_status(who);
}

// ---------------------------------------------------------------- backward
// The ReLU mask is recomputed as fma(raw, scale, shift) > 0 with the forward's own scale/shift, so it is
// bit-identical to the forward decision even when `out` had the FPN upsample added on top.
// Three launches (reduce -> finalize -> apply) cover up to DRN_MAX_GROUPS levels of equal C; the levels may share one
// BatchNorm module (dgamma/dbeta accumulate over them in order) or carry their own.
struct BnBwdLv {
  const void* dout;
  const void* raw;
  void* draw;
  const float* ss;
  const float* save;
  const float* gamma;
  float* dgamma;
  float* dbeta;
  float* partial;   // [nblk][2][C]
  float* coef;      // [3][C]
  int ld_dout, ld_raw, ld_draw, M, accumulate, nblk, yblk0, ablk0;
};
struct BnBwdParams {
  BnBwdLv lv[DRN_MAX_GROUPS];
  int n, C, relu, total_yblk, total_ablk;
};

// g = dOut * mask;  partial[blk][0][c] = sum g ; partial[blk][1][c] = sum g * xhat
// block 256 = 64 channel vectors x 4 row lanes (LDS-combined in a fixed order); grid (ceil(nvec/64), sum of nblk)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdParams P) {
  constexpr int N = V16<T>::N;
  __shared__ float red[2][4][64 * N + 1];
  int li = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.n && (int)blockIdx.y >= P.lv[i].yblk0) li = i;
  const BnBwdLv& G = P.lv[li];
  const int C = P.C, relu = P.relu, M = G.M;
  const int by = blockIdx.y - G.yblk0;
  const T* __restrict__ dout = (const T*)G.dout;
  const T* __restrict__ raw = (const T*)G.raw;
  const int nvec = C / N;
  const int rows_per = (M + G.nblk - 1) / G.nblk;
  const int r0 = by * rows_per, r1 = min(M, r0 + rows_per);
  const int vx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int v = blockIdx.x * 64 + vx;
  const bool live = v < nvec;
  const int c0 = v * N;
  float mean[N], istd[N], sc[N], sh[N], sg[N], sx[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    mean[k] = live ? G.save[c0 + k] : 0.f;
    istd[k] = live ? G.save[C + c0 + k] : 0.f;
    sc[k] = live ? G.ss[c0 + k] : 0.f;
    sh[k] = live ? G.ss[C + c0 + k] : 0.f;
    sg[k] = 0.f;
    sx[k] = 0.f;
  }
  if (live) {
    const long ldd = G.ld_dout, ldr = G.ld_raw;       // (kept in registers: G is indexed by a run-time level number)
    int m = r0 + ry;
    for (; m + 12 < r1; m += 16) {                    // four rows = eight 16-byte loads in flight per trip
      float g[4][N], x[4][N];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        V16<T>::load(dout + (long)(m + 4 * u) * ldd + c0, g[u]);
        V16<T>::load(raw + (long)(m + 4 * u) * ldr + c0, x[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const float gg = (relu && !(fmaf(x[u][k], sc[k], sh[k]) > 0.f)) ? 0.f : g[u][k];
          sg[k] += gg;
          sx[k] = fmaf(gg, (x[u][k] - mean[k]) * istd[k], sx[k]);
        }
    }
    for (; m < r1; m += 4) {
      float g[N], x[N];
      V16<T>::load(dout + (long)m * ldd + c0, g);
      V16<T>::load(raw + (long)m * ldr + c0, x);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float gg = (relu && !(fmaf(x[k], sc[k], sh[k]) > 0.f)) ? 0.f : g[k];
        sg[k] += gg;
        sx[k] = fmaf(gg, (x[k] - mean[k]) * istd[k], sx[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    red[0][ry][vx * N + k] = sg[k];
    red[1][ry][vx * N + k] = sx[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 64 * N; i += 256) {
    const int kind = i / (64 * N), cc = i % (64 * N);
    const int c = blockIdx.x * 64 * N + cc;
    if (c < C) G.partial[((long)by * 2 + kind) * C + c] = red[kind][0][cc] + red[kind][1][cc] + red[kind][2][cc] + red[kind][3][cc];
  }
}

// dgamma/dbeta (+)= level sums;  coef: dRaw = A*g + B*raw + Cc.   256 threads per level (16 channels x 16 lanes over the
// partial blocks), one slice of the workgroup per level; dgamma / dbeta are then applied level after level by the thread
// that owns the channel, so levels sharing one module accumulate in a fixed order.
__global__ __launch_bounds__(256 * DRN_MAX_GROUPS) void bn_bwd_finalize_kernel(const BnBwdParams P) {
  __shared__ double sh[DRN_MAX_GROUPS][2][16][17];
  __shared__ float s_dg[DRN_MAX_GROUPS][16], s_db[DRN_MAX_GROUPS][16];
  const int C = P.C;
  const int li = threadIdx.x >> 8, t = threadIdx.x & 255;
  const int ci = t & 15, j = t >> 4;
  const int c = blockIdx.x * 16 + ci;
  const bool live = c < C;
  const BnBwdLv& G = P.lv[li];                      // blockDim.x = 256 * n
  double sg = 0.0, sx = 0.0;
  if (live) {
    const int nblk = G.nblk;
    const float* __restrict__ pp = G.partial;
    int b = j;
    for (; b + 16 * 3 < nblk; b += 16 * 4) {          // four independent pairs of loads in flight per trip
      float v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v0[u] = pp[((long)(b + 16 * u) * 2 + 0) * C + c];
        v1[u] = pp[((long)(b + 16 * u) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sg += (double)v0[u];
        sx += (double)v1[u];
      }
    }
    for (; b < nblk; b += 16) {
      sg += (double)pp[((long)b * 2 + 0) * C + c];
      sx += (double)pp[((long)b * 2 + 1) * C + c];
    }
  }
  sh[li][0][ci][j] = sg;
  sh[li][1][ci][j] = sx;
  __syncthreads();
  if (live && j == 0) {
    sg = 0.0; sx = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { sg += sh[li][0][ci][k]; sx += sh[li][1][ci][k]; }
    const float mean = G.save[c], istd = G.save[C + c];
    const float s = G.gamma[c] * istd;
    const float dg = (float)sx, db = (float)sg;
    const float invM = 1.f / (float)G.M;
    G.coef[c] = s;
    G.coef[C + c] = -s * dg * istd * invM;
    G.coef[2 * C + c] = -s * db * invM + s * dg * istd * mean * invM;
    s_dg[li][ci] = dg;
    s_db[li][ci] = db;
  }
  __syncthreads();
  if (threadIdx.x < 16 && live)
    for (int k = 0; k < P.n; ++k) {
      const BnBwdLv& H = P.lv[k];
      if (H.dgamma) H.dgamma[c] = H.accumulate ? H.dgamma[c] + s_dg[k][ci] : s_dg[k][ci];
      if (H.dbeta) H.dbeta[c] = H.accumulate ? H.dbeta[c] + s_db[k][ci] : s_db[k][ci];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdParams P) {
  constexpr int N = V16<T>::N;
  int li = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.n && (int)blockIdx.x >= P.lv[i].ablk0) li = i;
  const BnBwdLv& G = P.lv[li];
  const int nblk = (li + 1 < P.n ? P.lv[li + 1].ablk0 : P.total_ablk) - G.ablk0;
  const int C = P.C, relu = P.relu, M = G.M;
  const T* __restrict__ dout = (const T*)G.dout;
  const T* __restrict__ raw = (const T*)G.raw;
  T* __restrict__ draw = (T*)G.draw;
  const int nvec = C / N;
  const int gtid = (blockIdx.x - G.ablk0) * 256 + threadIdx.x;
  const int v = gtid % nvec, c0 = v * N;
  const int rstride = (nblk * 256) / nvec;
  float sc[N], sh[N], ka[N], kb[N], kc[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    sc[k] = G.ss[c0 + k]; sh[k] = G.ss[C + c0 + k];
    ka[k] = G.coef[c0 + k]; kb[k] = G.coef[C + c0 + k]; kc[k] = G.coef[2 * C + c0 + k];
  }
  for (int m = gtid / nvec; m < M; m += rstride) {
    float g[N], x[N];
    V16<T>::load(dout + (long)m * G.ld_dout + c0, g);
    V16<T>::load(raw + (long)m * G.ld_raw + c0, x);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float gg = (relu && !(fmaf(x[k], sc[k], sh[k]) > 0.f)) ? 0.f : g[k];
      x[k] = fmaf(ka[k], gg, fmaf(kb[k], x[k], kc[k]));
    }
    V16<T>::store(draw + (long)m * G.ld_draw + c0, x);
  }
}

// ---- backward in TWO launches for C % 64 == 0 (every layer of the model): the same (row block x 64-channel tile) geometry as
// the forward.  Reduce: 256 threads = NV 16-byte channel vectors x RP row lanes, four rows (eight loads) in flight per thread,
// the row lanes combined through LDS in a fixed order -> partial[rb][2][C], rb <= 64 row blocks.  Apply: every workgroup sums
// the <= 64 x 2 partials of its 64 channels itself (double, fixed order) and derives the three coefficients before it touches
// its rows -- the finalize launch is gone; dgamma / dbeta are written by C/64 updater workgroups that walk the levels in order
// (levels sharing one module accumulate in that order).
struct BnBwd2Lv {
  const void* dout;
  const void* raw;
  void* draw;
  const float* ss;
  const float* save;
  const float* gamma;
  float* dgamma;
  float* dbeta;
  float* partial;   // [rb][2][C]
  int ld_dout, ld_raw, ld_draw, M, accumulate, rb, rrows, rblk0, arows, ablk0;
};
struct BnBwd2Params {
  BnBwd2Lv lv[DRN_MAX_GROUPS];
  int n, C, relu, nupd;
};

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce64_kernel(const BnBwd2Params P) {
  constexpr int N = V16<T>::N, NV = 64 / N, RP = 256 / NV;
  __shared__ float red[2][RP][65];
  int li = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.n && (int)blockIdx.x >= P.lv[i].rblk0) li = i;
  const BnBwd2Lv& G = P.lv[li];
  const int C = P.C, relu = P.relu, M = G.M, tid = threadIdx.x;
  const int ctiles = C >> 6;
  const int b = blockIdx.x - G.rblk0;
  const int rblk = b / ctiles, cbase = (b - rblk * ctiles) * 64;
  const int v = tid % NV, ry = tid / NV, c0 = cbase + v * N;
  const T* __restrict__ dout = (const T*)G.dout;
  const T* __restrict__ raw = (const T*)G.raw;
  float mean[N], istd[N], sc[N], sh[N], sg[N], sx[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    mean[k] = G.save[c0 + k];
    istd[k] = G.save[C + c0 + k];
    sc[k] = G.ss[c0 + k];
    sh[k] = G.ss[C + c0 + k];
    sg[k] = 0.f;
    sx[k] = 0.f;
  }
  const long ldd = G.ld_dout, ldr = G.ld_raw;
  const int row0 = rblk * G.rrows, row1 = min(M, row0 + G.rrows);
  constexpr int U = 4;
  for (int mb = row0 + ry; mb < row1; mb += U * RP) {
    float g[U][N], x[U][N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = min(mb + u * RP, M - 1);           // clamped index, masked use
      V16<T>::load(dout + (long)m * ldd + c0, g[u]);
      V16<T>::load(raw + (long)m * ldr + c0, x[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = mb + u * RP < row1;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float gg = (!in || (relu && !(fmaf(x[u][k], sc[k], sh[k]) > 0.f))) ? 0.f : g[u][k];
        sg[k] += gg;
        sx[k] = fmaf(gg, (x[u][k] - mean[k]) * istd[k], sx[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    red[0][ry][v * N + k] = sg[k];
    red[1][ry][v * N + k] = sx[k];
  }
  __syncthreads();
  if (tid < 128) {
    const int kind = tid >> 6, cc = tid & 63;
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < RP; ++r) a += red[kind][r][cc];
    G.partial[((long)rblk * 2 + kind) * C + cbase + cc] = a;
  }
}

// column sums (double) of the rb partial pairs of channel cbase + (tid & 63): 64 channels x 4 lanes, combined in LDS
__device__ __forceinline__ void bn_bwd_sum64(const float* __restrict__ pp, const int rb, const int C, const int cbase,
                                             double (*shd)[4][64], double& sg_out, double& sx_out) {
  constexpr int KMAX = 16;
  const int ci = threadIdx.x & 63, j = threadIdx.x >> 6;
  const float* __restrict__ p = pp + cbase + ci;
  double sg = 0.0, sx = 0.0;
  if (rb <= 4 * KMAX) {
    float v0[KMAX], v1[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      const int k = min(j + 4 * i, rb - 1);
      v0[i] = p[((long)k * 2 + 0) * C];
      v1[i] = p[((long)k * 2 + 1) * C];
    }
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
      if (j + 4 * i < rb) {
        sg += (double)v0[i];
        sx += (double)v1[i];
      }
  } else {
    for (int k = j; k < rb; k += 4) {
      sg += (double)p[((long)k * 2 + 0) * C];
      sx += (double)p[((long)k * 2 + 1) * C];
    }
  }
  shd[0][j][ci] = sg;
  shd[1][j][ci] = sx;
  __syncthreads();
  sg_out = (shd[0][0][ci] + shd[0][1][ci]) + (shd[0][2][ci] + shd[0][3][ci]);
  sx_out = (shd[1][0][ci] + shd[1][1][ci]) + (shd[1][2][ci] + shd[1][3][ci]);
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply64_kernel(const BnBwd2Params P) {
  constexpr int N = V16<T>::N, NV = 64 / N, RP = 256 / NV;
  __shared__ double shd[2][4][64];
  __shared__ float s_k[3][64];
  const int C = P.C, tid = threadIdx.x;
  if ((int)blockIdx.x < P.nupd) {
    const int cbase = blockIdx.x * 64, c = cbase + (tid & 63);
    for (int g = 0; g < P.n; ++g) {
      const BnBwd2Lv& G = P.lv[g];
      double sg, sx;
      bn_bwd_sum64(G.partial, G.rb, C, cbase, shd, sg, sx);
      if (tid < 64) {
        const float dg = (float)sx, db = (float)sg;
        if (G.dgamma) G.dgamma[c] = G.accumulate ? G.dgamma[c] + dg : dg;
        if (G.dbeta) G.dbeta[c] = G.accumulate ? G.dbeta[c] + db : db;
      }
    }
    return;
  }
  int li = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.n && (int)blockIdx.x >= P.lv[i].ablk0) li = i;
  const BnBwd2Lv& G = P.lv[li];
  const int relu = P.relu, M = G.M;
  const int ctiles = C >> 6;
  const int b = blockIdx.x - G.ablk0;
  const int rblk = b / ctiles, cbase = (b - rblk * ctiles) * 64;
  const int v = tid % NV, ry = tid / NV, c0 = cbase + v * N;
  const T* __restrict__ dout = (const T*)G.dout;
  const T* __restrict__ raw = (const T*)G.raw;
  T* __restrict__ draw = (T*)G.draw;
  const long ldd = G.ld_dout, ldr = G.ld_raw, ldw = G.ld_draw;
  const int row0 = rblk * G.arows, row1 = min(M, row0 + G.arows);
  constexpr int U = 4;
  typename V16<T>::raw_t gr[U], xr[U];
  auto load_batch = [&](int mb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = min(mb + u * RP, M - 1);
      gr[u] = V16<T>::ldraw(dout + (long)m * ldd + c0);
      xr[u] = V16<T>::ldraw(raw + (long)m * ldr + c0);
    }
  };
  load_batch(row0 + ry);          // requested before the merge of the partial sums (see bn_train_apply_kernel)
  {
    double sg, sx;
    bn_bwd_sum64(G.partial, G.rb, C, cbase, shd, sg, sx);
    if (tid < 64) {
      const int c = cbase + tid;
      const float mean = G.save[c], istd = G.save[C + c];
      const float s = G.gamma[c] * istd;
      const float dg = (float)sx, db = (float)sg;
      const float invM = 1.f / (float)M;
      s_k[0][tid] = s;
      s_k[1][tid] = -s * dg * istd * invM;
      s_k[2][tid] = -s * db * invM + s * dg * istd * mean * invM;
    }
    __syncthreads();
  }
  float sc[N], sh[N], ka[N], kb[N], kc[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    sc[k] = G.ss[c0 + k];
    sh[k] = G.ss[C + c0 + k];
    ka[k] = s_k[0][v * N + k];
    kb[k] = s_k[1][v * N + k];
    kc[k] = s_k[2][v * N + k];
  }
  for (int mb = row0 + ry; mb < row1; mb += U * RP) {
    if (mb != row0 + ry) load_batch(mb);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mb + u * RP;
      float g[N], x[N];
      V16<T>::cvt(gr[u], g);
      V16<T>::cvt(xr[u], x);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float gg = (relu && !(fmaf(x[k], sc[k], sh[k]) > 0.f)) ? 0.f : g[k];
        x[k] = fmaf(ka[k], gg, fmaf(kb[k], x[k], kc[k]));
      }
      if (m < row1) V16<T>::store(draw + (long)m * ldw + c0, x);
    }
  }
}

static int bn_bwd_launch64(const DrnBnBwdDesc* d, int n, int C, int relu, float* ws, int dtype, hipStream_t stream, const char* who) {
  const int vn = dtype == DRN_BF16 ? 8 : 4;
  const int RP = 256 / (64 / vn);
  const int ctiles = C / 64;
  BnBwd2Params P;
  memset(&P, 0, sizeof(P));
  P.n = n; P.C = C; P.relu = relu; P.nupd = ctiles;
  long m_all = 0;
  for (int i = 0; i < n; ++i) m_all += d[i].M;
  // reduce: row blocks of a multiple of 4*RP rows (four rows per thread and trip), sized for ~512 workgroups per launch, at most
  // 64 per level (what an apply workgroup re-reads: 64 x 2 x 64 floats = 32 KB)
  int rrows = (int)((m_all * ctiles + 511) / 512);
  rrows = cdiv(rrows < 1 ? 1 : rrows, 4 * RP) * (4 * RP);
  const int arows = ((m_all + 4 * RP - 1) / (4 * RP)) * ctiles > 4096 ? 8 * RP : 4 * RP;
  int rb_total = 0, ab = P.nupd;
  for (int i = 0; i < n; ++i) {
    const DrnBnBwdDesc& s = d[i];
    DRN_CHECK_ARG(s.dout && s.raw && s.scale_shift && s.save && s.gamma && s.draw && s.M > 0, "%s: bad level %d", who, i);
    DRN_CHECK_ARG(s.ld_dout % vn == 0 && s.ld_raw % vn == 0 && s.ld_draw % vn == 0, "%s: ld must be 16-byte multiples", who);
    BnBwd2Lv& G = P.lv[i];
    G.dout = s.dout; G.raw = s.raw; G.draw = s.draw; G.ss = s.scale_shift; G.save = s.save; G.gamma = s.gamma;
    G.dgamma = s.dgamma; G.dbeta = s.dbeta; G.ld_dout = s.ld_dout; G.ld_raw = s.ld_raw; G.ld_draw = s.ld_draw; G.M = s.M;
    G.accumulate = s.accumulate;
    int rr = rrows;
    if (cdiv(s.M, rr) > 64) rr = cdiv(cdiv(s.M, 64), 4 * RP) * (4 * RP);
    G.rrows = rr;
    G.rb = cdiv(s.M, rr);
    G.partial = ws + (long)i * (2 * 256 + 3) * C;
    G.rblk0 = rb_total;
    rb_total += G.rb * ctiles;
    G.arows = arows;
    G.ablk0 = ab;
    ab += cdiv(s.M, arows) * ctiles;
  }
  if (dtype == DRN_BF16) {
    bn_bwd_reduce64_kernel<bf16_t><<<rb_total, 256, 0, stream>>>(P);
    bn_bwd_apply64_kernel<bf16_t><<<ab, 256, 0, stream>>>(P);
  } else {
    bn_bwd_reduce64_kernel<float><<<rb_total, 256, 0, stream>>>(P);
    bn_bwd_apply64_kernel<float><<<ab, 256, 0, stream>>>(P);
  }
  return drn_launch_status(who);
}

// ---- backward in ONE launch (C % 64 == 0 and the whole grid resident at once).  The same (row block x 64-channel tile)
// workgroups, but a workgroup keeps its rows of dout and raw IN REGISTERS (up to NP x 32 rows x 64 channels of bf16 = 16 bytes x
// 2 x NP per thread) between the two halves: it sums them, publishes its (sum g, sum g*xhat) pairs as TAGGED 64-bit words
// {value, launch generation} (bn_merge.h: one write-through store each, readers poll the data itself), waits for the other row
// blocks of ITS level and channel tile, derives the three coefficients like bn_bwd_apply64_kernel does and writes draw -- every
// element of dout and raw is read once, and the second launch with its ~5 us floor is gone.  The first workgroup of every
// channel tile also writes dgamma / dbeta, level after level; workgroup 0 finally advances the generation word, once it has seen
// a pair of every workgroup (each read the word before it published).
struct BnBwd1Lv {
  const void* dout;
  const void* raw;
  void* draw;
  const float* ss;
  const float* save;
  const float* gamma;
  float* dgamma;
  float* dbeta;
  unsigned long long* pairs;   // [rb][2][C] tagged
  unsigned long long* totals;  // [2][C] tagged: (dgamma, dbeta) contribution of this level
  int ld_dout, ld_raw, ld_draw, M, accumulate, rb, blk0;
  // GB launches (DrnBnBwdDesc::gb_*): this layer's output was gated by the query (model/backbone.py:28-30); the gradient of the GATED
  // output arrives in dg, dout (or NULL) is the gradient of the un-gated one
  const void* dg;
  const float* gate;
  float* dgate;
  int ld_dg, ldg, L;
};
struct BnBwd1Params {
  BnBwd1Lv lv[DRN_MAX_GROUPS];
  int n, C, relu;
  int* gen_word;
};
static __device__ int g_bn_bwd_timeouts;

// Sums of the rb <= 64 tagged (sum g, sum g*xhat) pairs of channel cbase + (tid & 63): bn_bwd_sum64's lanes and order.  First a
// quiet wait -- wave 0 alone polls, lane i the (sum g*xhat) pair of row block i, channel cbase; the other waves park at the barrier
// -- then every thread reads its pairs and checks each against the launch's tag.  (Requesting the pairs BEFORE the wait, so that a
// workgroup arriving late pays one round trip instead of two, was measured: every launch +20 us -- a pair read too early is re-read
// behind a ~1 us sleep, one after the other.)
__device__ __forceinline__ void bn_bwd_wait_sum64(const unsigned long long* __restrict__ pp, const int rb, const int C, const int cbase,
                                                  double (*shd)[4][64], double& sg_out, double& sx_out, const BnTagged tg) {
  constexpr int KH = 8;                                // two rounds of 8 pairs per lane (the rows this thread holds stay in registers)
  const int ci = threadIdx.x & 63, j = threadIdx.x >> 6;
  const unsigned long long* __restrict__ p = pp + cbase + ci;
  const long long t0 = wall_clock64();
  if (threadIdx.x < 64) {
    const unsigned long long* q = pp + ((long)min((int)threadIdx.x, rb - 1) * 2 + 1) * C + cbase;
    bool ready = (int)threadIdx.x >= rb;
    for (;;) {
      if (!ready) ready = (unsigned)(bn_ld_pair(q) >> 32) == tg.want;
      if (__builtin_amdgcn_ballot_w64(!ready) == 0) break;
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > 200000000LL) {                                          // 2 s of the 100 MHz wall clock
        if (!ready) __hip_atomic_fetch_add(tg.timeouts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  double sg = 0.0, sx = 0.0;
  for (int h = 0; h < 2; ++h) {
    if (h * 4 * KH >= rb) break;                       // (workgroup-uniform)
    unsigned long long v0[KH], v1[KH];
#pragma unroll
    for (int i = 0; i < KH; ++i) {
      const int k = min(j + 4 * (h * KH + i), rb - 1);
      v0[i] = bn_ld_pair(p + ((long)k * 2 + 0) * C);
      v1[i] = bn_ld_pair(p + ((long)k * 2 + 1) * C);
    }
#pragma unroll
    for (int i = 0; i < KH; ++i) {
      const int k = min(j + 4 * (h * KH + i), rb - 1);
      while ((unsigned)(v0[i] >> 32) != tg.want) {     // (the pairs of one row block land within the same microsecond)
        if (bn_wait_expired(t0, tg.timeouts)) break;
        v0[i] = bn_ld_pair(p + ((long)k * 2 + 0) * C);
      }
      while ((unsigned)(v1[i] >> 32) != tg.want) {
        if (bn_wait_expired(t0, tg.timeouts)) break;
        v1[i] = bn_ld_pair(p + ((long)k * 2 + 1) * C);
      }
      if (j + 4 * (h * KH + i) < rb) {
        sg += (double)__uint_as_float((unsigned)v0[i]);
        sx += (double)__uint_as_float((unsigned)v1[i]);
      }
    }
  }
  shd[0][j][ci] = sg;
  shd[1][j][ci] = sx;
  __syncthreads();
  sg_out = (shd[0][0][ci] + shd[0][1][ci]) + (shd[0][2][ci] + shd[0][3][ci]);
  sx_out = (shd[1][0][ci] + shd[1][1][ci]) + (shd[1][2][ci] + shd[1][3][ci]);
  __syncthreads();
}

template <typename T>
__device__ __forceinline__ typename V16<T>::raw_t bn_pack_raw(const float (&v)[V16<T>::N]) {
  typename V16<T>::raw_t r;
#pragma unroll
  for (int e = 0; e < V16<T>::N; ++e) r[e] = (T)v[e];
  return r;
}

// GB: the query-gate backward of this layer (drn_gate_bwd's arithmetic, statement for statement) runs on the rows as they arrive --
// dout_eff = T(dout + dg * gate[clip]) replaces the loaded gradient in registers, dgate[clip][c] = sum_t dg * act with the layer's
// activation recomputed from raw (what bn_train_apply stored: T(relu(raw * scale + shift))) -- and drn_gate_bwd's launch, the write of
// its result and its re-read are gone.  Clips must lie inside one workgroup's rows (ROWS % L == 0) and passes inside one clip (L % RP == 0).
template <typename T, int NP, bool GB = false>
__global__ __launch_bounds__(256, 2) void bn_bwd_one_kernel(const BnBwd1Params P) {
  constexpr int N = V16<T>::N, NV = 64 / N, RP = 256 / NV, ROWS = NP * RP;
  __shared__ float red[2][RP][65];
  __shared__ double shd[2][4][64];
  __shared__ float s_k[3][64];
  const int tid = threadIdx.x, C = P.C, relu = P.relu;
  const unsigned gen = (unsigned)__hip_atomic_load(P.gen_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const BnTagged tg{gen + 1u, &g_bn_bwd_timeouts};
  int li = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.n && (int)blockIdx.x >= P.lv[i].blk0) li = i;
  const BnBwd1Lv& G = P.lv[li];
  const int M = G.M, ctiles = C >> 6;
  const int b = blockIdx.x - G.blk0;
  const int rblk = b / ctiles, ct = b - rblk * ctiles, cbase = ct * 64;
  const int v = tid % NV, ry = tid / NV, c0 = cbase + v * N;
  const T* __restrict__ dout = (const T*)G.dout;
  const T* __restrict__ raw = (const T*)G.raw;
  T* __restrict__ draw = (T*)G.draw;
  const long ldd = G.ld_dout, ldr = G.ld_raw, ldw = G.ld_draw;
  const int row0 = rblk * ROWS + ry;
  typename V16<T>::raw_t gr[NP], xr[NP];
  float sc[N], sh[N];
  if constexpr (GB) {
    const T* __restrict__ dg = (const T*)G.dg;
    const long ldg_rows = G.ld_dg;
    typename V16<T>::raw_t ar[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int m = min(row0 + p * RP, M - 1);
      gr[p] = V16<T>::ldraw_nt(dg + (long)m * ldg_rows + c0);
      ar[p] = dout ? V16<T>::ldraw_nt(dout + (long)m * ldd + c0) : gr[p];
      xr[p] = V16<T>::ldraw_nt(raw + (long)m * ldr + c0);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      sc[k] = G.ss[c0 + k];
      sh[k] = G.ss[C + c0 + k];
    }
    const int L = G.L, nclip = ROWS / L;
    for (int q = 0; q < nclip; ++q) {                   // the clips of this row block, one after the other (workgroup-uniform)
      const int sq = rblk * nclip + q;
      if ((long)sq * L >= M) break;
      float gt[N], dacc[N];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        gt[k] = G.gate[(long)sq * G.ldg + c0 + k];
        dacc[k] = 0.f;
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if ((p * RP) / L != q) continue;
        const bool in = row0 + p * RP < M;
        float g[N], x[N], a[N];
        V16<T>::cvt(gr[p], g);
        V16<T>::cvt(xr[p], x);
        V16<T>::cvt(ar[p], a);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          float y = fmaf(x[k], sc[k], sh[k]);
          y = DT<T>::round(relu ? fmaxf(y, 0.f) : y);
          if (in) dacc[k] = fmaf(g[k], y, dacc[k]);
          a[k] = dout ? fmaf(g[k], gt[k], a[k]) : g[k] * gt[k];
        }
        gr[p] = bn_pack_raw<T>(a);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int k = 0; k < N; ++k) red[0][ry][v * N + k] = dacc[k];
      __syncthreads();
      if (tid < 64) {
        float a0 = 0.f;
#pragma unroll
        for (int r = 0; r < RP; ++r) a0 += red[0][r][tid];
        G.dgate[(long)sq * C + cbase + tid] = a0;
      }
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int p = 0; p < NP; ++p) {                       // everything this workgroup will ever read, in flight at once
      const int m = min(row0 + p * RP, M - 1);           // clamped index, masked use
      gr[p] = V16<T>::ldraw_nt(dout + (long)m * ldd + c0);          // (the gradient and the raw conv output: last readers)
      xr[p] = V16<T>::ldraw_nt(raw + (long)m * ldr + c0);
    }
  }
  {
    float mean[N], istd[N], sg[N], sx[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      mean[k] = G.save[c0 + k];
      istd[k] = G.save[C + c0 + k];
      sc[k] = G.ss[c0 + k];
      sh[k] = G.ss[C + c0 + k];
      sg[k] = 0.f;
      sx[k] = 0.f;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const bool in = row0 + p * RP < M;
      float g[N], x[N];
      V16<T>::cvt(gr[p], g);
      V16<T>::cvt(xr[p], x);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float gg = (!in || (relu && !(fmaf(x[k], sc[k], sh[k]) > 0.f))) ? 0.f : g[k];
        sg[k] += gg;
        sx[k] = fmaf(gg, (x[k] - mean[k]) * istd[k], sx[k]);
      }
      __builtin_amdgcn_sched_barrier(0);               // (one pass at a time: the conversions of all NP passes hoisted to the top spill)
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      red[0][ry][v * N + k] = sg[k];
      red[1][ry][v * N + k] = sx[k];
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p)                         // the rows stay as LOADED (16 bytes per vector), not as the floats of the pass above
    asm volatile("" : "+v"(gr[p]), "+v"(xr[p]));
  __syncthreads();
  if (tid < 64) {                                      // ONE wave publishes both sums, the polled one (sum g*xhat) last
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int r = 0; r < RP; ++r) a0 += red[0][r][tid];
#pragma unroll
    for (int r = 0; r < RP; ++r) a1 += red[1][r][tid];
    __hip_atomic_store(G.pairs + ((long)rblk * 2 + 0) * C + cbase + tid, bn_tag_pack(a0, tg.want), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(G.pairs + ((long)rblk * 2 + 1) * C + cbase + tid, bn_tag_pack(a1, tg.want), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  float my_dg = 0.f, my_db = 0.f;
  {
    double sg, sx;
    bn_bwd_wait_sum64(G.pairs, G.rb, C, cbase, shd, sg, sx, tg);
    if (tid < 64) {                                    // the statements of bn_bwd_apply64_kernel
      const int c = cbase + tid;
      const float mean = G.save[c], istd = G.save[C + c];
      const float s = G.gamma[c] * istd;
      const float dg = (float)sx, db = (float)sg;
      my_dg = dg;
      my_db = db;
      if (rblk == 0 && li > 0) {                       // this level's totals, for the workgroup that writes dgamma / dbeta (below)
        __hip_atomic_store(G.totals + c, bn_tag_pack(dg, tg.want), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(G.totals + C + c, bn_tag_pack(db, tg.want), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const float invM = 1.f / (float)M;
      s_k[0][tid] = s;
      s_k[1][tid] = -s * dg * istd * invM;
      s_k[2][tid] = -s * db * invM + s * dg * istd * mean * invM;
    }
    __syncthreads();
  }
  {
    float ka[N], kb[N], kc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      ka[k] = s_k[0][v * N + k];
      kb[k] = s_k[1][v * N + k];
      kc[k] = s_k[2][v * N + k];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int m = row0 + p * RP;
      float g[N], x[N];
      V16<T>::cvt(gr[p], g);
      V16<T>::cvt(xr[p], x);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float gg = (relu && !(fmaf(x[k], sc[k], sh[k]) > 0.f)) ? 0.f : g[k];
        x[k] = fmaf(ka[k], gg, fmaf(kb[k], x[k], kc[k]));
      }
      if (m < M) V16<T>::store(draw + (long)m * ldw + c0, x);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- dgamma / dbeta of this channel tile, level after level (levels that share a module accumulate in that order): written by the
  // first row block of the first level, which has its own totals and receives the other levels' from THEIR first row blocks
  if (li == 0 && rblk == 0 && tid < 64) {
    const int c = cbase + tid;
    const long long t0 = wall_clock64();
    for (int g = 0; g < P.n; ++g) {
      const BnBwd1Lv& H = P.lv[g];
      float dg = my_dg, db = my_db;
      if (g > 0) {
        unsigned long long a = bn_ld_pair(H.totals + c), b2 = bn_ld_pair(H.totals + C + c);
        while ((unsigned)(a >> 32) != tg.want || (unsigned)(b2 >> 32) != tg.want) {
          __builtin_amdgcn_s_sleep(4);
          if (wall_clock64() - t0 > 200000000LL) {
            __hip_atomic_fetch_add(tg.timeouts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          a = bn_ld_pair(H.totals + c);
          b2 = bn_ld_pair(H.totals + C + c);
        }
        dg = __uint_as_float((unsigned)a);
        db = __uint_as_float((unsigned)b2);
      }
      if (H.dgamma) H.dgamma[c] = H.accumulate ? H.dgamma[c] + dg : dg;
      if (H.dbeta) H.dbeta[c] = H.accumulate ? H.dbeta[c] + db : db;
    }
  }
  // ---- the generation moves on once EVERY workgroup of the launch has published under the old one
  if (blockIdx.x == 0) {
    const long long t0 = wall_clock64();
    for (int g = 0; g < P.n; ++g) {
      const BnBwd1Lv& H = P.lv[g];
      const int cnt = H.rb * ctiles;
      for (int i = tid; i < cnt; i += 256) {
        const int r = i / ctiles, t = i - r * ctiles;
        const unsigned long long* q = H.pairs + ((long)r * 2 + 1) * C + t * 64;
        while ((unsigned)(bn_ld_pair(q) >> 32) != tg.want)
          if (bn_wait_expired(t0, tg.timeouts)) break;
      }
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(P.gen_word, (int)(gen + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename KernelT>
static int bn_resident_capacity(KernelT kernel) {
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, 256, 0) != hipSuccess) return 0;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, (const void*)kernel) != hipSuccess) return 0;
  if (fa.localSizeBytes > 0) return 0;                 // (a variant that spills is not trusted with the wait: gemm_nt_bn.hip)
  return per_cu * cus;
}

// rows per workgroup index (NP = 2 << index) the launch would use, or -1: the smallest row block whose grid is at most `max_wg`
// workgroups (drn_tune bn1_maxwg, 512: two per CU, what the chip holds of them) with at most 64 row blocks per level -- and the
// next bigger one when that still leaves a workgroup for every CU (512 workgroups of 64 rows: 10.8 / 11.5 / 11.6 us for the three
// backbone stages; 256 of 128 rows: 10.6 / 10.2 / 11.0.  Below one per CU it loses: 224 of 512 rows 20.3 us, 448 of 256 rows 18.0)
static long bn_bwd_one_grid(const DrnBnBwdDesc* d, int n, int ctiles, int rows, int* rbmax) {
  long total = 0;
  *rbmax = 0;
  for (int i = 0; i < n; ++i) {
    const int rb = cdiv(d[i].M, rows);
    total += (long)rb * ctiles;
    if (rb > *rbmax) *rbmax = rb;
  }
  return total;
}
// gated levels (gb_dg): clips inside one workgroup's rows, passes inside one clip, at most 8 passes (the prologue holds a third row set)
static bool bn_bwd_one_gb_ok(const DrnBnBwdDesc* d, int n, int rows, int RP, int idx) {
  for (int i = 0; i < n; ++i) {
    if (!d[i].gb_dg) continue;
    const int L = d[i].gb_L;
    if (idx > 2 || L <= 0 || rows % L != 0 || L % RP != 0 || d[i].M % L != 0) return false;
  }
  return true;
}
static int bn_bwd_one_plan(const DrnBnBwdDesc* d, int n, int C, int dtype, int* total_out) {
  if (C % 64 != 0 || n < 1 || n > DRN_MAX_GROUPS) return -1;
  const int max_wg = drn_tuning(DRN_TUNE_BN1_MAXWG);
  const int RP = dtype == DRN_BF16 ? 32 : 16, ctiles = C / 64;
  for (int idx = 0; idx < 4; ++idx) {
    int rbmax = 0;
    long total = bn_bwd_one_grid(d, n, ctiles, (2 << idx) * RP, &rbmax);
    const bool ok = bn_bwd_one_gb_ok(d, n, (2 << idx) * RP, RP, idx);
    if (rbmax <= 64 && total <= max_wg) {
      if (idx < 3) {
        int rb2 = 0;
        const long t2 = bn_bwd_one_grid(d, n, ctiles, (4 << idx) * RP, &rb2);
        if ((t2 >= 256 || !ok) && bn_bwd_one_gb_ok(d, n, (4 << idx) * RP, RP, idx + 1)) { ++idx; total = t2; }
        else if (!ok) continue;
      } else if (!ok) {
        continue;
      }
      if (total_out) *total_out = (int)total;
      return idx;
    }
  }
  return -1;
}

extern "C" int64_t drn_bn_bwd_one_ws_bytes(const DrnBnBwdDesc* descs, int n, int C, int dtype) {
  if (!descs || bn_bwd_one_plan(descs, n, C, dtype, nullptr) < 0) return 0;
  return 64 + (int64_t)n * 65 * 2 * C * 8;
}

extern "C" int drn_bn_bwd_one(const DrnBnBwdDesc* d, int n, int C, int relu, void* tagged_ws, int64_t ws_bytes, int dtype, void* stream_) {
  drn_clear_status();
  const char* who = "drn_bn_bwd_one";
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(d && n >= 1 && n <= DRN_MAX_GROUPS && C > 0 && tagged_ws && ((uintptr_t)tagged_ws & 63) == 0, "%s: bad args", who);
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "%s: bad dtype %d", who, dtype);
  int total = 0;
  const int idx = bn_bwd_one_plan(d, n, C, dtype, &total);
  if (idx < 0) {
    drn_set_error("%s: the launch does not fit the chip at once (or C %% 64 != 0): use drn_bn_bwd_multi", who);
    return DRN_ERR_UNSUPPORTED;
  }
  DRN_CHECK_ARG(ws_bytes >= drn_bn_bwd_one_ws_bytes(d, n, C, dtype), "%s: workspace too small (drn_bn_bwd_one_ws_bytes)", who);
  const int vn = dtype == DRN_BF16 ? 8 : 4, RP = 256 / (64 / vn), rows = (2 << idx) * RP, ctiles = C / 64;
  BnBwd1Params P;
  memset(&P, 0, sizeof(P));
  P.n = n; P.C = C; P.relu = relu;
  P.gen_word = (int*)tagged_ws;
  unsigned long long* pairs = (unsigned long long*)((char*)tagged_ws + 64);
  int blk = 0;
  bool gated = false, all_gated = true;
  for (int i = 0; i < n; ++i) {
    const DrnBnBwdDesc& s = d[i];
    DRN_CHECK_ARG((s.dout || s.gb_dg) && s.raw && s.scale_shift && s.save && s.gamma && s.draw && s.M > 0, "%s: bad level %d", who, i);
    DRN_CHECK_ARG(s.ld_dout % vn == 0 && s.ld_raw % vn == 0 && s.ld_draw % vn == 0, "%s: ld must be 16-byte multiples", who);
    BnBwd1Lv& G = P.lv[i];
    G.dout = s.dout; G.raw = s.raw; G.draw = s.draw; G.ss = s.scale_shift; G.save = s.save; G.gamma = s.gamma;
    G.dgamma = s.dgamma; G.dbeta = s.dbeta; G.ld_dout = s.ld_dout; G.ld_raw = s.ld_raw; G.ld_draw = s.ld_draw; G.M = s.M;
    G.accumulate = s.accumulate;
    G.dg = s.gb_dg; G.gate = s.gb_gate; G.dgate = s.gb_dgate; G.ld_dg = s.gb_ld_dg; G.ldg = s.gb_ldg; G.L = s.gb_L;
    gated = gated || s.gb_dg != nullptr;
    all_gated = all_gated && s.gb_dg != nullptr;
    DRN_CHECK_ARG(!s.gb_dg || (s.gb_gate && s.gb_dgate && s.gb_ld_dg % vn == 0 && s.gb_ldg >= C), "%s: level %d: gb_* incomplete", who, i);
    G.rb = cdiv(s.M, rows);
    G.pairs = pairs + (long)i * 65 * 2 * C;
    G.totals = G.pairs + (long)64 * 2 * C;
    G.blk0 = blk;
    blk += G.rb * ctiles;
  }
  static int capacity[2][4];
  int& cap = capacity[dtype == DRN_BF16][idx];
  DRN_CHECK_ARG(!gated || all_gated, "%s: gated and plain levels do not mix in one launch", who);
  static int capacity_gb[2][3];
#define BN1_CASE(TT, NPV) do { \
    if (!cap) cap = bn_resident_capacity(bn_bwd_one_kernel<TT, NPV>); \
    if (total > cap) { drn_set_error("%s: %d workgroups exceed the %d the chip holds at once", who, total, cap); return DRN_ERR_UNSUPPORTED; } \
    bn_bwd_one_kernel<TT, NPV><<<total, 256, 0, stream>>>(P); } while (0)
#define BN1_CASE_GB(TT, NPV) do { \
    int& capg = capacity_gb[dtype == DRN_BF16][idx]; \
    if (!capg) capg = bn_resident_capacity(bn_bwd_one_kernel<TT, NPV, true>); \
    if (total > capg) { drn_set_error("%s: %d workgroups exceed the %d the chip holds at once", who, total, capg); return DRN_ERR_UNSUPPORTED; } \
    bn_bwd_one_kernel<TT, NPV, true><<<total, 256, 0, stream>>>(P); } while (0)
  if (gated) {
    if (dtype == DRN_BF16) { switch (idx) { case 0: BN1_CASE_GB(bf16_t, 2); break; case 1: BN1_CASE_GB(bf16_t, 4); break; default: BN1_CASE_GB(bf16_t, 8); } }
    else { switch (idx) { case 0: BN1_CASE_GB(float, 2); break; case 1: BN1_CASE_GB(float, 4); break; default: BN1_CASE_GB(float, 8); } }
  } else if (dtype == DRN_BF16) {
    switch (idx) { case 0: BN1_CASE(bf16_t, 2); break; case 1: BN1_CASE(bf16_t, 4); break; case 2: BN1_CASE(bf16_t, 8); break; default: BN1_CASE(bf16_t, 16); }
  } else {
    switch (idx) { case 0: BN1_CASE(float, 2); break; case 1: BN1_CASE(float, 4); break; case 2: BN1_CASE(float, 8); break; default: BN1_CASE(float, 16); }
  }
#undef BN1_CASE
#undef BN1_CASE_GB
  return drn_launch_status(who);
}

// Watchdog of drn_bn_bwd_one's wait: workgroups that gave up after 2 s (0 in a healthy run; results of such a launch are invalid).
// Synchronises the device.
extern "C" int drn_bn_bwd_one_timeouts(int reset) {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_bn_bwd_timeouts), sizeof(int)) != hipSuccess) return -1;
  if (reset && v) {
    const int z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bn_bwd_timeouts), &z, sizeof(int));
  }
  return v;
}

// draw may alias dout (in place).  ws >= n * (2*256 + 3) * C floats.
static int bn_bwd_launch(const DrnBnBwdDesc* d, int n, int C, int relu, float* ws, int dtype, void* stream_, const char* who) {
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(d && n >= 1 && n <= DRN_MAX_GROUPS && C > 0 && ws, "%s: bad args", who);
  for (int i = 0; i < n; ++i) DRN_CHECK_ARG(!d[i].gb_dg, "%s: DrnBnBwdDesc::gb_* is drn_bn_bwd_one's (run drn_gate_bwd first)", who);
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "%s: bad dtype %d", who, dtype);
  const int vn = dtype == DRN_BF16 ? 8 : 4;
  DRN_CHECK_ARG(C % vn == 0, "%s: C must be a 16-byte multiple", who);
  if (C % 64 == 0) return bn_bwd_launch64(d, n, C, relu, ws, dtype, stream, who);
  BnBwdParams P;
  memset(&P, 0, sizeof(P));
  P.n = n; P.C = C; P.relu = relu;
  int yb = 0, ab = 0;
  long m_all = 0;
  for (int i = 0; i < n; ++i) m_all += d[i].M;
  for (int i = 0; i < n; ++i) {
    const DrnBnBwdDesc& s = d[i];
    DRN_CHECK_ARG(s.dout && s.raw && s.scale_shift && s.save && s.gamma && s.draw && s.M > 0, "%s: bad level %d", who, i);
    DRN_CHECK_ARG(s.ld_dout % vn == 0 && s.ld_raw % vn == 0 && s.ld_draw % vn == 0, "%s: ld must be 16-byte multiples", who);
    BnBwdLv& G = P.lv[i];
    G.dout = s.dout; G.raw = s.raw; G.draw = s.draw; G.ss = s.scale_shift; G.save = s.save; G.gamma = s.gamma;
    G.dgamma = s.dgamma; G.dbeta = s.dbeta; G.ld_dout = s.ld_dout; G.ld_raw = s.ld_raw; G.ld_draw = s.ld_draw; G.M = s.M;
    G.accumulate = s.accumulate;
    // row blocks: 16 rows each (4 per row lane), 64 when the launch has two or more column blocks and many rows -- the shared
    // head's 14336 x 1024 gradient went as 1280 workgroups of 4-8 rows per thread and took 31 us for 59 MB
    const long cbk = cdiv(C / vn, 64);
    const int rows_blk = cbk * (m_all / 64) >= 256 ? 64 : (cbk * (m_all / 32) >= 256 ? 32 : 16);   // (the coarsest that still fills the chip)
    G.nblk = s.M >= 256 * rows_blk ? 256 : (s.M >= rows_blk ? s.M / rows_blk : 1);
    G.partial = ws + (long)i * (2 * 256 + 3) * C;
    G.coef = G.partial + (long)2 * 256 * C;
    G.yblk0 = yb; G.ablk0 = ab;
    yb += G.nblk;
    ab += row_grid(s.M, C / vn);
  }
  P.total_yblk = yb; P.total_ablk = ab;
  dim3 grid(cdiv(C / vn, 64), yb);
  if (dtype == DRN_BF16) bn_bwd_reduce_kernel<bf16_t><<<grid, 256, 0, stream>>>(P);
  else bn_bwd_reduce_kernel<float><<<grid, 256, 0, stream>>>(P);
  bn_bwd_finalize_kernel<<<cdiv(C, 16), 256 * n, 0, stream>>>(P);
  if (dtype == DRN_BF16) bn_bwd_apply_kernel<bf16_t><<<ab, 256, 0, stream>>>(P);
  else bn_bwd_apply_kernel<float><<<ab, 256, 0, stream>>>(P);
  return drn_launch_status(who);
}

extern "C" int drn_bn_bwd_multi(const DrnBnBwdDesc* descs, int n, int C, int relu, float* ws, int dtype, void* stream) {
  drn_clear_status();
  return bn_bwd_launch(descs, n, C, relu, ws, dtype, stream, "drn_bn_bwd_multi");
}

extern "C" int drn_bn_bwd(const void* dout, int ld_dout, const void* raw, int ld_raw, const float* scale_shift, const float* save,
                          const float* gamma, void* draw, int ld_draw, float* dgamma, float* dbeta, int accumulate, int M, int C,
                          int relu, float* ws /* >= (2*256+3)*C floats */, int dtype, void* stream) {
  drn_clear_status();
  DrnBnBwdDesc d;
  d.dout = dout; d.ld_dout = ld_dout; d.raw = raw; d.ld_raw = ld_raw; d.scale_shift = scale_shift; d.save = save; d.gamma = gamma;
  d.draw = draw; d.ld_draw = ld_draw; d.dgamma = dgamma; d.dbeta = dbeta; d.accumulate = accumulate; d.M = M;
  d.gb_dg = nullptr; d.gb_gate = nullptr; d.gb_dgate = nullptr; d.gb_ld_dg = d.gb_ldg = d.gb_L = 0;
  return bn_bwd_launch(&d, 1, C, relu, ws, dtype, stream, "drn_bn_bwd");
}
