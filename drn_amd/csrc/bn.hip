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
#include "../../include/drn_hip.h"

struct BnFinGroup {
  const float* stats;
  int tiles, M;
  float* scale_shift;
  float* save;
};
struct BnFinParams {
  int ngroups;
  BnFinGroup g[DRN_MAX_GROUPS];
};

// 256 threads = 16 channels x 16 lanes; the lanes split the per-tile partial sums, LDS combines them in a fixed order.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const BnFinParams P, int C, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ conv_bias,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float momentum, float eps) {
  __shared__ double sh[2][16][17];
  const int ci = threadIdx.x & 15, j = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + ci;
  const bool live = c < C;
  float rm = 0.f, rv = 1.f, cb = 0.f;
  if (live && j == 0) {
    rm = running_mean ? running_mean[c] : 0.f;
    rv = running_var ? running_var[c] : 1.f;
    cb = conv_bias ? conv_bias[c] : 0.f;
  }
  for (int g = 0; g < P.ngroups; ++g) {   // shared modules: levels update the running stats in order
    const BnFinGroup& G = P.g[g];
    // slab t holds (sum, M2 about the slab mean) of n_t = min(128, M - 128 t) rows; merge in double (Chan et al.)
    double s = 0.0;
    if (live)
      for (int t = j; t < G.tiles; t += 16) s += (double)G.stats[((long)t * 2 + 0) * C + c];
    __syncthreads();
    sh[0][ci][j] = s;
    __syncthreads();
    double mean = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) mean += sh[0][ci][k];
    mean /= G.M;
    double q = 0.0;
    if (live)
      for (int t = j; t < G.tiles; t += 16) {
        const int nt = min(128, G.M - t * 128);
        const double d = (double)G.stats[((long)t * 2 + 0) * C + c] / nt - mean;
        q += (double)G.stats[((long)t * 2 + 1) * C + c] + nt * d * d;
      }
    __syncthreads();
    sh[1][ci][j] = q;
    __syncthreads();
    if (live && j == 0) {
      q = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) q += sh[1][ci][k];
      double var = q / G.M;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float sc = gamma[c] * invstd;
      G.scale_shift[c] = sc;
      G.scale_shift[C + c] = beta[c] - (float)mean * sc;
      G.save[c] = (float)mean;
      G.save[C + c] = invstd;
      const double unbiased = G.M > 1 ? var * ((double)G.M / (G.M - 1)) : var;
      rm = (1.f - momentum) * rm + momentum * ((float)mean + cb);
      rv = (1.f - momentum) * rv + momentum * (float)unbiased;
    }
  }
  if (live && j == 0) {
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
  }
}

extern "C" int drn_bn_finalize(const DrnBnGroup* groups, int ngroups, int C, const float* gamma, const float* beta,
                               const float* conv_bias, float* running_mean, float* running_var, float momentum, float eps,
                               void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(groups && ngroups >= 1 && ngroups <= DRN_MAX_GROUPS && C > 0 && gamma && beta, "drn_bn_finalize: bad args");
  BnFinParams P;
  P.ngroups = ngroups;
  for (int g = 0; g < ngroups; ++g) {
    DRN_CHECK_ARG(groups[g].stats && groups[g].scale_shift && groups[g].save && groups[g].M > 0 && groups[g].tiles > 0,
                  "drn_bn_finalize: bad group %d", g);
    P.g[g].stats = groups[g].stats; P.g[g].tiles = groups[g].tiles; P.g[g].M = groups[g].M;
    P.g[g].scale_shift = groups[g].scale_shift; P.g[g].save = groups[g].save;
  }
  bn_finalize_kernel<<<cdiv(C, 16), 256, 0, (hipStream_t)stream>>>(P, C, gamma, beta, conv_bias, running_mean, running_var, momentum, eps);
  return drn_launch_status("drn_bn_finalize");
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
// Launch geometry keeps (total threads) % nvec == 0, so each thread owns one 16-byte channel vector for all its
// rows: scale/shift live in registers and the row loop has no integer division by nvec.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ raw, int ld_raw, const float* __restrict__ ss,
                                                       T* __restrict__ out, int ld_out, int M, int C, int L,
                                                       const T* __restrict__ up, int ld_up, const float* __restrict__ gate, int ldg,
                                                       T* __restrict__ gated, int ld_gated, int relu) {
  constexpr int N = V16<T>::N;
  const int nvec = C / N;
  const int gtid = blockIdx.x * 256 + threadIdx.x;
  const int v = gtid % nvec, c0 = v * N;
  const int rstride = (gridDim.x * 256) / nvec;
  float sc[N], sh[N];
#pragma unroll
  for (int k = 0; k < N; ++k) { sc[k] = ss[c0 + k]; sh[k] = ss[C + c0 + k]; }
  for (int m = gtid / nvec; m < M; m += rstride) {
    float x[N];
    V16<T>::load(raw + (long)m * ld_raw + c0, x);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float y = fmaf(x[k], sc[k], sh[k]);
      x[k] = relu ? fmaxf(y, 0.f) : y;
    }
    const int s = m / L;
    if (up) {
      const int t = m - s * L;
      float u[N];
      V16<T>::load(up + ((long)s * (L >> 1) + (t >> 1)) * ld_up + c0, u);
#pragma unroll
      for (int k = 0; k < N; ++k) x[k] += u[k];
    }
    V16<T>::store(out + (long)m * ld_out + c0, x);
    if (gated) {
      const float* gp = gate + (long)s * ldg + c0;
#pragma unroll
      for (int k = 0; k < N; ++k) x[k] *= gp[k];
      V16<T>::store(gated + (long)m * ld_gated + c0, x);
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

extern "C" int drn_bn_apply(const void* raw, int ld_raw, const float* scale_shift, void* out, int ld_out, int M, int C, int L,
                            const void* up, int ld_up, const float* gate, int ldg, void* gated, int ld_gated, int relu, int dtype,
                            void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(raw && scale_shift && out && M > 0 && C > 0 && L > 0 && M % L == 0, "drn_bn_apply: bad args");
  DRN_CHECK_ARG(!up || (L % 2 == 0), "drn_bn_apply: upsample-add needs an even sequence length");
  DRN_CHECK_ARG((gate != nullptr) == (gated != nullptr), "drn_bn_apply: gate and gated must come together");
  DISPATCH_DT(dtype, "drn_bn_apply", {
    constexpr int N = V16<T>::N;
    DRN_CHECK_ARG(C % N == 0 && ld_raw % N == 0 && ld_out % N == 0 && (!up || ld_up % N == 0) && (!gated || ld_gated % N == 0),
                  "drn_bn_apply: C/ld must be 16-byte multiples");
    bn_apply_kernel<T><<<row_grid(M, C / N), 256, 0, (hipStream_t)stream>>>(
        (const T*)raw, ld_raw, scale_shift, (T*)out, ld_out, M, C, L, (const T*)up, ld_up, gate, ldg, (T*)gated, ld_gated, relu);
  });
  return drn_launch_status("drn_bn_apply");
}

// ---------------------------------------------------------------- backward
// The ReLU mask is recomputed as fma(raw, scale, shift) > 0 with the forward's own scale/shift, so it is
// bit-identical to the forward decision even when `out` had the FPN upsample added on top.
// g = dOut * mask;  partial[blk][0][c] = sum g ; partial[blk][1][c] = sum g * xhat
// block 256 = 64 channel vectors x 4 row lanes (LDS-combined in a fixed order); grid (ceil(nvec/64), nblk)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, int ld_dout, const T* __restrict__ raw, int ld_raw,
                                                            const float* __restrict__ ss, const float* __restrict__ save, int M, int C,
                                                            int relu, float* __restrict__ partial) {
  constexpr int N = V16<T>::N;
  __shared__ float red[2][4][64 * N + 1];
  const int nvec = C / N;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  const int vx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int v = blockIdx.x * 64 + vx;
  const bool live = v < nvec;
  const int c0 = v * N;
  float mean[N], istd[N], sc[N], sh[N], sg[N], sx[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    mean[k] = live ? save[c0 + k] : 0.f;
    istd[k] = live ? save[C + c0 + k] : 0.f;
    sc[k] = live ? ss[c0 + k] : 0.f;
    sh[k] = live ? ss[C + c0 + k] : 0.f;
    sg[k] = 0.f;
    sx[k] = 0.f;
  }
  if (live)
    for (int m = r0 + ry; m < r1; m += 4) {
      float g[N], x[N];
      V16<T>::load(dout + (long)m * ld_dout + c0, g);
      V16<T>::load(raw + (long)m * ld_raw + c0, x);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float gg = (relu && !(fmaf(x[k], sc[k], sh[k]) > 0.f)) ? 0.f : g[k];
        sg[k] += gg;
        sx[k] = fmaf(gg, (x[k] - mean[k]) * istd[k], sx[k]);
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
    if (c < C) partial[((long)blockIdx.y * 2 + kind) * C + c] = red[kind][0][cc] + red[kind][1][cc] + red[kind][2][cc] + red[kind][3][cc];
  }
}

// dgamma/dbeta (+)= level sums;  coef: dRaw = A*g + B*raw + Cc.   256 threads = 16 channels x 16 lanes over the partials.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int M, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ save,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                              float* __restrict__ coef) {
  __shared__ double sh[2][16][17];
  const int ci = threadIdx.x & 15, j = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + ci;
  double sg = 0.0, sx = 0.0;
  if (c < C)
    for (int b = j; b < nblk; b += 16) {
      sg += (double)partial[((long)b * 2 + 0) * C + c];
      sx += (double)partial[((long)b * 2 + 1) * C + c];
    }
  sh[0][ci][j] = sg;
  sh[1][ci][j] = sx;
  __syncthreads();
  if (c >= C || j != 0) return;
  sg = 0.0; sx = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { sg += sh[0][ci][k]; sx += sh[1][ci][k]; }
  const float mean = save[c], istd = save[C + c];
  const float s = gamma[c] * istd;
  const float dg = (float)sx, db = (float)sg;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + dg : dg;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + db : db;
  const float invM = 1.f / (float)M;
  coef[c] = s;
  coef[C + c] = -s * dg * istd * invM;
  coef[2 * C + c] = -s * db * invM + s * dg * istd * mean * invM;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, int ld_dout, const T* __restrict__ raw, int ld_raw,
                                                           const float* __restrict__ ss, const float* __restrict__ coef,
                                                           T* __restrict__ draw, int ld_draw, int M, int C, int relu) {
  constexpr int N = V16<T>::N;
  const int nvec = C / N;
  const int gtid = blockIdx.x * 256 + threadIdx.x;
  const int v = gtid % nvec, c0 = v * N;
  const int rstride = (gridDim.x * 256) / nvec;
  float sc[N], sh[N], ka[N], kb[N], kc[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    sc[k] = ss[c0 + k]; sh[k] = ss[C + c0 + k];
    ka[k] = coef[c0 + k]; kb[k] = coef[C + c0 + k]; kc[k] = coef[2 * C + c0 + k];
  }
  for (int m = gtid / nvec; m < M; m += rstride) {
    float g[N], x[N];
    V16<T>::load(dout + (long)m * ld_dout + c0, g);
    V16<T>::load(raw + (long)m * ld_raw + c0, x);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float gg = (relu && !(fmaf(x[k], sc[k], sh[k]) > 0.f)) ? 0.f : g[k];
      x[k] = fmaf(ka[k], gg, fmaf(kb[k], x[k], kc[k]));
    }
    V16<T>::store(draw + (long)m * ld_draw + c0, x);
  }
}

// draw may alias dout (in place).
extern "C" int drn_bn_bwd(const void* dout, int ld_dout, const void* raw, int ld_raw, const float* scale_shift, const float* save,
                          const float* gamma, void* draw, int ld_draw, float* dgamma, float* dbeta, int accumulate, int M, int C,
                          int relu, float* ws /* >= (2*256+3)*C floats */, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(dout && raw && scale_shift && save && gamma && draw && ws && M > 0 && C > 0, "drn_bn_bwd: bad args");
  const int nblk = M >= 256 * 16 ? 256 : (M >= 16 ? M / 16 : 1);
  float* coef = ws + (long)2 * 256 * C;
  DISPATCH_DT(dtype, "drn_bn_bwd", {
    constexpr int N = V16<T>::N;
    DRN_CHECK_ARG(C % N == 0 && ld_dout % N == 0 && ld_raw % N == 0 && ld_draw % N == 0, "drn_bn_bwd: C/ld must be 16-byte multiples");
    dim3 grid(cdiv(C / N, 64), nblk);
    bn_bwd_reduce_kernel<T><<<grid, 256, 0, stream>>>((const T*)dout, ld_dout, (const T*)raw, ld_raw, scale_shift, save, M, C, relu, ws);
    bn_bwd_finalize_kernel<<<cdiv(C, 16), 256, 0, stream>>>(ws, nblk, M, C, gamma, save, dgamma, dbeta, accumulate, coef);
    bn_bwd_apply_kernel<T><<<row_grid(M, C / N), 256, 0, stream>>>((const T*)dout, ld_dout, (const T*)raw, ld_raw,
                                                                                 scale_shift, coef, (T*)draw, ld_draw, M, C, relu);
  });
  return drn_launch_status("drn_bn_bwd");
}
