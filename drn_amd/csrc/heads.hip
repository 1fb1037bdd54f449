// Dense per-location output heads with 1-2 output channels: cls_logits (512->1, k3), bbox_pred
// (512->2, k3, then exp(scale_l * z)) and the last iou_scores conv (256->1, k1)
// (model/fcos.py:43-49, 68, 96-102).  These are GEMV-shaped (N <= 2): one wavefront per location,
// lanes split the channels with 16-byte loads, wave shuffles finish the dot products -- no MFMA.
// All pyramid levels go through one launch (shared weights); outputs are fp32.
#include "vec.h"
#include "../../include/drn_hip.h"

#define HEAD_MAX_N 2
#define HEAD_MAX_TAPS 3

struct HeadGroup {
  const void* X;   // activations (channels-last), possibly a column slice of a wider buffer
  void* dX;        // backward only: gradient buffer with the same geometry
  int ldx, M, L, row_start;
  const float* scale;  // exp mode: one float per level
};
struct HeadParams {
  int ngroups, total_rows;
  HeadGroup g[DRN_MAX_GROUPS];
  int N, C, taps, pad, exp_mode;
};

__device__ __forceinline__ int head_group_of(const HeadParams& P, int r) {
  int g = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.ngroups && r >= P.g[i].row_start) g = i;
  return g;
}

// out[r][n] (and z[r][n] in exp mode); r = concatenated row over levels
template <typename T>
__global__ __launch_bounds__(256) void head_out_fwd_kernel(const HeadParams P, const float* __restrict__ W /*[N][C][taps]*/,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           float* __restrict__ z) {
  extern __shared__ float wl[];  // [N][taps][C]
  constexpr int VN = V16<T>::N;
  const int N = P.N, C = P.C, taps = P.taps;
  for (int i = threadIdx.x; i < N * taps * C; i += blockDim.x) {
    const int c = i % C, tn = i / C, tap = tn % taps, n = tn / taps;
    wl[i] = W[((long)n * C + c) * taps + tap];
  }
  __syncthreads();
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int nvec = C / VN;
  for (int r = blockIdx.x * 4 + w; r < P.total_rows; r += gridDim.x * 4) {
    const int g = head_group_of(P, r);
    const HeadGroup& G = P.g[g];
    const int m = r - G.row_start;
    const int s = m / G.L, t = m - s * G.L;
    const T* __restrict__ X = (const T*)G.X;
    float acc[HEAD_MAX_N] = {0.f, 0.f};
    for (int tap = 0; tap < taps; ++tap) {
      const int st = t + tap - P.pad;
      if (st < 0 || st >= G.L) continue;
      const T* row = X + (long)(s * G.L + st) * G.ldx;
      for (int v = l; v < nvec; v += 64) {
        float x[VN];
        V16<T>::load(row + v * VN, x);
#pragma unroll
        for (int n = 0; n < HEAD_MAX_N; ++n)
          if (n < N) {
            const float* wp = wl + (n * taps + tap) * C + v * VN;
#pragma unroll
            for (int k = 0; k < VN; ++k) acc[n] = fmaf(x[k], wp[k], acc[n]);
          }
      }
    }
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n)
      if (n < N) {
        float v = wave_sum(acc[n]) + bias[n];
        if (l == 0) {
          if (P.exp_mode) {
            z[(long)r * N + n] = v;
            v = expf(G.scale[0] * v);
          }
          out[(long)r * N + n] = v;
        }
      }
  }
}

// dX[m][c] (+)= sum_n sum_tap dz[row(s, t - tap + pad)][n] * W[n][c][tap]
template <typename T>
__global__ __launch_bounds__(256) void head_out_bwd_data_kernel(const HeadParams P, const float* __restrict__ W,
                                                                const float* __restrict__ dz, int accumulate) {
  extern __shared__ float wl[];  // [N][taps][C]
  constexpr int VN = V16<T>::N;
  const int N = P.N, C = P.C, taps = P.taps;
  for (int i = threadIdx.x; i < N * taps * C; i += blockDim.x) {
    const int c = i % C, tn = i / C, tap = tn % taps, n = tn / taps;
    wl[i] = W[((long)n * C + c) * taps + tap];
  }
  __syncthreads();
  const int nvec = C / VN;
  const long total = (long)P.total_rows * nvec;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const int r = (int)(i / nvec);
    const int g = head_group_of(P, r);
    const HeadGroup& G = P.g[g];
    const int m = r - G.row_start;
    const int s = m / G.L, t = m - s * G.L;
    T* dst = (T*)G.dX + (long)m * G.ldx + v * VN;
    float a[VN];
    if (accumulate) V16<T>::load(dst, a);
    else {
#pragma unroll
      for (int k = 0; k < VN; ++k) a[k] = 0.f;
    }
    for (int tap = 0; tap < taps; ++tap) {
      const int to = t - tap + P.pad;   // output position that read this input through `tap`
      if (to < 0 || to >= G.L) continue;
      const float* d = dz + (long)(G.row_start + s * G.L + to) * N;
#pragma unroll
      for (int n = 0; n < HEAD_MAX_N; ++n)
        if (n < N) {
          const float dv = d[n];
          const float* wp = wl + (n * taps + tap) * C + v * VN;
#pragma unroll
          for (int k = 0; k < VN; ++k) a[k] = fmaf(dv, wp[k], a[k]);
        }
    }
    V16<T>::store(dst, a);
  }
}

// partial[blk][n][tap][c] = sum over the block's rows of dz[r][n] * X[src(r,tap)][c]
// grid (ceil(nvec/64), nblk); block 256 = 64 channel vectors x 4 row lanes
template <typename T>
__global__ __launch_bounds__(256) void head_out_bwd_w_kernel(const HeadParams P, const float* __restrict__ dz, float* __restrict__ partial) {
  constexpr int VN = V16<T>::N;
  __shared__ float red[4][64 * VN + 1];
  const int N = P.N, C = P.C, taps = P.taps;
  const int vx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int v = blockIdx.x * 64 + vx;
  const bool live = v * VN < C;
  const int rows_per = (P.total_rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(P.total_rows, r0 + rows_per);
  float acc[HEAD_MAX_N][HEAD_MAX_TAPS][VN];
#pragma unroll
  for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
    for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp)
#pragma unroll
      for (int k = 0; k < VN; ++k) acc[n][tp][k] = 0.f;
  if (live) {
    for (int r = r0 + ry; r < r1; r += 4) {
      const int g = head_group_of(P, r);
      const HeadGroup& G = P.g[g];
      const int m = r - G.row_start;
      const int s = m / G.L, t = m - s * G.L;
      const T* __restrict__ X = (const T*)G.X;
      float d[HEAD_MAX_N];
#pragma unroll
      for (int n = 0; n < HEAD_MAX_N; ++n) d[n] = n < N ? dz[(long)r * N + n] : 0.f;
#pragma unroll
      for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp) {
        if (tp >= taps) continue;
        const int st = t + tp - P.pad;
        if (st < 0 || st >= G.L) continue;
        float x[VN];
        V16<T>::load(X + (long)(s * G.L + st) * G.ldx + v * VN, x);
#pragma unroll
        for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
          for (int k = 0; k < VN; ++k) acc[n][tp][k] = fmaf(d[n], x[k], acc[n][tp][k]);
      }
    }
  }
  for (int n = 0; n < N; ++n)
    for (int tp = 0; tp < taps; ++tp) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < VN; ++k) {
        float val = 0.f;
#pragma unroll
        for (int nn = 0; nn < HEAD_MAX_N; ++nn)
#pragma unroll
          for (int tt = 0; tt < HEAD_MAX_TAPS; ++tt)
            if (nn == n && tt == tp) val = acc[nn][tt][k];
        red[ry][vx * VN + k] = val;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < 64 * VN; i += 256) {
        const int c = blockIdx.x * 64 * VN + i;
        if (c < C) partial[(((long)blockIdx.y * N + n) * taps + tp) * C + c] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
      }
    }
}

// dW[n][c][tap] (+)= sum_blk partial[blk][n][tap][c];  256 threads = 16 outputs x 16 lanes over the partial blocks
__global__ __launch_bounds__(256) void head_out_bwd_w_final_kernel(const float* __restrict__ partial, int nblk, int N, int C, int taps,
                                                                   float* __restrict__ dW, int accumulate) {
  __shared__ float sh[16][17];
  const int total = N * taps * C;
  const int oi = threadIdx.x & 15, j = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + oi;
  float s = 0.f;
  if (i < total)
    for (int b = j; b < nblk; b += 16) s += partial[(long)b * total + i];
  sh[oi][j] = s;
  __syncthreads();
  if (j != 0 || i >= total) return;
  s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += sh[oi][k];
  const int c = i % C, tn = i / C, tap = tn % taps, n = tn / taps;
  float* dst = dW + ((long)n * C + c) * taps + tap;
  *dst = accumulate ? *dst + s : s;
}

// exp-mode chain rule + bias gradients, one workgroup:
//   dz[r][n] = exp_mode ? scale_l * out[r][n] * dout[r][n] : dout[r][n]
//   dscale[l] (+)= sum_{r in level l} z * out * dout ;  dbias[n] (+)= sum_r dz[r][n]
__global__ __launch_bounds__(1024) void head_out_bwd_pre_kernel(const HeadParams P, const float* __restrict__ dout,
                                                                const float* __restrict__ out, const float* __restrict__ z,
                                                                float* __restrict__ dz, float* __restrict__ dbias,
                                                                float* __restrict__ dscale, int accumulate) {
  __shared__ float sh[17];
  const int N = P.N;
  float db[HEAD_MAX_N] = {0.f, 0.f}, ds[DRN_MAX_GROUPS] = {0.f, 0.f, 0.f, 0.f};
  for (int r = threadIdx.x; r < P.total_rows; r += blockDim.x) {
    const int g = head_group_of(P, r);
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n)
      if (n < N) {
        float d = dout[(long)r * N + n];
        if (P.exp_mode) {
          const float rd = out[(long)r * N + n] * d;
#pragma unroll
          for (int l = 0; l < DRN_MAX_GROUPS; ++l)
            if (l == g) ds[l] += z[(long)r * N + n] * rd;
          d = P.g[g].scale[0] * rd;
        }
        dz[(long)r * N + n] = d;
        db[n] += d;
      }
  }
#pragma unroll
  for (int n = 0; n < HEAD_MAX_N; ++n) db[n] = block_sum(db[n], sh);
#pragma unroll
  for (int l = 0; l < DRN_MAX_GROUPS; ++l) ds[l] = block_sum(ds[l], sh);
  if (threadIdx.x == 0) {
    for (int n = 0; n < N; ++n) dbias[n] = accumulate ? dbias[n] + db[n] : db[n];
    if (P.exp_mode)
      for (int l = 0; l < P.ngroups; ++l) dscale[l] = accumulate ? dscale[l] + ds[l] : ds[l];
  }
}

static int fill_head_params(HeadParams& P, const DrnHeadGroup* groups, int ngroups, int N, int C, int taps, int exp_mode, int dtype,
                            bool need_dx, const char* who) {
  DRN_CHECK_ARG(groups && ngroups >= 1 && ngroups <= DRN_MAX_GROUPS, "%s: bad group count", who);
  DRN_CHECK_ARG(N >= 1 && N <= HEAD_MAX_N && (taps == 1 || taps == 3) && C > 0, "%s: N<=2, taps in {1,3} required", who);
  const int vn = dtype == DRN_BF16 ? 8 : 4;
  memset(&P, 0, sizeof(P));
  P.ngroups = ngroups;
  int rows = 0;
  for (int g = 0; g < ngroups; ++g) {
    DRN_CHECK_ARG(groups[g].X && groups[g].M > 0 && groups[g].L > 0 && groups[g].M % groups[g].L == 0, "%s: bad group %d", who, g);
    DRN_CHECK_ARG(C % vn == 0 && groups[g].ldx % vn == 0, "%s: C/ldx must be 16-byte multiples", who);
    DRN_CHECK_ARG(!need_dx || groups[g].dX, "%s: dX missing in group %d", who, g);
    DRN_CHECK_ARG(!exp_mode || groups[g].scale, "%s: scale missing in group %d", who, g);
    P.g[g].X = groups[g].X; P.g[g].dX = groups[g].dX; P.g[g].ldx = groups[g].ldx; P.g[g].M = groups[g].M; P.g[g].L = groups[g].L;
    P.g[g].scale = groups[g].scale; P.g[g].row_start = rows;
    rows += groups[g].M;
  }
  P.total_rows = rows;
  P.N = N; P.C = C; P.taps = taps; P.pad = (taps - 1) / 2; P.exp_mode = exp_mode;
  return DRN_OK;
}

extern "C" int drn_head_out_fwd(const DrnHeadGroup* groups, int ngroups, const float* W, const float* bias, int N, int C, int taps,
                                int exp_mode, float* out, float* z, int dtype, void* stream) {
  drn_clear_status();
  HeadParams P;
  int rc = fill_head_params(P, groups, ngroups, N, C, taps, exp_mode, dtype, false, "drn_head_out_fwd");
  if (rc) return rc;
  DRN_CHECK_ARG(W && bias && out && (!exp_mode || z), "drn_head_out_fwd: null pointer");
  const size_t shm = (size_t)N * taps * C * sizeof(float);
  int nb = cdiv(P.total_rows, 4);
  if (nb > 1024) nb = 1024;
  DISPATCH_DT(dtype, "drn_head_out_fwd", { head_out_fwd_kernel<T><<<nb, 256, shm, (hipStream_t)stream>>>(P, W, bias, out, z); });
  return drn_launch_status("drn_head_out_fwd");
}

extern "C" int drn_head_out_bwd(const DrnHeadGroup* groups, int ngroups, const float* W, const float* dout, const float* out,
                                const float* z, int N, int C, int taps, int exp_mode, int accumulate_dx, float* dW, float* dbias,
                                float* dscale, int accumulate_dw, float* ws /* >= R*N + 64 + 256*N*taps*C floats */, int dtype,
                                void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  HeadParams P;
  int rc = fill_head_params(P, groups, ngroups, N, C, taps, exp_mode, dtype, true, "drn_head_out_bwd");
  if (rc) return rc;
  DRN_CHECK_ARG(W && dout && dW && dbias && ws && (!exp_mode || (out && z && dscale)), "drn_head_out_bwd: null pointer");
  float* dz = ws;
  float* part = ws + (((long)P.total_rows * N + 63) / 64) * 64;
  const size_t shm = (size_t)N * taps * C * sizeof(float);
  const int nblk = P.total_rows >= 256 * 16 ? 256 : (P.total_rows >= 16 ? P.total_rows / 16 : 1);
  head_out_bwd_pre_kernel<<<1, 1024, 0, stream>>>(P, dout, out, z, dz, dbias, dscale, accumulate_dw);
  DISPATCH_DT(dtype, "drn_head_out_bwd", {
    constexpr int VN = V16<T>::N;
    head_out_bwd_data_kernel<T><<<ew_blocks((long)P.total_rows * (C / VN), 256, 2048), 256, shm, stream>>>(P, W, dz, accumulate_dx);
    dim3 grid(cdiv(C / VN, 64), nblk);
    head_out_bwd_w_kernel<T><<<grid, 256, 0, stream>>>(P, dz, part);
  });
  head_out_bwd_w_final_kernel<<<cdiv(N * taps * C, 16), 256, 0, stream>>>(part, nblk, N, C, taps, dW, accumulate_dw);
  return drn_launch_status("drn_head_out_bwd");
}
