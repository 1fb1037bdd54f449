// Language-guided pooling (model/LGP.py:29-51): per clip and per pair of adjacent positions (2j, 2j+1),
//   s_p = <x[b,2j+p,:], q'[b,:]>,  att = softmax_p(s),  out[b,j,:] = att_0 x[b,2j,:] + att_1 x[b,2j+1,:]
// with q' = BN(conv1x1(tile(q))) prepared by the caller (the 1x1 conv on a tiled query is one small GEMM and its BN
// runs over the batch only).  HBM-bound: each x row is read once, kept in registers for the dot products and the
// blend; one wavefront per (clip, pair), wave-shuffle reductions, 16-byte channel vectors.
#include "vec.h"
#include "../../include/drn_hip.h"

#define LGP_MAXV 4   // channel vectors per lane: C <= 64 * 4 * V16<T>::N

template <typename T>
__global__ __launch_bounds__(256) void lgp_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ qn, T* __restrict__ out,
                                                      int ld_out, float* __restrict__ att, int B, int t, int C) {
  constexpr int N = V16<T>::N;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int half = t >> 1, nvec = C / N;
  for (int pr = blockIdx.x * 4 + w; pr < B * half; pr += gridDim.x * 4) {
    const int b = pr / half, j = pr - b * half;
    const T* r0 = x + ((long)b * t + 2 * j) * ldx;
    const T* r1 = r0 + ldx;
    const float* q = qn + (long)b * C;
    float x0[LGP_MAXV][N], x1[LGP_MAXV][N];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int u = 0; u < LGP_MAXV; ++u) {
      const int v = l + u * 64;
      if (v < nvec) {
        V16<T>::load(r0 + v * N, x0[u]);
        V16<T>::load(r1 + v * N, x1[u]);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const float qq = q[v * N + k];
          s0 = fmaf(x0[u][k], qq, s0);
          s1 = fmaf(x1[u][k], qq, s1);
        }
      }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    const float mx = fmaxf(s0, s1);
    const float e0 = expf(s0 - mx), e1 = expf(s1 - mx);
    const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
    if (l == 0) {
      att[(long)pr * 2] = a0;
      att[(long)pr * 2 + 1] = a1;
    }
    T* o = out + (long)pr * ld_out;
#pragma unroll
    for (int u = 0; u < LGP_MAXV; ++u) {
      const int v = l + u * 64;
      if (v < nvec) {
        float y[N];
#pragma unroll
        for (int k = 0; k < N; ++k) y[k] = a0 * x0[u][k] + a1 * x1[u][k];
        V16<T>::store(o + v * N, y);
      }
    }
  }
}

extern "C" int drn_lgp_fwd(const void* x, int ldx, const float* qn, void* out, int ld_out, float* att, int B, int t, int C, int dtype,
                           void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(x && qn && out && att && B > 0 && t > 0 && t % 2 == 0 && C > 0, "drn_lgp_fwd: bad args (t must be even)");
  DISPATCH_DT(dtype, "drn_lgp_fwd", {
    constexpr int N = V16<T>::N;
    DRN_CHECK_ARG(C % N == 0 && ldx % N == 0 && ld_out % N == 0 && C <= 64 * LGP_MAXV * N, "drn_lgp_fwd: C must be a 16-byte multiple and <= %d", 64 * LGP_MAXV * N);
    int nb = cdiv(B * (t / 2), 4);
    if (nb > 4096) nb = 4096;
    lgp_fwd_kernel<T><<<nb, 256, 0, (hipStream_t)stream>>>((const T*)x, ldx, qn, (T*)out, ld_out, att, B, t, C);
  });
  return drn_launch_status("drn_lgp_fwd");
}

// dx rows, and per-block partial sums of dq'[b,c] = sum_tau ds[b,tau] x[b,tau,c].
// grid (ceil(half/4), B): the 4 waves of a block take 4 consecutive pairs of the SAME clip.
template <typename T>
__global__ __launch_bounds__(256) void lgp_bwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ qn,
                                                      const float* __restrict__ att, const T* __restrict__ dout, int ld_dout,
                                                      T* __restrict__ dx, int ld_dx, float* __restrict__ dq_partial, int B, int t, int C) {
  constexpr int N = V16<T>::N;
  extern __shared__ float red[];   // [4][C]
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int half = t >> 1, nvec = C / N;
  const int b = blockIdx.y, j = blockIdx.x * 4 + w;
  const bool live = j < half;
  const float* q = qn + (long)b * C;
  if (live) {
    const long pr = (long)b * half + j;
    const T* r0 = x + ((long)b * t + 2 * j) * ldx;
    const T* r1 = r0 + ldx;
    const T* g = dout + pr * ld_dout;
    float x0[LGP_MAXV][N], x1[LGP_MAXV][N], gg[LGP_MAXV][N];
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int u = 0; u < LGP_MAXV; ++u) {
      const int v = l + u * 64;
      if (v < nvec) {
        V16<T>::load(r0 + v * N, x0[u]);
        V16<T>::load(r1 + v * N, x1[u]);
        V16<T>::load(g + v * N, gg[u]);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          d0 = fmaf(gg[u][k], x0[u][k], d0);
          d1 = fmaf(gg[u][k], x1[u][k], d1);
        }
      }
    }
    d0 = wave_sum(d0);
    d1 = wave_sum(d1);
    const float a0 = att[pr * 2], a1 = att[pr * 2 + 1];
    const float dot = a0 * d0 + a1 * d1;
    const float ds0 = a0 * (d0 - dot), ds1 = a1 * (d1 - dot);   // softmax backward
    T* o0 = dx + ((long)b * t + 2 * j) * ld_dx;
    T* o1 = o0 + ld_dx;
#pragma unroll
    for (int u = 0; u < LGP_MAXV; ++u) {
      const int v = l + u * 64;
      if (v < nvec) {
        float y0[N], y1[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const float qq = q[v * N + k];
          y0[k] = fmaf(gg[u][k], a0, ds0 * qq);
          y1[k] = fmaf(gg[u][k], a1, ds1 * qq);
          red[w * C + v * N + k] = ds0 * x0[u][k] + ds1 * x1[u][k];
        }
        V16<T>::store(o0 + v * N, y0);
        V16<T>::store(o1 + v * N, y1);
      }
    }
  } else {
    for (int c = l; c < C; c += 64) red[w * C + c] = 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256)
    dq_partial[((long)b * gridDim.x + blockIdx.x) * C + c] = red[c] + red[C + c] + red[2 * C + c] + red[3 * C + c];
}

// dq'[b][c] = sum_blk partial[b][blk][c]
__global__ void lgp_dq_reduce_kernel(const float* __restrict__ partial, int nblk, int B, int C, float* __restrict__ dqn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float s = 0.f;
  for (int k = 0; k < nblk; ++k) s += partial[((long)b * nblk + k) * C + c];
  dqn[i] = s;
}

extern "C" int drn_lgp_bwd(const void* x, int ldx, const float* qn, const float* att, const void* dout, int ld_dout, void* dx, int ld_dx,
                           float* dqn, float* ws /* >= B*ceil(t/8)*C floats */, int B, int t, int C, int dtype, void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(x && qn && att && dout && dx && dqn && ws && B > 0 && t > 0 && t % 2 == 0 && C > 0, "drn_lgp_bwd: bad args");
  const int nblk = cdiv(t / 2, 4);
  DISPATCH_DT(dtype, "drn_lgp_bwd", {
    constexpr int N = V16<T>::N;
    DRN_CHECK_ARG(C % N == 0 && ldx % N == 0 && ld_dout % N == 0 && ld_dx % N == 0 && C <= 64 * LGP_MAXV * N, "drn_lgp_bwd: bad C/ld");
    dim3 grid(nblk, B);
    lgp_bwd_kernel<T><<<grid, 256, 4 * C * sizeof(float), stream>>>((const T*)x, ldx, qn, att, (const T*)dout, ld_dout, (T*)dx, ld_dx, ws, B, t, C);
  });
  lgp_dq_reduce_kernel<<<cdiv(B * C, 256), 256, 0, stream>>>(ws, nblk, B, C, dqn);
  return drn_launch_status("drn_lgp_bwd");
}
