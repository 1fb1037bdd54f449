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

// Up to HEAD_MAX_SETS independent heads (own weights, own column slice of the level tensors, own outputs) share one launch:
// blockIdx.z selects the set.  cls_logits and bbox_pred read the two halves of the same tower output; launched together
// their workgroups fill the chip side by side instead of one half-empty launch after the other.
#define HEAD_MAX_SETS 2
struct HeadSet {
  HeadParams P;
  const float* W;      // [N][taps][C]: the re-laid fp32 copy of the nn.Conv1d weight (N, C, taps)
  const float* bias;
  float* out;
  float* z;
  const float* dout;
  float* partial;      // backward: per-row-block partial sums
  float* dW;
  float* dbias;
  float* dscale;
  int accumulate_dx, accumulate_dw, nblk, dscale_stride;
};
struct HeadMulti {
  HeadSet s[HEAD_MAX_SETS];
};

__device__ __forceinline__ int head_group_of(const HeadParams& P, int r) {
  int g = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.ngroups && r >= P.g[i].row_start) g = i;
  return g;
}

// Work decomposition shared by the three kernels: a lane owns one 16-byte channel vector (its N x taps x VN weights live
// in registers), a wavefront owns a run of consecutive rows of the level-concatenated row space.
#define HEAD_RPW 8     // rows per wavefront (forward / data gradient)

// A lane's N x taps x VN weights come from `Wt`, the fp32 copy of the nn.Conv1d weight re-laid as [N][taps][C] (kept current by the
// optimizer like every other GEMM-layout copy: drn_adam_tiled): VN consecutive floats per (n, tap) = one or two 16-byte loads
// out of a 12 KB, L2-resident table -- no LDS staging pass, no barrier in front of the activations.
template <typename T>
__device__ __forceinline__ void head_load_w(const float* __restrict__ Wt, int C, int taps, int N, int c0, bool live,
                                            float (&wr)[HEAD_MAX_N][HEAD_MAX_TAPS][V16<T>::N]) {
  constexpr int VN = V16<T>::N;
#pragma unroll
  for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
    for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp) {
      const bool on = live && n < N && tp < taps;
      const float* src = Wt + ((long)((on ? n : 0) * taps + (on ? tp : 0)) * C + (live ? c0 : 0));     // always a valid address
#pragma unroll
      for (int q = 0; q < VN / 4; ++q) {
        const f32x4 t = *(const f32x4*)(src + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) wr[n][tp][q * 4 + k] = on ? t[k] : 0.f;
      }
    }
}

// dz[r][n] = exp_mode ? scale_l * out[r][n] * dout[r][n] : dout[r][n]   (chain rule of exp(scale * z))
__device__ __forceinline__ float head_dz(const HeadParams& P, const HeadGroup& G, const float* __restrict__ dout,
                                         const float* __restrict__ out, long idx) {
  const float d = dout[idx];
  return P.exp_mode ? G.scale[0] * out[idx] * d : d;
}

// A wave's run of HEAD_RPW rows [r0, r0 + HEAD_RPW) of the level-concatenated row space takes the WINDOW path when all of them
// exist and lie in one level: the k = 3 taps of consecutive rows overlap, so rows m0-1 .. m0+HEAD_RPW (clamped into the level;
// a tap that leaves its sequence is masked per (row, tap)) are fetched once -- HEAD_RPW + 2 loads instead of 3 x HEAD_RPW.
struct HeadRun {
  int g, m0;       // level of the run, first row inside it
  bool window;
};
__device__ __forceinline__ HeadRun head_run(const HeadParams& P, int r0) {
  HeadRun R;
  R.g = 0; R.m0 = 0; R.window = false;
  if (r0 >= P.total_rows) return R;
  R.g = head_group_of(P, r0);
  R.m0 = r0 - P.g[R.g].row_start;
  R.window = P.taps == 3 && P.pad == 1 && r0 + HEAD_RPW <= P.total_rows && head_group_of(P, r0 + HEAD_RPW - 1) == R.g;
  return R;
}

// out[r][n] (and z[r][n] in exp mode); r = concatenated row over levels.  grid = ceil(rows / (4*HEAD_RPW)), block 256.
// The launch is one residency wave long, so its duration is one wave's dependency chain: the activations of a window run and the
// lane's weights are requested together (one memory trip), then FMAs and the wave reduction.
template <typename T>
__global__ __launch_bounds__(256) void head_out_fwd_kernel(const HeadMulti MS) {
  constexpr int VN = V16<T>::N;
  const HeadSet& S = MS.s[blockIdx.z];
  const HeadParams& P = S.P;
  const float* __restrict__ W = S.W;
  const float* __restrict__ bias = S.bias;
  float* __restrict__ out = S.out;
  float* __restrict__ z = S.z;
  const int N = P.N, C = P.C, taps = P.taps;
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * HEAD_RPW;   // wave-uniform: row bookkeeping on the scalar unit
  const int nvec = C / VN;
  if (r0 >= P.total_rows) return;
  const HeadRun R = head_run(P, r0);
  const T* __restrict__ Xw = (const T*)P.g[R.g].X;
  const int Mw = P.g[R.g].M, Lw = P.g[R.g].L;
  const long ldw = P.g[R.g].ldx;
  typename V16<T>::raw_t xr[HEAD_RPW + 2];
  auto fetch_window = [&](int v) {
#pragma unroll
    for (int j = 0; j < HEAD_RPW + 2; ++j) {
      const int m = min(max(R.m0 - 1 + j, 0), Mw - 1);        // clamped rows are masked below
      xr[j] = V16<T>::ldraw(Xw + (long)m * ldw + v * VN);
    }
  };
  if (R.window && lane < nvec) fetch_window(lane);      // in flight while the weights arrive
  float acc[HEAD_RPW][HEAD_MAX_N];
#pragma unroll
  for (int i = 0; i < HEAD_RPW; ++i)
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n) acc[i][n] = 0.f;
  for (int vb = 0; vb < nvec; vb += 64) {
    const int v = vb + lane;
    const bool live = v < nvec;
    float wr[HEAD_MAX_N][HEAD_MAX_TAPS][VN];
    head_load_w<T>(W, C, taps, N, v * VN, live, wr);
    if (!live) continue;
    if (R.window) {
      if (vb) fetch_window(v);
      const int t0 = R.m0 % Lw;
      float x[HEAD_RPW + 2][VN];
#pragma unroll
      for (int j = 0; j < HEAD_RPW + 2; ++j) V16<T>::cvt(xr[j], x[j]);
#pragma unroll
      for (int i = 0; i < HEAD_RPW; ++i) {
        int t = t0 + i;
        t = t >= Lw ? t - Lw * (t / Lw) : t;
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
          const int st = t + tp - 1;
          if (st >= 0 && st < Lw)
#pragma unroll
            for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
              for (int k = 0; k < VN; ++k) acc[i][n] = fmaf(x[i + tp][k], wr[n][tp][k], acc[i][n]);
        }
      }
      continue;
    }
    // four rows at a time: all 12 source addresses first, then 12 independent 16-byte loads, then the FMAs (a masked tap
    // re-reads the row itself and is skipped in the arithmetic) -- no control flow between the loads
#pragma unroll
    for (int h = 0; h < HEAD_RPW; h += 4) {
      const T* src[4][HEAD_MAX_TAPS];
      bool ok[4][HEAD_MAX_TAPS];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool rv = r0 + h + i < P.total_rows;
        const int r = rv ? r0 + h + i : P.total_rows - 1;
        const HeadGroup& G = P.g[head_group_of(P, r)];
        const int m = r - G.row_start;
        const int s = m / G.L, t = m - s * G.L;
#pragma unroll
        for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp) {
          const int st = t + tp - P.pad;
          ok[i][tp] = rv && tp < taps && st >= 0 && st < G.L;
          src[i][tp] = (const T*)G.X + (long)(s * G.L + (ok[i][tp] ? st : t)) * G.ldx + v * VN;
        }
      }
      float x[4][HEAD_MAX_TAPS][VN];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp) V16<T>::load(src[i][tp], x[i][tp]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp)
          if (ok[i][tp])
#pragma unroll
            for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
              for (int k = 0; k < VN; ++k) acc[h + i][n] = fmaf(x[i][tp][k], wr[n][tp][k], acc[h + i][n]);
    }
  }
  // All HEAD_RPW x HEAD_MAX_N dot products are reduced over the 64 lanes TOGETHER: at offsets 32, 16, 8, 4 a lane hands the
  // half of its values the partner will own to it and adds what it receives (8 + 4 + 2 + 1 shuffles), then offsets 2, 1
  // finish the single remaining value: 17 shuffles instead of 6 per value (96), same pairwise summation tree as wave_sum.
  // Value (i, n) = index i*HEAD_MAX_N + n ends in the four lanes with (lane >> 2) == index.
  float red[HEAD_RPW * HEAD_MAX_N];
#pragma unroll
  for (int i = 0; i < HEAD_RPW; ++i)
#pragma unroll
    for (int n = 0; n < HEAD_MAX_N; ++n) red[i * HEAD_MAX_N + n] = acc[i][n];
  static_assert(HEAD_RPW * HEAD_MAX_N == 16, "the multi-value reduction below is written for 16 values");
#pragma unroll
  for (int stage = 0; stage < 4; ++stage) {
    const int off = 32 >> stage, half = 8 >> stage;
    const bool upper = (lane & off) != 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < half) {
        const float send = upper ? red[k] : red[k + half];
        const float keep = upper ? red[k + half] : red[k];
        red[k] = keep + __shfl_xor(send, off, 64);
      }
  }
  float tot = red[0];
  tot += __shfl_xor(tot, 2, 64);
  tot += __shfl_xor(tot, 1, 64);
  if ((lane & 3) == 0) {
    const int idx = lane >> 2, i = idx / HEAD_MAX_N, n = idx % HEAD_MAX_N;
    const int r = r0 + i;
    if (r < P.total_rows && n < N) {
      const HeadGroup& G = P.g[head_group_of(P, r)];
      float v = tot + bias[n];
      if (P.exp_mode) {
        z[(long)r * N + n] = v;
        v = expf(G.scale[0] * v);
      }
      out[(long)r * N + n] = v;
    }
  }
}

// ---- backward: data gradient and weight-gradient partials in ONE launch (blockIdx.y < nblk_data: data gradient; the rest:
// weight gradient).  Both walk the same runs of HEAD_RPW rows per wave and read the same HEAD_RPW + 2 gradient rows
// dz[m0-1 .. m0+HEAD_RPW] (wave-uniform: scalar loads, requested before anything else).
//   data:   dX[m][c] (+)= sum_n sum_tap dz[row(s, t - tap + 1)][n] * W[n][c][tap]
//   weight: partial[wg][n][tap][c] = sum over the workgroup's rows of dz[r][n] * X[src(r, tap)][c], X through the same sliding window
//           as the forward pass (HEAD_RPW + 2 loads per run, the next run's in flight while this one's FMAs run), the four waves
//           of a workgroup combined through LDS in a fixed order; + HEAD_EXTRA values: [0..1] = sum dz[r][n] (bias gradient),
//           [2..5] = per level sum z*out*dout (scale gradient, exp mode).
#define HEAD_EXTRA 8
#define HEAD_WRUNS 2     // runs per wave in the weight-gradient part: HEAD_WRUNS * HEAD_RPW * 4 rows per workgroup

// dzw[j][n], j = 0 .. HEAD_RPW+1 <-> row m0 - 1 + j of level R.g, zero outside the level.  The window is wave-uniform, but it is
// FETCHED by the lanes -- lane 2j + n loads the operands of dz[j][n], one memory round trip for the whole window -- and then
// broadcast with v_readlane.  (As scalar loads -- round 2 -- the compiler put a branch and an `s_waitcnt lgkmcnt(0)` behind almost
// every one of the ~40 s_load_dword of a window: dozens of SERIAL trips through a scalar cache and an L2 that start every kernel
// cold, the bulk of the 33 us this launch took.)  zsum (w part, exp mode): sum over the run's rows of z * out * dout, per n.
__device__ __forceinline__ void head_dz_window(const HeadParams& P, const HeadRun& R, const float* __restrict__ dout,
                                               const float* __restrict__ out, const float* __restrict__ z,
                                               float (&dzw)[HEAD_RPW + 2][HEAD_MAX_N], float (&zsum)[HEAD_MAX_N]) {
  static_assert(HEAD_MAX_N == 2 && (HEAD_RPW + 2) * HEAD_MAX_N <= 64, "one lane per (row, n) of the window");
  const HeadGroup& G = P.g[R.g];
  const int lane = threadIdx.x & 63;
  const int j = lane >> 1, n = lane & 1;
  const int m = R.m0 - 1 + j;
  const bool in = j < HEAD_RPW + 2 && m >= 0 && m < G.M && n < P.N;
  const long idx = (long)(G.row_start + (in ? m : R.m0)) * P.N + (n < P.N ? n : 0);     // always a valid address; masked afterwards
  const float d = dout[idx];
  float v = d, zv = 0.f;
  if (P.exp_mode) {                                       // (uniform)
    const float od = out[idx] * d;
    v = G.scale[0] * od;
    if (z) zv = (in && j >= 1 && j <= HEAD_RPW) ? z[idx] * od : 0.f;
  }
  v = in ? v : 0.f;
#pragma unroll
  for (int jj = 0; jj < HEAD_RPW + 2; ++jj)
#pragma unroll
    for (int nn = 0; nn < HEAD_MAX_N; ++nn)
      dzw[jj][nn] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), jj * HEAD_MAX_N + nn));
#pragma unroll
  for (int nn = 0; nn < HEAD_MAX_N; ++nn) {
    float sacc = 0.f;
    if (z)
#pragma unroll
      for (int jj = 1; jj <= HEAD_RPW; ++jj) sacc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zv), jj * HEAD_MAX_N + nn));
    zsum[nn] = sacc;
  }
}

template <typename T>
__device__ __forceinline__ void head_bwd_data_part(const HeadSet& S, const int by) {
  constexpr int VN = V16<T>::N;
  const HeadParams& P = S.P;
  const float* __restrict__ W = S.W;
  const float* __restrict__ dout = S.dout;
  const float* __restrict__ out = S.out;
  const int accumulate = S.accumulate_dx;
  const int N = P.N, C = P.C, taps = P.taps;
  const int v = blockIdx.x * 64 + (threadIdx.x & 63);
  const int r0 = (by * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * HEAD_RPW;   // wave-uniform: the dz scalars come through the scalar cache
  if (r0 >= P.total_rows) return;                    // (wave-uniform)
  const HeadRun R = head_run(P, r0);
  float dzw[HEAD_RPW + 2][HEAD_MAX_N], zs_unused[HEAD_MAX_N];
  if (R.window) head_dz_window(P, R, dout, out, nullptr, dzw, zs_unused);     // all 64 lanes fetch: before any lane leaves
  if (v * VN >= C) return;
  float wr[HEAD_MAX_N][HEAD_MAX_TAPS][VN];
  head_load_w<T>(W, C, taps, N, v * VN, true, wr);
  if (R.window) {
    const HeadGroup& G = P.g[R.g];
    const int L = G.L;
    const int t0 = R.m0 % L;
    T* __restrict__ dX = (T*)G.dX + (long)R.m0 * G.ldx + v * VN;
#pragma unroll
    for (int i = 0; i < HEAD_RPW; ++i) {
      int t = t0 + i;
      t = t >= L ? t - L * (t / L) : t;
      float a[VN];
      if (accumulate) V16<T>::load(dX + (long)i * G.ldx, a);
      else {
#pragma unroll
        for (int k = 0; k < VN; ++k) a[k] = 0.f;
      }
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        const int to = t - tp + 1;                         // output position that read this input through tap `tp`
        if (to >= 0 && to < L)
#pragma unroll
          for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
            for (int k = 0; k < VN; ++k) a[k] = fmaf(dzw[i + 2 - tp][n], wr[n][tp][k], a[k]);
      }
      V16<T>::store(dX + (long)i * G.ldx, a);
    }
    return;
  }
  // general path (runs that straddle two levels or the end): four rows at a time
#pragma unroll
  for (int h = 0; h < HEAD_RPW; h += 4) {
    float dv[4][HEAD_MAX_TAPS][HEAD_MAX_N];
    T* dst[4];
    bool rv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rv[i] = r0 + h + i < P.total_rows;
      const int r = rv[i] ? r0 + h + i : P.total_rows - 1;
      const HeadGroup& G = P.g[head_group_of(P, r)];
      const int m = r - G.row_start;
      const int s = m / G.L, t = m - s * G.L;
      dst[i] = (T*)G.dX + (long)m * G.ldx + v * VN;
#pragma unroll
      for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp) {
        const int to = t - tp + P.pad;   // output position that read this input through tap `tp`
        const bool ok = rv[i] && tp < taps && to >= 0 && to < G.L;
        const long ro = (long)(G.row_start + s * G.L + (ok ? to : t)) * N;
#pragma unroll
        for (int n = 0; n < HEAD_MAX_N; ++n) {          // always a valid address; masked afterwards
          const float z = head_dz(P, G, dout, out, ro + (n < N ? n : 0));
          dv[i][tp][n] = (ok && n < N) ? z : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!rv[i]) continue;
      float a[VN];
      if (accumulate) V16<T>::load(dst[i], a);
      else {
#pragma unroll
        for (int k = 0; k < VN; ++k) a[k] = 0.f;
      }
#pragma unroll
      for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp)
#pragma unroll
        for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
          for (int k = 0; k < VN; ++k) a[k] = fmaf(dv[i][tp][n], wr[n][tp][k], a[k]);
      V16<T>::store(dst[i], a);
    }
  }
}

template <typename T>
__device__ __forceinline__ void head_bwd_w_part(const HeadSet& S, float* sred, const int wb) {
  constexpr int VN = V16<T>::N;
  const HeadParams& P = S.P;
  const float* __restrict__ dout = S.dout;
  const float* __restrict__ out = S.out;
  const float* __restrict__ z = S.z;
  if (wb >= S.nblk) return;
  const int N = P.N, C = P.C, taps = P.taps;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int v = blockIdx.x * 64 + lane;
  const bool live = v * VN < C;
  const long pstride = (long)N * taps * C + HEAD_EXTRA;
  float* __restrict__ prow = S.partial + (long)wb * pstride;
  float acc[HEAD_MAX_N][HEAD_MAX_TAPS][VN];
#pragma unroll
  for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
    for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp)
#pragma unroll
      for (int k = 0; k < VN; ++k) acc[n][tp][k] = 0.f;
  float ex[HEAD_EXTRA];
#pragma unroll
  for (int k = 0; k < HEAD_EXTRA; ++k) ex[k] = 0.f;
  const int rbase = (wb * 4 + w) * (HEAD_WRUNS * HEAD_RPW);
  typename V16<T>::raw_t xr[2][HEAD_RPW + 2];
  auto fetch = [&](const HeadRun& R, typename V16<T>::raw_t (&buf)[HEAD_RPW + 2]) {
    const HeadGroup& G = P.g[R.g];
    const T* __restrict__ X = (const T*)G.X;
#pragma unroll
    for (int j = 0; j < HEAD_RPW + 2; ++j) {
      const int m = min(max(R.m0 - 1 + j, 0), G.M - 1);
      buf[j] = V16<T>::ldraw(X + (long)m * G.ldx + v * VN);
    }
  };
  HeadRun Rn = head_run(P, rbase);
  if (Rn.window && live) fetch(Rn, xr[0]);
#pragma unroll
  for (int run = 0; run < HEAD_WRUNS; ++run) {
    const int r0 = rbase + run * HEAD_RPW;
    const HeadRun R = Rn;
    if (run + 1 < HEAD_WRUNS) {
      Rn = head_run(P, r0 + HEAD_RPW);
      if (Rn.window && live) fetch(Rn, xr[(run + 1) & 1]);
    }
    if (r0 >= P.total_rows) continue;
    if (R.window) {
      const HeadGroup& G = P.g[R.g];
      const int L = G.L, t0 = R.m0 % L;
      float dzw[HEAD_RPW + 2][HEAD_MAX_N], zsum[HEAD_MAX_N];
      head_dz_window(P, R, dout, out, P.exp_mode ? z : nullptr, dzw, zsum);
#pragma unroll
      for (int i = 0; i < HEAD_RPW; ++i)
#pragma unroll
        for (int n = 0; n < HEAD_MAX_N; ++n) ex[n] += dzw[i + 1][n];
#pragma unroll
      for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
        for (int l = 0; l < DRN_MAX_GROUPS; ++l)
          if (l == R.g) ex[2 + l] += zsum[n];
      if (live) {
        float x[HEAD_RPW + 2][VN];
#pragma unroll
        for (int j = 0; j < HEAD_RPW + 2; ++j) V16<T>::cvt(xr[run & 1][j], x[j]);
#pragma unroll
        for (int i = 0; i < HEAD_RPW; ++i) {
          int t = t0 + i;
          t = t >= L ? t - L * (t / L) : t;
#pragma unroll
          for (int tp = 0; tp < 3; ++tp) {
            const int st = t + tp - 1;
            if (st >= 0 && st < L)
#pragma unroll
              for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
                for (int k = 0; k < VN; ++k) acc[n][tp][k] = fmaf(dzw[i + 1][n], x[i + tp][k], acc[n][tp][k]);
          }
        }
      }
      continue;
    }
    // general path: row by row
    for (int i = 0; i < HEAD_RPW; ++i) {
      const int r = r0 + i;
      if (r >= P.total_rows) break;
      const int g = head_group_of(P, r);
      const HeadGroup& G = P.g[g];
      const int m = r - G.row_start;
      const int s = m / G.L, t = m - s * G.L;
      const T* __restrict__ X = (const T*)G.X;
      float d[HEAD_MAX_N];
#pragma unroll
      for (int n = 0; n < HEAD_MAX_N; ++n) {
        d[n] = n < N ? head_dz(P, G, dout, out, (long)r * N + n) : 0.f;
        ex[n] += d[n];
        if (P.exp_mode && n < N) {
          const float zz = z[(long)r * N + n] * out[(long)r * N + n] * dout[(long)r * N + n];
#pragma unroll
          for (int l = 0; l < DRN_MAX_GROUPS; ++l)
            if (l == g) ex[2 + l] += zz;
        }
      }
      if (!live) continue;
#pragma unroll
      for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp) {
        const int st = t + tp - P.pad;
        if (tp >= taps || st < 0 || st >= G.L) continue;
        float x[VN];
        V16<T>::load(X + (long)(s * G.L + st) * G.ldx + v * VN, x);
#pragma unroll
        for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
          for (int k = 0; k < VN; ++k) acc[n][tp][k] = fmaf(d[n], x[k], acc[n][tp][k]);
      }
    }
  }
  // the four waves of the workgroup, added in wave order
  float (*red)[64 * VN + 1] = (float (*)[64 * VN + 1])sred;
#pragma unroll
  for (int n = 0; n < HEAD_MAX_N; ++n)
#pragma unroll
    for (int tp = 0; tp < HEAD_MAX_TAPS; ++tp) {
      if (n >= N || tp >= taps) continue;          // uniform
      __syncthreads();
#pragma unroll
      for (int k = 0; k < VN; ++k) red[w][lane * VN + k] = acc[n][tp][k];
      __syncthreads();
      for (int i = threadIdx.x; i < 64 * VN; i += 256) {
        const int c = blockIdx.x * 64 * VN + i;
        if (c < C) prow[((long)n * taps + tp) * C + c] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
      }
    }
  if (blockIdx.x == 0) {   // every lane of a wave saw the same rows: lane 0 speaks for it
    __syncthreads();
    if (lane == 0)
#pragma unroll
      for (int k = 0; k < HEAD_EXTRA; ++k) red[w][k] = ex[k];
    __syncthreads();
    if (threadIdx.x < HEAD_EXTRA) prow[(long)N * taps * C + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// grid (ceil(nvec/64), nblk_data + nblk_w, heads); block 256
template <typename T>
__global__ __launch_bounds__(256) void head_out_bwd_kernel(const HeadMulti MS, const int nblk_data) {
  constexpr int VN = V16<T>::N;
  __shared__ float smem[4 * (64 * VN + 1)];
  const HeadSet& S = MS.s[blockIdx.z];
  if ((int)blockIdx.y < nblk_data) head_bwd_data_part<T>(S, blockIdx.y);
  else head_bwd_w_part<T>(S, smem, blockIdx.y - nblk_data);
}

// dW[n][c][tap] / dbias[n] / dscale[l] (+)= sum_blk partial[blk][...];  256 threads = 16 outputs x 16 lanes over the row blocks
__global__ __launch_bounds__(256) void head_out_bwd_w_final_kernel(const HeadMulti MS) {
  __shared__ float sh[16][17];
  const HeadSet& S = MS.s[blockIdx.z];
  const float* __restrict__ partial = S.partial;
  const int nblk = S.nblk, N = S.P.N, C = S.P.C, taps = S.P.taps, ngroups = S.P.ngroups, exp_mode = S.P.exp_mode;
  float* __restrict__ dW = S.dW;
  float* __restrict__ dbias = S.dbias;
  float* __restrict__ dscale = S.dscale;
  const int accumulate = S.accumulate_dw;
  const int nw = N * taps * C, total = nw + HEAD_EXTRA;
  const int oi = threadIdx.x & 15, j = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + oi;
  float s = 0.f;
  if (i < total) {
    int b = j;
    for (; b + 16 * 3 < nblk; b += 16 * 4) {      // four independent loads in flight per trip
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = partial[(long)(b + 16 * u) * total + i];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u];
    }
    for (; b < nblk; b += 16) s += partial[(long)b * total + i];
  }
  sh[oi][j] = s;
  __syncthreads();
  if (j != 0 || i >= total) return;
  s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += sh[oi][k];
  float* dst;
  if (i < nw) {
    const int c = i % C, tn = i / C, tap = tn % taps, n = tn / taps;
    dst = dW + ((long)n * C + c) * taps + tap;
  } else {
    const int e = i - nw;
    if (e < 2) {
      if (e >= N) return;
      dst = dbias + e;
    } else {
      if (!exp_mode || e - 2 >= ngroups) return;
      dst = dscale + (long)(e - 2) * S.dscale_stride;
    }
  }
  *dst = accumulate ? *dst + s : s;
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

static int fill_head_set(HeadSet& S, const DrnHeadCall& c, int dtype, bool backward, const char* who) {
  int rc = fill_head_params(S.P, c.groups, c.ngroups, c.N, c.C, c.taps, c.exp_mode, dtype, backward, who);
  if (rc) return rc;
  S.W = c.W; S.bias = c.bias; S.out = c.out; S.z = c.z; S.dout = c.dout; S.partial = c.ws; S.dW = c.dW; S.dbias = c.dbias;
  S.dscale = c.dscale; S.accumulate_dx = c.accumulate_dx; S.accumulate_dw = c.accumulate_dw;
  S.dscale_stride = c.dscale_stride > 0 ? c.dscale_stride : 1;
  S.nblk = cdiv(S.P.total_rows, 4 * HEAD_WRUNS * HEAD_RPW);      // weight-gradient workgroups = partial rows (drn_heads_ws_rows)
  if (!backward) DRN_CHECK_ARG(c.W && c.bias && c.out && (!c.exp_mode || c.z), "%s: null pointer", who);
  else DRN_CHECK_ARG(c.W && c.dout && c.dW && c.dbias && c.ws && (!c.exp_mode || (c.out && c.z && c.dscale)), "%s: null pointer", who);
  return DRN_OK;
}

extern "C" int64_t drn_heads_ws_elems(int total_rows, int N, int C, int taps) {
  return (int64_t)cdiv(total_rows, 4 * HEAD_WRUNS * HEAD_RPW) * ((int64_t)N * taps * C + HEAD_EXTRA);
}

extern "C" int drn_heads_fwd(const DrnHeadCall* calls, int ncalls, int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(calls && ncalls >= 1 && ncalls <= HEAD_MAX_SETS, "drn_heads_fwd: 1..%d heads per launch", HEAD_MAX_SETS);
  HeadMulti MS;
  memset(&MS, 0, sizeof(MS));
  int rows = 0;
  for (int i = 0; i < ncalls; ++i) {
    int rc = fill_head_set(MS.s[i], calls[i], dtype, false, "drn_heads_fwd");
    if (rc) return rc;
    rows = rows > MS.s[i].P.total_rows ? rows : MS.s[i].P.total_rows;
  }
  dim3 grid(cdiv(rows, 4 * HEAD_RPW), 1, ncalls);
  DISPATCH_DT(dtype, "drn_heads_fwd", { head_out_fwd_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(MS); });
  return drn_launch_status("drn_heads_fwd");
}

extern "C" int drn_heads_bwd(const DrnHeadCall* calls, int ncalls, int dtype, void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(calls && ncalls >= 1 && ncalls <= HEAD_MAX_SETS, "drn_heads_bwd: 1..%d heads per launch", HEAD_MAX_SETS);
  HeadMulti MS;
  memset(&MS, 0, sizeof(MS));
  int rows = 0, cmax = 0, nblk = 0, tot = 0;
  for (int i = 0; i < ncalls; ++i) {
    int rc = fill_head_set(MS.s[i], calls[i], dtype, true, "drn_heads_bwd");
    if (rc) return rc;
    const HeadSet& S = MS.s[i];
    rows = rows > S.P.total_rows ? rows : S.P.total_rows;
    cmax = cmax > S.P.C ? cmax : S.P.C;
    nblk = nblk > S.nblk ? nblk : S.nblk;
    const int t = S.P.N * S.P.taps * S.P.C + HEAD_EXTRA;
    tot = tot > t ? tot : t;
  }
  DISPATCH_DT(dtype, "drn_heads_bwd", {
    constexpr int VN = V16<T>::N;
    const int nblk_data = cdiv(rows, 4 * HEAD_RPW);
    dim3 grid(cdiv(cmax / VN, 64), nblk_data + nblk, ncalls);       // data gradient and weight-gradient partials side by side
    head_out_bwd_kernel<T><<<grid, 256, 0, stream>>>(MS, nblk_data);
  });
  head_out_bwd_w_final_kernel<<<dim3(cdiv(tot, 16), 1, ncalls), 256, 0, stream>>>(MS);
  return drn_launch_status("drn_heads_bwd");
}

// single-head entry points (one set per launch)
extern "C" int drn_head_out_fwd(const DrnHeadGroup* groups, int ngroups, const float* W, const float* bias, int N, int C, int taps,
                                int exp_mode, float* out, float* z, int dtype, void* stream) {
  DrnHeadCall c;
  memset(&c, 0, sizeof(c));
  c.groups = groups; c.ngroups = ngroups; c.W = W; c.bias = bias; c.N = N; c.C = C; c.taps = taps; c.exp_mode = exp_mode;
  c.out = out; c.z = z;
  return drn_heads_fwd(&c, 1, dtype, stream);
}

extern "C" int drn_head_out_bwd(const DrnHeadGroup* groups, int ngroups, const float* W, const float* dout, const float* out,
                                const float* z, int N, int C, int taps, int exp_mode, int accumulate_dx, float* dW, float* dbias,
                                float* dscale, int accumulate_dw, float* ws /* >= drn_heads_ws_elems() floats */, int dtype,
                                void* stream) {
  DrnHeadCall c;
  memset(&c, 0, sizeof(c));
  c.groups = groups; c.ngroups = ngroups; c.W = W; c.dout = dout; c.out = (float*)out; c.z = (float*)z; c.N = N; c.C = C; c.taps = taps;
  c.exp_mode = exp_mode; c.accumulate_dx = accumulate_dx; c.dW = dW; c.dbias = dbias; c.dscale = dscale; c.accumulate_dw = accumulate_dw;
  c.ws = ws;
  return drn_heads_bwd(&c, 1, dtype, stream);
}
