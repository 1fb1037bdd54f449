// Batch-sized dense layers of the query side (model/language_module.py:20-23,55-56; model/main_model.py:37-50):
//     Y[M][N] = X[M][K] * W[N][K]^T (+ bias) (ReLU)         with M = clips per GPU (<= 64), fp32 throughout.
// These are weight-streaming problems (a few MB of W, 32 rows of X): library GEMMs put 16-32 workgroups on them and take
// 10-25 us each.  Here a workgroup owns 16 output columns and one K slice, its 4 waves split that slice again and feed the
// exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) straight from global memory with every operand requested up front; the
// K slices meet in a deterministic second pass (no atomics).  The same kernel serves the input gradients through the
// cached transposed weight copies (dX = dY * W = dY * (W^T)^T).
#include "common.h"
#include "../../include/drn_hip.h"

#define SK_THREADS 256
#define SK_MAX_BT 4     // batch tiles of 16 rows

struct SkinnyArgs {
  const float* X;
  const float* W;
  const float* bias;
  float* Y;        // final output, or the partial buffer [ksplit][M][N] when ksplit > 1
  int ldx, ldy, M, N, K, kslice, relu, direct;
};

// grid (N/16, ksplit); wave w of a workgroup (NW = 4 or 16 waves) reduces k in [ks*kslice + w*kslice/NW, ... + kslice/NW)
template <int NBT, int KU, int NW>
__global__ __launch_bounds__(64 * NW) void skinny_nt_kernel(const SkinnyArgs A) {
  __shared__ float red[NW][NBT][64][4];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int n0 = blockIdx.x * 16, ks = blockIdx.y;
  const int kq = A.kslice / NW;
  const int row = l & 15, kc = (l >> 4) * 4;
  f32x4 acc[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) acc[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int kbeg = ks * A.kslice + w * kq;
  for (int k0 = kbeg; k0 < kbeg + kq; k0 += 16 * KU) {
    f32x4 a[KU][NBT], b[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int k = k0 + u * 16 + kc;
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) {
        const int m = bt * 16 + row;
        a[u][bt] = m < A.M ? *(const f32x4*)(A.X + (long)m * A.ldx + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      b[u] = *(const f32x4*)(A.W + (long)(n0 + row) * A.K + k);
    }
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][bt][e], b[u][e], acc[bt], 0, 0, 0);
  }
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][bt][l][r] = acc[bt][r];
  __syncthreads();
  // D layout: m = bt*16 + (l>>4)*4 + r, n = n0 + (l&15); wave w < 4 finishes register r = w of every lane
  if (w >= 4) return;
  const int r = w, n = n0 + (l & 15);
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) {
    const int m = bt * 16 + (l >> 4) * 4 + r;
    if (m >= A.M) continue;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) v += red[q][bt][l][r];
    if (A.direct) {
      if (A.bias) v += A.bias[n];
      if (A.relu) v = fmaxf(v, 0.f);
      A.Y[(long)m * A.ldy + n] = v;
    } else {
      A.Y[((long)ks * A.M + m) * A.N + n] = v;
    }
  }
}

__global__ void skinny_reduce_kernel(const float* __restrict__ part, int ksplit, int M, int N, const float* __restrict__ bias, int relu,
                                     float* __restrict__ Y, int ldy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int m = i / N, n = i - m * N;
  float v = 0.f;
  for (int s = 0; s < ksplit; ++s) v += part[(long)s * M * N + i];
  if (bias) v += bias[n];
  if (relu) v = fmaxf(v, 0.f);
  Y[(long)m * ldy + n] = v;
}

// 16 waves per workgroup split K inside the workgroup (no second pass); K is split over workgroups as well only when
// there are too few column tiles to matter otherwise
#include <stdlib.h>
// Measured on MI355X (scripts/bench_skinny.py): 4 waves per workgroup are enough; what matters is that a long K
// (>= 2048: qInput, the stacked qInput{t} input gradient, the level-0 gate's input gradient) is cut into slices of >= 256
// over up to 8 workgroups per column tile until ~256 workgroups exist -- 25 -> 14 us for N=1024, K=4096 -- while short-K
// problems stay single-pass (the second pass would cost more than it saves).  DRN_SKINNY_NW=16 selects the 16-wave variant.
static int skinny_waves(int K) {
  if (const char* e = getenv("DRN_SKINNY_NW")) return (atoi(e) == 16 && K % 256 == 0) ? 16 : 4;
  return 4;
}
static int skinny_ksplit(int N, int K) {
  if (K < 2048) return 1;
  const int per = 16 * skinny_waves(K);
  int ks = 1;
  while (ks < 8 && (N / 16) * ks < 256 && K / (2 * ks) >= 256 && K % (per * 2 * ks) == 0) ks *= 2;
  return ks;
}

extern "C" int64_t drn_skinny_ws_elems(int M, int N, int K) {
  const int ks = skinny_ksplit(N, K);
  return ks > 1 ? (int64_t)ks * M * N : 0;
}

extern "C" int drn_skinny_linear(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N, int K,
                                 int relu, float* ws, void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(X && W && Y && M > 0 && M <= 16 * SK_MAX_BT && N > 0 && K > 0, "drn_skinny_linear: bad args (M <= %d)", 16 * SK_MAX_BT);
  DRN_CHECK_ARG(N % 16 == 0 && K % 64 == 0 && ldx % 4 == 0 && (((uintptr_t)X | (uintptr_t)W) & 15) == 0,
                "drn_skinny_linear: need N %% 16 == 0, K %% 64 == 0, 16-byte aligned operands");
  const int ks = skinny_ksplit(N, K);
  DRN_CHECK_ARG(ks == 1 || ws, "drn_skinny_linear: workspace required (drn_skinny_ws_elems)");
  SkinnyArgs A;
  A.X = X; A.W = W; A.bias = bias; A.ldx = ldx; A.ldy = ldy; A.M = M; A.N = N; A.K = K; A.relu = relu;
  A.kslice = K / ks;
  A.direct = ks == 1;
  A.Y = ks == 1 ? Y : ws;
  const int nbt = cdiv(M, 16);
  const int nw = skinny_waves(K);
  const int it = A.kslice / (16 * nw);    // 16-wide K iterations per wave
  dim3 grid(N / 16, ks);
#define SK_LAUNCH(NBT, KU, NW) skinny_nt_kernel<NBT, KU, NW><<<grid, 64 * NW, 0, stream>>>(A)
#define SK_KU(NBT, NW)                               \
  do {                                               \
    if (it % 4 == 0) SK_LAUNCH(NBT, 4, NW);          \
    else if (it % 2 == 0) SK_LAUNCH(NBT, 2, NW);     \
    else SK_LAUNCH(NBT, 1, NW);                      \
  } while (0)
#define SK_NBT(NW)                                   \
  do {                                               \
    if (nbt == 1) SK_KU(1, NW);                      \
    else if (nbt == 2) SK_KU(2, NW);                 \
    else SK_KU(4, NW);                               \
  } while (0)
  if (nw == 16) SK_NBT(16);
  else SK_NBT(4);
#undef SK_NBT
#undef SK_KU
#undef SK_LAUNCH
  int rc = drn_launch_status("drn_skinny_linear");
  if (rc || ks == 1) return rc;
  skinny_reduce_kernel<<<cdiv(M * N, 256), 256, 0, stream>>>(ws, ks, M, N, bias, relu, Y, ldy);
  return drn_launch_status("drn_skinny_linear(reduce)");
}
