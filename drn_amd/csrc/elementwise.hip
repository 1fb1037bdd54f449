// HBM-bound helpers of the DRN path: casts / weight packing, position embedding
// (model/main_model.py:51-55), FPN top-down pair-sum (backward of F.interpolate nearest x2,
// model/FPN.py:63), query-gate backward (model/backbone.py:28-30), column sums (bias gradients).
// All activation tensors are channels-last; 16 bytes per lane per access.
#include "vec.h"
#include "../../include/drn_hip.h"

// ---------------------------------------------------------------- cast fp32 -> T
template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, long nvec) {
  constexpr int N = V16<T>::N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    float v[N];
#pragma unroll
    for (int j = 0; j < N; j += 4) {
      const f32x4 t = *(const f32x4*)(in + i * N + j);
      v[j] = t[0]; v[j + 1] = t[1]; v[j + 2] = t[2]; v[j + 3] = t[3];
    }
    V16<T>::store(out + i * N, v);
  }
}
template <typename T>
__global__ void cast_tail_kernel(const float* __restrict__ in, T* __restrict__ out, long start, long n) {
  const long i = start + (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) DT<T>::st(out + i, in[i]);
}

extern "C" int drn_cast(const float* in, void* out, int64_t n, int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(in && out && n >= 0, "drn_cast: bad args");
  if (n == 0) return DRN_OK;
  DISPATCH_DT(dtype, "drn_cast", {
    constexpr int N = V16<T>::N;
    const long nvec = n / N;
    if (nvec) cast_kernel<T><<<ew_blocks(nvec, 256), 256, 0, (hipStream_t)stream>>>(in, (T*)out, nvec);
    if (n - nvec * N) cast_tail_kernel<T><<<1, 64, 0, (hipStream_t)stream>>>(in, (T*)out, nvec * N, n);
  });
  return drn_launch_status("drn_cast");
}

// ---------------------------------------------------------------- 2-D transpose out[k][m] = in[m][k]
// 64 x 64 tiles through LDS: 16-byte loads along k, element scatter into the transposed tile, 16-byte stores along m.
template <typename T>
__global__ __launch_bounds__(256) void transpose2d_kernel(const T* __restrict__ in, int ld_in, T* __restrict__ out, int ld_out, int M,
                                                          int K) {
  constexpr int VN = V16<T>::N;
  constexpr int CPR = 64 / VN;                 // 16-byte chunks per 64-element tile row
  __shared__ T tile[64][64 + 2 * VN / 4 + 2];  // [k][m], odd dword pitch
  const int m0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  for (int q = threadIdx.x; q < 64 * CPR; q += 256) {
    const int r = q / CPR, cv = q % CPR;
    const int m = m0 + r, k = k0 + cv * VN;
    if (m < M && k < K) {                       // K % VN == 0: a chunk never crosses the row end
      T v[VN];
      *(uint4*)v = *(const uint4*)(in + (long)m * ld_in + k);
#pragma unroll
      for (int e = 0; e < VN; ++e) tile[cv * VN + e][r] = v[e];
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 64 * CPR; q += 256) {
    const int r = q / CPR, cv = q % CPR;       // r: k within the tile, cv: chunk of m
    const int k = k0 + r, m = m0 + cv * VN;
    if (k < K && m < M) {
      T v[VN];
#pragma unroll
      for (int e = 0; e < VN; ++e) v[e] = tile[r][cv * VN + e];
      *(uint4*)(out + (long)k * ld_out + m) = *(const uint4*)v;
    }
  }
}
extern "C" int drn_transpose2d(const void* in, int ld_in, void* out, int ld_out, int M, int K, int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(in && out && M > 0 && K > 0, "drn_transpose2d: bad args");
  DISPATCH_DT(dtype, "drn_transpose2d", {
    constexpr int VN = V16<T>::N;
    DRN_CHECK_ARG(M % VN == 0 && K % VN == 0 && ld_in % VN == 0 && ld_out % VN == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0,
                  "drn_transpose2d: dims / strides must be 16-byte multiples");
    dim3 grid(cdiv(K, 64), cdiv(M, 64));
    transpose2d_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)in, ld_in, (T*)out, ld_out, M, K);
  });
  return drn_launch_status("drn_transpose2d");
}

// ---------------------------------------------------------------- cast + transpose in one pass over the fp32 input
// out[m][k] = (T) in[m][k]  and  outT[k][m] = (T) in[m][k]: the proposal features are cast once per step anyway; writing
// the K-major copy from the same tile saves re-reading them for the prop_fc weight gradient (NT product of transposes).
// Tile: TM rows (m) x TK columns (k) per NT-thread workgroup.  A thread owns 8 consecutive source floats at a time (TM*TK/8/NT
// such units, all their 16-byte loads requested before the first use): they become one 16-byte store of `out` (bf16) or two
// (f32) and eight 2-byte column writes of the LDS tile [k][m]; the tile is then flushed as 16-byte pieces of outT rows
// (TM x 2-byte segments).  134 MB in, 2 x 67 MB out at B*T = 8192, D = 4096.
// (Four consecutive rows per thread, so that the four values of one k go to the tile as ONE 8-byte write instead of four 2-byte ones
// -- 8 LDS write instructions per thread instead of 32 -- measured the same inside the step: the pass is not LDS-bound.)
template <typename T, int TM, int TK, int NT>
__device__ __forceinline__ void cast_transpose_tile(const float* __restrict__ in, T* __restrict__ out, T* __restrict__ outT, int M, int K,
                                                    const int m0, const int k0);
// tiles_k > 0: a THROTTLED launch -- gridDim.x workgroups walk the tiles_k x tiles_m tiles in a grid-stride loop, so only gridDim.x
// workgroups are ever resident (drn_cast_transpose_throttled: the pass when it runs beside the query encoder's latency-bound launches).
template <typename T, int TM, int TK, int NT>
__global__ __launch_bounds__(NT) void cast_transpose_kernel(const float* __restrict__ in, T* __restrict__ out, T* __restrict__ outT,
                                                            int M, int K, int tiles_k = 0, int tiles_m = 0) {
  if (tiles_k > 0) {
    for (int t = blockIdx.x; t < tiles_k * tiles_m; t += gridDim.x) {
      cast_transpose_tile<T, TM, TK, NT>(in, out, outT, M, K, (t / tiles_k) * TM, (t % tiles_k) * TK);
      __syncthreads();
    }
    return;
  }
  cast_transpose_tile<T, TM, TK, NT>(in, out, outT, M, K, blockIdx.y * TM, blockIdx.x * TK);
}
template <typename T, int TM, int TK, int NT>
__device__ __forceinline__ void cast_transpose_tile(const float* __restrict__ in, T* __restrict__ out, T* __restrict__ outT,
                                                    int M, int K, const int m0, const int k0) {
  constexpr int VN = V16<T>::N;              // elements per 16-byte piece of the outputs
  constexpr int PITCH = TM + 16 / (int)sizeof(T);          // [k][m] tile, rows stay 16-byte aligned
  constexpr int CPR = TM / VN;                               // 16-byte pieces per tile row (a power of two)
  constexpr int UPR = TK / 8;                                // 8-float units per source row of the tile
  constexpr int NU = TM * UPR / NT;                          // units per thread
  __shared__ __attribute__((aligned(16))) T tile[TK][PITCH];
  f32x4 v[NU][2];
  int mm[NU], kk[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {                            // unit = 8 floats; all 2*NU 16-byte loads requested before the first use
    const int q = threadIdx.x + NT * u;
    mm[u] = q / UPR;
    kk[u] = (q % UPR) * 8;
    const int m = m0 + mm[u], k = k0 + kk[u];
    const bool ok = m < M && k < K;                         // K % 8 == 0 (checked by the host)
    const float* src = in + (long)m * K + k;
    v[u][0] = ok ? __builtin_nontemporal_load((const f32x4*)src) : (f32x4){0.f, 0.f, 0.f, 0.f};          // (the fp32 features are read once)
    v[u][1] = ok ? __builtin_nontemporal_load((const f32x4*)(src + 4)) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int m = m0 + mm[u], k = k0 + kk[u];
    T t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      DT<T>::st(&t[e], v[u][e >> 2][e & 3]);
      // (16-byte piece index XOR row group: the lanes of a source row write tile rows a multiple of 128 bytes apart)
      tile[kk[u] + e][((((mm[u] / VN) ^ (kk[u] >> 3)) & (CPR - 1)) * VN) + (mm[u] % VN)] = t[e];
    }
    if (m < M && k < K) {
      T* dst = out + (long)m * K + k;
#pragma unroll
      for (int h = 0; h < 8 / VN; ++h) *(uint4*)(dst + h * VN) = *(const uint4*)(t + h * VN);
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < TK * CPR; q += NT) {
    const int r = q / CPR, cv = q % CPR;
    const int k = k0 + r, m = m0 + cv * VN;
    if (k < K && m < M)        // (the transposed copy is read ~1.5 ms later, by prop_fc's weight gradient: keep it out of the caches now)
      __builtin_nontemporal_store(*(const f32x4*)&tile[r][((cv ^ (r >> 3)) & (CPR - 1)) * VN], (f32x4*)(outT + (long)k * M + m));
  }
}
extern "C" int drn_cast_transpose_throttled(const float* in, void* out, void* outT, int M, int K, int dtype, int max_workgroups, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(in && out && outT && M > 0 && K > 0 && max_workgroups > 0, "drn_cast_transpose_throttled: bad args");
  DISPATCH_DT(dtype, "drn_cast_transpose_throttled", {
    constexpr int VN = V16<T>::N;
    DRN_CHECK_ARG(M % VN == 0 && K % 8 == 0 && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)outT) & 15) == 0,
                  "drn_cast_transpose_throttled: M must be a 16-byte multiple in the output type, K a multiple of 8");
    const int tk = cdiv(K, 128), tm = cdiv(M, 64);
    const long tiles = (long)tk * tm;
    cast_transpose_kernel<T, 64, 128, 256><<<(int)(tiles < max_workgroups ? tiles : max_workgroups), 256, 0, (hipStream_t)stream>>>(
        in, (T*)out, (T*)outT, M, K, tk, tm);
  });
  return drn_launch_status("drn_cast_transpose_throttled");
}

extern "C" int drn_cast_transpose(const float* in, void* out, void* outT, int M, int K, int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(in && out && outT && M > 0 && K > 0, "drn_cast_transpose: bad args");
  DISPATCH_DT(dtype, "drn_cast_transpose", {
    constexpr int VN = V16<T>::N;
    DRN_CHECK_ARG(M % VN == 0 && K % 8 == 0 && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)outT) & 15) == 0,
                  "drn_cast_transpose: M must be a 16-byte multiple in the output type, K a multiple of 8");
    // 64 x 128 tiles: 512-byte source row segments.  Measured at B*T = 8192, D = 4096 with cold caches (scripts/bench_ew.py):
    // 128 x 64 tiles (256-byte segments) 60 us, 64 x 128 53.5, 128 x 128 54, 64 x 256 56.5, 32 x 128 (64-byte outT pieces) 84
    cast_transpose_kernel<T, 64, 128, 256><<<dim3(cdiv(K, 128), cdiv(M, 64)), 256, 0, (hipStream_t)stream>>>(in, (T*)out, (T*)outT, M, K);
  });
  return drn_launch_status("drn_cast_transpose");
}

// ---------------------------------------------------------------- weight packing (permute + cast)
// out[a][b][c] = in[a*sa + b*sb + c*sc]
template <typename T>
__global__ void pack_kernel(const float* __restrict__ in, T* __restrict__ out, int A, int B, int C, long sa, long sb, long sc) {
  const long total = (long)A * B * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long ab = i / C;
    const int b = (int)(ab % B);
    const long a = ab / B;
    DT<T>::st(out + i, in[a * sa + b * sb + c * sc]);
  }
}

extern "C" int drn_pack_weight(const float* in, void* out, int A, int B, int C, int64_t sa, int64_t sb, int64_t sc, int dtype,
                               void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(in && out && A > 0 && B > 0 && C > 0, "drn_pack_weight: bad args");
  DISPATCH_DT(dtype, "drn_pack_weight", {
    pack_kernel<T><<<ew_blocks((long)A * B * C, 256), 256, 0, (hipStream_t)stream>>>(in, (T*)out, A, B, C, sa, sb, sc);
  });
  return drn_launch_status("drn_pack_weight");
}

// Several weights in one launch (the whole model's GEMM operands after an optimizer step): per-launch cost dominates
// these small tensors, so the items ride in the kernel-argument block and workgroups are dealt out by size.
#define DRN_PACK_MAX 24
struct PackItem {
  const float* in;
  void* out;
  long sa, sb, sc, ldo;   // ldo: elements between consecutive (a,b) rows of out (C when contiguous)
  int A, B, C, blk_start;
};
struct PackParams {
  PackItem it[DRN_PACK_MAX];
  int n, total_blocks;
};

template <typename T>
__global__ void __launch_bounds__(256) pack_multi_kernel(PackParams P) {
  __shared__ float tile[64][65];
  int g = 0;
  for (int i = 1; i < P.n; ++i)
    if ((int)blockIdx.x >= P.it[i].blk_start) g = i;
  const PackItem& it = P.it[g];
  const int nblk = (g + 1 < P.n ? P.it[g + 1].blk_start : P.total_blocks) - it.blk_start;
  const int lb = blockIdx.x - it.blk_start;
  T* out = (T*)it.out;
  if (it.sb == 1 && it.sa == it.B && it.sc == (long)it.A * it.B && (it.C & 3) == 0 && (((long)it.A * it.B) & 3) == 0) {
    // data-gradient layout of a contiguous (Cout, Cin, k) weight: a plain 2-D transpose (R x S) -> (S x R) with
    // R = Cout, S = Cin*k; 64x64 tiles through LDS so that both the fp32 reads and the T writes are coalesced
    const int R = it.C, S = it.A * it.B;
    const int ts = (S + 63) >> 6, tr = (R + 63) >> 6;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    for (int t = lb; t < ts * tr; t += nblk) {
      const int r0 = (t / ts) << 6, s0 = (t % ts) << 6;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 16 * j, sx = s0 + tx * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && sx < S) v = *(const float4*)(it.in + (long)r * S + sx);
        tile[ty + 16 * j][tx * 4 + 0] = v.x; tile[ty + 16 * j][tx * 4 + 1] = v.y;
        tile[ty + 16 * j][tx * 4 + 2] = v.z; tile[ty + 16 * j][tx * 4 + 3] = v.w;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sl = ty + 16 * j, sx = s0 + sl, r = r0 + tx * 4;
        if (sx < S && r < R) {
          T* o = out + (long)sx * it.ldo + r;
          DT<T>::st(o + 0, tile[tx * 4 + 0][sl]); DT<T>::st(o + 1, tile[tx * 4 + 1][sl]);
          DT<T>::st(o + 2, tile[tx * 4 + 2][sl]); DT<T>::st(o + 3, tile[tx * 4 + 3][sl]);
        }
      }
      __syncthreads();
    }
    return;
  }
  const long total = (long)it.A * it.B * it.C;
  for (long i = (long)lb * blockDim.x + threadIdx.x; i < total; i += (long)nblk * blockDim.x) {
    const int c = (int)(i % it.C);
    const long ab = i / it.C;
    const int b = (int)(ab % it.B);
    const long a = ab / it.B;
    DT<T>::st(out + ab * it.ldo + c, it.in[a * it.sa + b * it.sb + c * it.sc]);
  }
}

extern "C" int drn_pack_weights(const DrnPackDesc* d, int n, int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(d && n > 0, "drn_pack_weights: bad args");
  for (int i = 0; i < n; ++i)
    DRN_CHECK_ARG(d[i].in && d[i].out && d[i].A > 0 && d[i].B > 0 && d[i].C > 0, "drn_pack_weights: bad item");
  for (int base = 0; base < n; base += DRN_PACK_MAX) {
    PackParams P;
    P.n = n - base < DRN_PACK_MAX ? n - base : DRN_PACK_MAX;
    int blocks = 0;
    for (int i = 0; i < P.n; ++i) {
      const DrnPackDesc& s = d[base + i];
      PackItem& it = P.it[i];
      it.in = s.in; it.out = s.out; it.sa = s.sa; it.sb = s.sb; it.sc = s.sc; it.A = s.A; it.B = s.B; it.C = s.C;
      it.ldo = s.ldo > 0 ? s.ldo : s.C;
      it.blk_start = blocks;
      const long total = (long)s.A * s.B * s.C;
      long nb = (total + 2047) / 2048;            // 8 elements per thread
      blocks += (int)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb));
    }
    P.total_blocks = blocks;
    DISPATCH_DT(dtype, "drn_pack_weights", { pack_multi_kernel<T><<<blocks, 256, 0, (hipStream_t)stream>>>(P); });
  }
  return drn_launch_status("drn_pack_weights");
}

// ---------------------------------------------------------------- position embedding
// out[m][j] = W[j][0]*f0 + W[j][1]*f1 + W[j][2]*f2 + b[j]   (nn.Linear(3,256), main_model.py:34,55)
template <typename T>
__global__ void pos_embed_kernel(const float* __restrict__ feat, const float* __restrict__ W, const float* __restrict__ b,
                                 T* __restrict__ out, int ld, int M, int C) {
  const long total = (long)M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % C);
    const long m = i / C;
    const float* f = feat + m * 3;
    // same association order as a row-times-matrix product: ((f0*w0 + f1*w1) + f2*w2) + b
    float v = f[0] * W[j * 3 + 0];
    v = fmaf(f[1], W[j * 3 + 1], v);
    v = fmaf(f[2], W[j * 3 + 2], v);
    DT<T>::st(out + m * ld + j, v + b[j]);
  }
}
extern "C" int drn_pos_embed_fwd(const float* feat, const float* W, const float* b, void* out, int ld_out, int M, int C, int dtype,
                                 void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(feat && W && b && out && M > 0 && C > 0, "drn_pos_embed_fwd: bad args");
  DISPATCH_DT(dtype, "drn_pos_embed_fwd",
              { pos_embed_kernel<T><<<ew_blocks((long)M * C, 256), 256, 0, (hipStream_t)stream>>>(feat, W, b, (T*)out, ld_out, M, C); });
  return drn_launch_status("drn_pos_embed_fwd");
}

// Stream `bytes` of a buffer through the caches at HBM speed (16-byte loads, nothing written): the bf16 copy of the
// prop_fc weight is 2.9 ms old when the next forward needs it and long evicted from the 256 MB Infinity Cache; pulling its
// 32 MB back in right before the GEMM costs ~8 us and saves the GEMM ~29 us of first-touch latency (296 -> 267 us).
__global__ __launch_bounds__(256) void touch_kernel(const uint4* __restrict__ p, long n16) {
  unsigned acc = 0;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += 4 * stride) {      // four loads in flight per trip
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = i + u * stride < n16 ? p[i + u * stride] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  asm volatile("" : : "v"(acc));      // keeps the loads alive; nothing is written (no sink buffer: the library allocates nothing)
}
extern "C" int drn_touch(const void* p, int64_t bytes, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(p && bytes >= 0 && (((uintptr_t)p) & 15) == 0, "drn_touch: bad args (16-byte aligned buffer expected)");
  if (bytes < 16) return DRN_OK;
  const long n16 = bytes / 16;
  touch_kernel<<<ew_blocks((n16 + 3) / 4, 256), 256, 0, (hipStream_t)stream>>>((const uint4*)p, n16);   // one trip of 4 loads per thread
  return drn_launch_status("drn_touch");
}

// feat[m] = (float)[start, end, end - start] from the proposal boundaries (model/main_model.py:51-55 builds it with a
// subtraction, a cat and a cast: three launches of a few hundred elements each)
template <typename S>
__global__ void pos_feat_kernel(const S* __restrict__ se, float* __restrict__ feat, int M) {
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
    const S s0 = se[m * 2], e0 = se[m * 2 + 1];
    feat[m * 3 + 0] = (float)s0;
    feat[m * 3 + 1] = (float)e0;
    feat[m * 3 + 2] = (float)(e0 - s0);          // difference in the input precision, then the cast: as the reference
  }
}
extern "C" int drn_pos_feat(const void* start_end, int is_f64, float* feat, int M, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(start_end && feat && M > 0, "drn_pos_feat: bad args");
  if (is_f64) pos_feat_kernel<double><<<ew_blocks(M, 256), 256, 0, (hipStream_t)stream>>>((const double*)start_end, feat, M);
  else pos_feat_kernel<float><<<ew_blocks(M, 256), 256, 0, (hipStream_t)stream>>>((const float*)start_end, feat, M);
  return drn_launch_status("drn_pos_feat");
}

// partial[blk][k][j], k<4: sum_m dOut[m][j] * {f0,f1,f2,1}; block = 32 rows x all channels, 8 row lanes per channel vector
template <typename T>
__global__ __launch_bounds__(256) void pos_embed_bwd_kernel(const T* __restrict__ dout, int ld, const float* __restrict__ feat, int M,
                                                            int C, float* __restrict__ partial) {
  __shared__ float red[8][4][33];
  const int rows_per = (M + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
  const int jl = threadIdx.x & 31, ry = threadIdx.x >> 5;
  for (int j0 = 0; j0 < C; j0 += 32) {
    const int j = j0 + jl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (j < C)
      for (int m = r0 + ry; m < r1; m += 8) {
        const float g = DT<T>::ld(dout + (long)m * ld + j);
        a0 = fmaf(g, feat[m * 3 + 0], a0);
        a1 = fmaf(g, feat[m * 3 + 1], a1);
        a2 = fmaf(g, feat[m * 3 + 2], a2);
        a3 += g;
      }
    red[ry][0][jl] = a0; red[ry][1][jl] = a1; red[ry][2][jl] = a2; red[ry][3][jl] = a3;
    __syncthreads();
    if (threadIdx.x < 128) {
      const int k = threadIdx.x >> 5;
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sum += red[r][k][jl];
      if (j < C) partial[((long)blockIdx.x * 4 + k) * C + j] = sum;
    }
    __syncthreads();
  }
}
// The same sums with 16-byte channel vectors: 32 vectors x 8 row lanes, FOUR rows per thread requested before the first use
// (the scalar kernel above walks 32 channels at a time behind two barriers each: 8 dependent round trips for C = 256).
template <typename T>
__global__ __launch_bounds__(256) void pos_embed_bwd_vec_kernel(const T* __restrict__ dout, int ld, const float* __restrict__ feat,
                                                                int M, int C, float* __restrict__ partial) {
  constexpr int N = V16<T>::N;
  __shared__ float red[8][4][N * 33];
  const int rows_per = (M + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
  const int vl = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int nvec = C / N;
  for (int v0 = 0; v0 < nvec; v0 += 32) {
    const int v = v0 + vl;
    float a[4][N];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < N; ++e) a[k][e] = 0.f;
    if (v < nvec)
      for (int m = r0 + ry; m < r1; m += 32) {
        float g[4][N], f[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int mm = min(m + 8 * i, r1 - 1);
          V16<T>::load(dout + (long)mm * ld + v * N, g[i]);
          f[i][0] = feat[mm * 3 + 0]; f[i][1] = feat[mm * 3 + 1]; f[i][2] = feat[mm * 3 + 2];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float w = m + 8 * i < r1 ? 1.f : 0.f;
#pragma unroll
          for (int e = 0; e < N; ++e) {
            const float ge = g[i][e] * w;
            a[0][e] = fmaf(ge, f[i][0], a[0][e]);
            a[1][e] = fmaf(ge, f[i][1], a[1][e]);
            a[2][e] = fmaf(ge, f[i][2], a[2][e]);
            a[3][e] += ge;
          }
        }
      }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < N; ++e) red[ry][k][e * 33 + vl] = a[k][e];
    __syncthreads();
    for (int o = threadIdx.x; o < 4 * 32 * N; o += 256) {
      const int k = o / (32 * N), c = o % (32 * N);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sum += red[r][k][(c % N) * 33 + c / N];
      if (v0 * N + c < C) partial[((long)blockIdx.x * 4 + k) * C + v0 * N + c] = sum;
    }
    __syncthreads();
  }
}
// block = 32 channels x 8 partial lanes
__global__ __launch_bounds__(256) void pos_embed_bwd_final_kernel(const float* __restrict__ partial, int nblk, int C,
                                                                  float* __restrict__ dW, float* __restrict__ db, int accumulate) {
  // grid (C/32, 4): blockIdx.y = which of the four sums (f0, f1, f2 weights / bias); 32 channels x 8 lanes over the blocks
  __shared__ float red[8][33];
  const int jl = threadIdx.x & 31, by = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + jl;
  const int k = blockIdx.y;
  float s = 0.f;
  if (j < C) {
#pragma unroll 8
    for (int b = by; b < nblk; b += 8) s += partial[((long)b * 4 + k) * C + j];
  }
  red[by][jl] = s;
  __syncthreads();
  if (threadIdx.x < 32 && j < C) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) sum += red[r][jl];
    float* dst = k < 3 ? dW + j * 3 + k : db + j;
    *dst = accumulate ? *dst + sum : sum;
  }
}
extern "C" int drn_pos_embed_bwd(const void* dout, int ld, const float* feat, int M, int C, float* dW, float* db, int accumulate,
                                 float* ws /* >= 256*4*C floats */, int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(dout && feat && dW && db && ws && M > 0 && C > 0, "drn_pos_embed_bwd: bad args");
  const int nblk = M < 64 ? 1 : (M < 32 * 256 ? (M + 31) / 32 : 256);   // (128-row blocks: 64 workgroups, 37 us instead of 12)
  DISPATCH_DT(dtype, "drn_pos_embed_bwd", {
    if (C % V16<T>::N == 0 && ld % V16<T>::N == 0 && ((uintptr_t)dout & 15) == 0)
      pos_embed_bwd_vec_kernel<T><<<nblk, 256, 0, (hipStream_t)stream>>>((const T*)dout, ld, feat, M, C, ws);
    else
      pos_embed_bwd_kernel<T><<<nblk, 256, 0, (hipStream_t)stream>>>((const T*)dout, ld, feat, M, C, ws);
  });
  pos_embed_bwd_final_kernel<<<dim3(cdiv(C, 32), 4), 256, 0, (hipStream_t)stream>>>(ws, nblk, C, dW, db, accumulate);
  return drn_launch_status("drn_pos_embed_bwd");
}

// ---------------------------------------------------------------- position-embedding gradient THROUGH the conv that reads it
// conv0's input is cat(gated features, position embedding) (model/backbone.py:31-32) and the embedding is a Linear(3, P) of
// the per-row features f (model/main_model.py:34,51-55).  Its weight gradient needs the conv's input gradient on those P
// channels only as a sum over rows, dWp[c][j] = sum_m dX[m][Cin-P+c] * f[m][j], and dX = sum_tap dY(shifted) x W[:, tap, c] is
// linear, so the row sum moves inside:
//   Q[tap][j][o] = sum over output rows (s, to) with t = to*stride - pad + tap inside [0, L):  dY[s,to,o] * f[s*L+t][j]   (j = 3: 1)
//   dWp[c][j]    = sum_{tap,o} Wd[Cin-P+c][tap][o] * Q[tap][j][o]          dbp[c] = the j = 3 column
// The conv's input-gradient GEMM then skips those P columns (T = 256: 544 -> 512 tiles of 256x256 = two full rounds on 256
// CUs instead of two and an eighth), and the gradient is no longer rounded to the activation dtype on the way.
// Three small launches: partial Q per row block -> Q -> the (P x 4) x (k*Cout) product.
template <typename T, int K>
__global__ __launch_bounds__(256) void conv_tail_q_kernel(const T* __restrict__ dy, int ld, const float* __restrict__ feat, int M,
                                                          int Lo, int L, int stride, int pad, int C, float* __restrict__ partial) {
  constexpr int N = V16<T>::N;
  __shared__ float red[8][4][N * 33];
  const int rows_per = (M + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
  const int vl = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int nvec = C / N;
  for (int v0 = 0; v0 < nvec; v0 += 32) {
    const int v = v0 + vl;
    float a[K][4][N];
#pragma unroll
    for (int tp = 0; tp < K; ++tp)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < N; ++e) a[tp][j][e] = 0.f;
    if (v < nvec)
      for (int m = r0 + ry; m < r1; m += 32) {
        float g[4][N], f[4][K][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool in = m + 8 * i < r1;
          const int mm = min(m + 8 * i, r1 - 1);
          V16<T>::load(dy + (long)mm * ld + v * N, g[i]);
          const int sq = mm / Lo, to = mm - sq * Lo;
#pragma unroll
          for (int tp = 0; tp < K; ++tp) {
            const int t = to * stride - pad + tp;
            const bool ok = in && t >= 0 && t < L;
            const float* fr = feat + ((long)sq * L + (ok ? t : 0)) * 3;
            f[i][tp][0] = ok ? fr[0] : 0.f;
            f[i][tp][1] = ok ? fr[1] : 0.f;
            f[i][tp][2] = ok ? fr[2] : 0.f;
            f[i][tp][3] = ok ? 1.f : 0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int tp = 0; tp < K; ++tp)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int e = 0; e < N; ++e) a[tp][j][e] = fmaf(g[i][e], f[i][tp][j], a[tp][j][e]);
      }
#pragma unroll
    for (int tp = 0; tp < K; ++tp) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < N; ++e) red[ry][j][e * 33 + vl] = a[tp][j][e];
      __syncthreads();
      for (int o = threadIdx.x; o < 4 * 32 * N; o += 256) {
        const int j = o / (32 * N), c = o % (32 * N);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) sum += red[r][j][(c % N) * 33 + c / N];
        if (v0 * N + c < C) partial[((long)blockIdx.x * (4 * K) + tp * 4 + j) * C + v0 * N + c] = sum;
      }
      __syncthreads();
    }
  }
}
// Q[q][o] = sum over row blocks; grid (C/32, 4K), 32 channels x 8 lanes over the blocks
__global__ __launch_bounds__(256) void conv_tail_qsum_kernel(const float* __restrict__ partial, int nblk, int nq, int C,
                                                             float* __restrict__ Q) {
  __shared__ float red[8][33];
  const int jl = threadIdx.x & 31, by = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + jl;
  const int q = blockIdx.y;
  float s = 0.f;
  if (j < C) {
#pragma unroll 8
    for (int b = by; b < nblk; b += 8) s += partial[((long)b * nq + q) * C + j];
  }
  red[by][jl] = s;
  __syncthreads();
  if (threadIdx.x < 32 && j < C) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) sum += red[r][jl];
    Q[(long)q * C + j] = sum;
  }
}
// one wave per embedding channel c: 16-byte pieces of its k*Cout weights against the four Q columns
template <typename T>
__global__ __launch_bounds__(256) void conv_tail_w_kernel(const T* __restrict__ Wd, long ldw, const float* __restrict__ Q, int K,
                                                          int C, int P, float* __restrict__ dW, float* __restrict__ db,
                                                          int accumulate) {
  constexpr int N = V16<T>::N;
  const int l = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= P) return;
  const T* __restrict__ w = Wd + (long)c * ldw;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int nvec = K * C / N;
  for (int v = l; v < nvec; v += 64) {
    const int kk = v * N, tp = kk / C, o = kk - tp * C;
    float wv[N];
    V16<T>::load(w + kk, wv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* q = Q + ((long)(tp * 4 + j)) * C + o;
#pragma unroll
      for (int e = 0; e < N; ++e) acc[j] = fmaf(wv[e], q[e], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) acc[j] += __shfl_xor(acc[j], sft, 64);
  if (l < 4) {
    const float val = l == 0 ? acc[0] : (l == 1 ? acc[1] : (l == 2 ? acc[2] : acc[3]));
    float* dst = l < 3 ? dW + c * 3 + l : db + c;
    *dst = accumulate ? *dst + val : val;
  }
}
extern "C" int64_t drn_conv_tail_bwd_ws_elems(int M, int k, int Cout) {
  const int nblk = M < 64 ? 1 : (M < 32 * 256 ? (M + 31) / 32 : 256);
  return ((int64_t)nblk + 1) * 4 * k * Cout;
}
extern "C" int drn_conv_tail_bwd(const void* dY, int ld_dy, int B, int Lo, int Cout, const void* Wd, int64_t ldw, int k, int stride,
                                 int pad, const float* feat, int L, int P, float* dW, float* db, int accumulate, float* ws,
                                 int dtype, void* stream) {
  drn_clear_status();
  const char* who = "drn_conv_tail_bwd";
  DRN_CHECK_ARG(dY && Wd && feat && dW && db && ws && B > 0 && Lo > 0 && L > 0 && Cout > 0 && P > 0 && stride > 0 && pad >= 0,
                "%s: bad args", who);
  DRN_CHECK_ARG(k == 1 || k == 3, "%s: kernel size %d (1 or 3)", who, k);
  const int M = B * Lo;
  const int nblk = M < 64 ? 1 : (M < 32 * 256 ? (M + 31) / 32 : 256);
  float* Q = ws + (long)nblk * 4 * k * Cout;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DT(dtype, "drn_conv_tail_bwd", {
    constexpr int N = V16<T>::N;
    DRN_CHECK_ARG(Cout % N == 0 && ld_dy % N == 0 && ldw % N == 0 && ((uintptr_t)dY & 15) == 0 && ((uintptr_t)Wd & 15) == 0,
                  "%s: Cout / ld / pointers must be 16-byte multiples", who);
    if (k == 3) conv_tail_q_kernel<T, 3><<<nblk, 256, 0, st>>>((const T*)dY, ld_dy, feat, M, Lo, L, stride, pad, Cout, ws);
    else conv_tail_q_kernel<T, 1><<<nblk, 256, 0, st>>>((const T*)dY, ld_dy, feat, M, Lo, L, stride, pad, Cout, ws);
    conv_tail_qsum_kernel<<<dim3(cdiv(Cout, 32), 4 * k), 256, 0, st>>>(ws, nblk, 4 * k, Cout, Q);
    conv_tail_w_kernel<T><<<cdiv(P, 4), 256, 0, st>>>((const T*)Wd, ldw, Q, k, Cout, P, dW, db, accumulate);
  });
  return drn_launch_status(who);
}

// ---------------------------------------------------------------- FPN top-down backward: dst[s,t] += src[s,2t] + src[s,2t+1]
template <typename T>
__global__ void pairsum_add_kernel(T* __restrict__ dst, int ld_dst, const T* __restrict__ src, int ld_src, int Mdst, int C,
                                   int accumulate, const T* __restrict__ base = nullptr, int ld_base = 0) {
  constexpr int N = V16<T>::N;
  const int nvec = C / N;
  const long total = (long)Mdst * nvec;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const long m = i / nvec;  // rows of src are exactly 2m, 2m+1 (sequence lengths are even)
    float d[N], a[N], b[N];
    if (base) V16<T>::load(base + m * ld_base + v * N, d);           // out of place: dst = base + pair sums
    else if (accumulate) V16<T>::load(dst + m * ld_dst + v * N, d);
    V16<T>::load(src + (2 * m) * ld_src + v * N, a);
    V16<T>::load(src + (2 * m + 1) * ld_src + v * N, b);
#pragma unroll
    for (int k = 0; k < N; ++k) d[k] = (accumulate || base) ? d[k] + (a[k] + b[k]) : a[k] + b[k];
    V16<T>::store(dst + m * ld_dst + v * N, d);
  }
}
extern "C" int drn_pairsum_add(void* dst, int ld_dst, const void* src, int ld_src, int Mdst, int C, int accumulate, int dtype,
                               void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(dst && src && Mdst > 0 && C > 0, "drn_pairsum_add: bad args");
  DISPATCH_DT(dtype, "drn_pairsum_add", {
    DRN_CHECK_ARG(C % V16<T>::N == 0 && ld_dst % V16<T>::N == 0 && ld_src % V16<T>::N == 0, "drn_pairsum_add: C/ld must be 16-byte multiples");
    pairsum_add_kernel<T><<<ew_blocks((long)Mdst * (C / V16<T>::N), 256), 256, 0, (hipStream_t)stream>>>((T*)dst, ld_dst, (const T*)src, ld_src, Mdst, C, accumulate);
  });
  return drn_launch_status("drn_pairsum_add");
}

// dst = base + pair sums (the same backward when the incoming gradient `base` must stay untouched: no clone + in-place add)
extern "C" int drn_pairsum_add_to(void* dst, int ld_dst, const void* base, int ld_base, const void* src, int ld_src, int Mdst,
                                  int C, int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(dst && base && src && Mdst > 0 && C > 0, "drn_pairsum_add_to: bad args");
  DISPATCH_DT(dtype, "drn_pairsum_add_to", {
    DRN_CHECK_ARG(C % V16<T>::N == 0 && ld_dst % V16<T>::N == 0 && ld_src % V16<T>::N == 0 && ld_base % V16<T>::N == 0,
                  "drn_pairsum_add_to: C/ld must be 16-byte multiples");
    pairsum_add_kernel<T><<<ew_blocks((long)Mdst * (C / V16<T>::N), 256), 256, 0, (hipStream_t)stream>>>(
        (T*)dst, ld_dst, (const T*)src, ld_src, Mdst, C, 0, (const T*)base, ld_base);
  });
  return drn_launch_status("drn_pairsum_add_to");
}

// The two steps of the three-level top-down backward (model/FPN.py:63-68) in ONE launch: d1 = own1 + pairs(d0), d2 = own2 + pairs(d1).
// A level-2 thread recomputes the two d1 rows it needs from own1 and d0 -- rounded to T where the two-launch order stores and
// re-reads them -- so the levels need no order between them and the bits are the same.
template <typename T>
__global__ __launch_bounds__(256) void pairsum_chain3_kernel(const T* __restrict__ d0, int ld0, const T* __restrict__ own1, int ldo1,
                                                             T* __restrict__ d1, int ld1, const T* __restrict__ own2, int ldo2,
                                                             T* __restrict__ d2, int ld2, int M1, int C) {
  constexpr int N = V16<T>::N;
  const int nvec = C / N, M2 = M1 / 2;
  const long n1 = (long)M1 * nvec, total = n1 + (long)M2 * nvec;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int v = (int)(i % nvec);
      const long m = i / nvec;
      float d[N], a[N], b[N];
      V16<T>::load(own1 + m * ldo1 + v * N, d);
      V16<T>::load(d0 + (2 * m) * ld0 + v * N, a);
      V16<T>::load(d0 + (2 * m + 1) * ld0 + v * N, b);
#pragma unroll
      for (int k = 0; k < N; ++k) d[k] = d[k] + (a[k] + b[k]);
      V16<T>::store(d1 + m * ld1 + v * N, d);
    } else {
      const long j = i - n1;
      const int v = (int)(j % nvec);
      const long m = j / nvec;
      float o[N], p[2][N], q[4][N];
      V16<T>::load(own2 + m * ldo2 + v * N, o);
#pragma unroll
      for (int u = 0; u < 2; ++u) V16<T>::load(own1 + (2 * m + u) * ldo1 + v * N, p[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) V16<T>::load(d0 + (4 * m + u) * ld0 + v * N, q[u]);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float da = DT<T>::round(p[0][k] + (q[0][k] + q[1][k])), db = DT<T>::round(p[1][k] + (q[2][k] + q[3][k]));
        o[k] = o[k] + (da + db);
      }
      V16<T>::store(d2 + m * ld2 + v * N, o);
    }
  }
}
extern "C" int drn_pairsum_chain3(const void* d0, int ld0, const void* own1, int ldo1, void* d1, int ld1, const void* own2, int ldo2,
                                  void* d2, int ld2, int M1, int C, int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(d0 && own1 && d1 && own2 && d2 && M1 > 0 && M1 % 2 == 0 && C > 0, "drn_pairsum_chain3: bad args");
  DISPATCH_DT(dtype, "drn_pairsum_chain3", {
    constexpr int N = V16<T>::N;
    DRN_CHECK_ARG(C % N == 0 && ld0 % N == 0 && ldo1 % N == 0 && ld1 % N == 0 && ldo2 % N == 0 && ld2 % N == 0,
                  "drn_pairsum_chain3: C/ld must be 16-byte multiples");
    pairsum_chain3_kernel<T><<<ew_blocks((long)(M1 + M1 / 2) * (C / N), 256), 256, 0, (hipStream_t)stream>>>(
        (const T*)d0, ld0, (const T*)own1, ldo1, (T*)d1, ld1, (const T*)own2, ldo2, (T*)d2, ld2, M1, C);
  });
  return drn_launch_status("drn_pairsum_chain3");
}

// ---------------------------------------------------------------- query gate, forward, as its own pass
// out[s,t,c] = z[s,t,c] * gate[s,c]  (model/backbone.py:28-30 for level 0: `q * x` on prop_fc's output).  The GEMM epilogue
// applies the gate itself when the gate exists before the GEMM starts; this pass is for the schedule that runs the query encoder
// BESIDE the prop_fc GEMM (drn_amd.graph.ForkedStep): the GEMM writes z, the gate arrives later.  16-byte vectors, 4 rows per thread.
template <typename T>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const T* __restrict__ z, int ld_z, const float* __restrict__ gate, int ldg,
                                                       T* __restrict__ out, int ld_out, int M, int L, int C) {
  constexpr int N = V16<T>::N, U = 4;
  const int nvec = C / N;
  const long total = (long)((M + U - 1) / U) * nvec;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const int m0 = (int)(i / nvec) * U;
    const int c0 = v * N;
    typename V16<T>::raw_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = V16<T>::ldraw(z + (long)min(m0 + u, M - 1) * ld_z + c0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = m0 + u;
      if (m >= M) break;
      const float* gp = gate + (long)(m / L) * ldg + c0;
      float x[N];
      V16<T>::cvt(r[u], x);
#pragma unroll
      for (int k = 0; k < N; ++k) x[k] *= gp[k];
      V16<T>::store(out + (long)m * ld_out + c0, x);
    }
  }
}
extern "C" int drn_gate_fwd(const void* z, int ld_z, const float* gate, int ldg, void* out, int ld_out, int nseq, int L, int C,
                            int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(z && gate && out && nseq > 0 && L > 0 && C > 0, "drn_gate_fwd: bad args");
  DISPATCH_DT(dtype, "drn_gate_fwd", {
    constexpr int N = V16<T>::N;
    DRN_CHECK_ARG(C % N == 0 && ld_z % N == 0 && ld_out % N == 0 && ldg % 4 == 0 && ((((uintptr_t)z) | ((uintptr_t)out) | ((uintptr_t)gate)) & 15) == 0,
                  "drn_gate_fwd: C / ld must be 16-byte multiples");
    const int M = nseq * L;
    gate_fwd_kernel<T><<<ew_blocks((long)cdiv(M, 4) * (C / N), 256, 8192), 256, 0, (hipStream_t)stream>>>((const T*)z, ld_z, gate, ldg, (T*)out,
                                                                                                        ld_out, M, L, C);
  });
  return drn_launch_status("drn_gate_fwd");
}

// ---------------------------------------------------------------- query-gate backward
// forward was G[s,t,c] = act[s,t,c] * gate[s,c].  Here:
//   dC[s,t,c] = (add ? add[s,t,c] : 0) + dG[s,t,c] * gate[s,c]        dgate[s,c] = sum_t dG[s,t,c] * act[s,t,c]
//   dsum[s,c] = sum_t dG[s,t,c] * gate[s,c]   (optional: per-sequence column sums of the gated gradient = bias gradient partials)
// grid (ceil(nvec/8), nseq); block 256 = 8 channel vectors x 32 row lanes.
template <typename T, int RY>
__global__ __launch_bounds__(8 * RY) void gate_bwd_kernel(const T* __restrict__ dG, int ld_dg, const T* __restrict__ act, int ld_act,
                                                       const float* __restrict__ gate, int ldg, const T* __restrict__ add, int ld_add,
                                                       T* __restrict__ dC, int ld_dc, float* __restrict__ dgate, int ld_dgate,
                                                       float* __restrict__ dsum, int L, int C) {
  constexpr int N = V16<T>::N;
  // 8 channel vectors (one 128-byte line of bf16) x RY row lanes per workgroup (was 32 x 8, then 16 x 16, then 8 x 32): every
  // halving of the sequential row trips shortened the launch -- at 64-128 workgroups of 16 dependent trips it was pure latency
  // (19 us for 4 MB at C = 256).  RY = 32 / 64 / 128 by sequence length: at most 4 rows per thread, all in flight at once.
  constexpr int VX = 8, NT = VX * RY;
  __shared__ float red[RY][VX * N + 1];
  const int vx = threadIdx.x & (VX - 1), ry = threadIdx.x / VX;
  const int v = blockIdx.x * VX + vx;
  const int s = blockIdx.y;
  const int c0 = v * N;
  const bool live = c0 < C;
  float gt[N], acc[N], cs[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    gt[k] = live ? gate[(long)s * ldg + c0 + k] : 0.f;
    acc[k] = 0.f;
    cs[k] = 0.f;
  }
  if (live) {
#pragma unroll 4
    for (int t = ry; t < L; t += RY) {
      const long m = (long)s * L + t;
      float g[N], a[N];
      V16<T>::load(dG + m * ld_dg + c0, g);
      V16<T>::load(act + m * ld_act + c0, a);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        acc[k] = fmaf(g[k], a[k], acc[k]);
        cs[k] += g[k];
      }
      if (dC) {
        float o[N];
        if (add) V16<T>::load(add + m * ld_add + c0, o);
#pragma unroll
        for (int k = 0; k < N; ++k) o[k] = add ? fmaf(g[k], gt[k], o[k]) : g[k] * gt[k];
        V16<T>::store(dC + m * ld_dc + c0, o);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) red[ry][vx * N + k] = acc[k];
  __syncthreads();
  for (int i = threadIdx.x; i < VX * N; i += NT) {
    const int c = blockIdx.x * VX * N + i;
    if (c < C) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < RY; ++r) sum += red[r][i];
      dgate[(long)s * ld_dgate + c] = sum;
    }
  }
  if (dsum) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) red[ry][vx * N + k] = cs[k] * gt[k];
    __syncthreads();
    for (int i = threadIdx.x; i < VX * N; i += NT) {
      const int c = blockIdx.x * VX * N + i;
      if (c < C) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < RY; ++r) sum += red[r][i];
        dsum[(long)s * C + c] = sum;
      }
    }
  }
}
extern "C" int drn_gate_bwd(const void* dG, int ld_dg, const void* act, int ld_act, const float* gate, int ldg, const void* add,
                            int ld_add, void* dC, int ld_dc, float* dgate, int ld_dgate, float* dsum, int nseq, int L, int C,
                            int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(dG && act && gate && dgate && nseq > 0 && L > 0 && C > 0 && (dC || !add), "drn_gate_bwd: bad args");
  DISPATCH_DT(dtype, "drn_gate_bwd", {
    constexpr int N = V16<T>::N;
    DRN_CHECK_ARG(C % N == 0 && ld_dg % N == 0 && ld_act % N == 0 && (!dC || ld_dc % N == 0) && (!add || ld_add % N == 0),
                  "drn_gate_bwd: C/ld must be 16-byte multiples");
    dim3 grid(cdiv(C / N, 8), nseq);
auto go = [&](auto ry) {
      constexpr int RY_ = decltype(ry)::value;
      gate_bwd_kernel<T, RY_><<<grid, 8 * RY_, 0, (hipStream_t)stream>>>((const T*)dG, ld_dg, (const T*)act, ld_act, gate, ldg, (const T*)add,
                                                                           ld_add, (T*)dC, ld_dc, dgate, ld_dgate, dsum, L, C);
    };
    if (L > 256) go(std::integral_constant<int, 128>());
    else if (L > 128) go(std::integral_constant<int, 64>());
    else go(std::integral_constant<int, 32>());
  });
  return drn_launch_status("drn_gate_bwd");
}

// Variant for the input stage: the gated gradient is only ever consumed as the K-major operand of the prop_fc weight
// gradient, so it is written TRANSPOSED, dCT[c][s*L + t] = dG[s,t,c] * gate[s,c], RI rows at a time through an LDS tile, and
// never in its natural layout.  A workgroup owns one clip and CV channel vectors (CV * N channels); 256 / CV row lanes.
//   <32, 32>: 256 (bf16) channels x 32 rows per pass -- any L % 32 == 0
//   <16, 128>: 128 channels x 128 rows per pass: 256-byte segments on BOTH sides (row reads and transposed row writes) and
//              16 row loads in flight per thread -- L % 128 == 0 (the benchmarked T = 256)
template <typename T, int CV, int RI>
__global__ __launch_bounds__(256) void gate_bwd_t_kernel(const T* __restrict__ dG, int ld_dg, const T* __restrict__ act, int ld_act,
                                                         const float* __restrict__ gate, int ldg, T* __restrict__ dCT, long ldt,
                                                         float* __restrict__ dgate, int ld_dgate, float* __restrict__ dsum, int L,
                                                         int C) {
  constexpr int N = V16<T>::N;
  constexpr int RL = 256 / CV, RPT = RI / RL;            // row lanes, rows per thread and pass
  constexpr int TP = RI + 16 / (int)sizeof(T);           // [channel][row] tile, pitch in elements (16-byte aligned rows)
  constexpr int EPC = 16 / (int)sizeof(T), CPR = RI / EPC;   // elements per 16-byte piece, pieces per tile row (a power of two)
  __shared__ float red[RL][CV * N + 1];
  __shared__ __attribute__((aligned(16))) T tile[CV * N][TP];
  const int vx = threadIdx.x % CV, ry = threadIdx.x / CV;
  const int v = blockIdx.x * CV + vx;
  const int s = blockIdx.y;
  const int c0 = v * N;
  const bool live = c0 < C;
  float gt[N], acc[N], cs[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    gt[k] = live ? gate[(long)s * ldg + c0 + k] : 0.f;
    acc[k] = 0.f;
    cs[k] = 0.f;
  }
  for (int t0 = 0; t0 < L; t0 += RI) {
    uint4 graw[RPT], araw[RPT];                            // raw 16-byte pieces: every row load of the pass is requested first
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const long m = (long)s * L + t0 + i * RL + ry;
      graw[i] = live ? *(const uint4*)(dG + m * ld_dg + c0) : make_uint4(0, 0, 0, 0);
      araw[i] = live ? *(const uint4*)(act + m * ld_act + c0) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      float g[N], a[N], o[N];
      V16<T>::load((const T*)&graw[i], g);
      V16<T>::load((const T*)&araw[i], a);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        acc[k] = fmaf(g[k], a[k], acc[k]);
        cs[k] += g[k];
        o[k] = g[k] * gt[k];
      }
      // transposed on the way in; the 16-byte piece index of a row is XORed with the channel group so that the 16 lanes of
      // a row lane (channels 8 apart = rows of the tile a multiple of 128 bytes apart) spread over all banks
      const int row = i * RL + ry;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const int ch = vx * N + k;
        DT<T>::st(&tile[ch][(((row / EPC) ^ (ch >> 3)) & (CPR - 1)) * EPC + (row % EPC)], o[k]);
      }
    }
    __syncthreads();
    // flush: consecutive lanes take consecutive 16-byte pieces of one dCT row (whole 64- / 256-byte segments per request)
    for (int q = threadIdx.x; q < CV * N * CPR; q += 256) {
      const int ch = q / CPR, part = q % CPR;
      const int c = blockIdx.x * CV * N + ch;
      if (c < C)
        *(uint4*)(dCT + (long)c * ldt + (long)s * L + t0 + part * EPC) = *(const uint4*)&tile[ch][((part ^ (ch >> 3)) & (CPR - 1)) * EPC];
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < N; ++k) red[ry][vx * N + k] = acc[k];
  __syncthreads();
  for (int i = threadIdx.x; i < CV * N; i += 256) {
    const int c = blockIdx.x * CV * N + i;
    if (c < C) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < RL; ++r) sum += red[r][i];
      dgate[(long)s * ld_dgate + c] = sum;
    }
  }
  if (dsum) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) red[ry][vx * N + k] = cs[k] * gt[k];
    __syncthreads();
    for (int i = threadIdx.x; i < CV * N; i += 256) {
      const int c = blockIdx.x * CV * N + i;
      if (c < C) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < RL; ++r) sum += red[r][i];
        dsum[(long)s * C + c] = sum;
      }
    }
  }
}
extern "C" int drn_gate_bwd_t(const void* dG, int ld_dg, const void* act, int ld_act, const float* gate, int ldg, void* dCT,
                              int64_t ldt, float* dgate, int ld_dgate, float* dsum, int nseq, int L, int C, int dtype, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(dG && act && gate && dCT && dgate && nseq > 0 && L > 0 && C > 0, "drn_gate_bwd_t: bad args");
  DRN_CHECK_ARG(L % 32 == 0, "drn_gate_bwd_t: sequence length must be a multiple of 32");
  DISPATCH_DT(dtype, "drn_gate_bwd_t", {
    constexpr int N = V16<T>::N;
    DRN_CHECK_ARG(C % N == 0 && ld_dg % N == 0 && ld_act % N == 0 && ldt % N == 0 && (((uintptr_t)dCT) & 15) == 0,
                  "drn_gate_bwd_t: C/ld must be 16-byte multiples");
    if (L % 128 == 0 && sizeof(T) == 2)
      gate_bwd_t_kernel<T, 16, 128><<<dim3(cdiv(C / N, 16), nseq), 256, 0, (hipStream_t)stream>>>(
          (const T*)dG, ld_dg, (const T*)act, ld_act, gate, ldg, (T*)dCT, ldt, dgate, ld_dgate, dsum, L, C);
    else
      gate_bwd_t_kernel<T, 32, 32><<<dim3(cdiv(C / N, 32), nseq), 256, 0, (hipStream_t)stream>>>(
          (const T*)dG, ld_dg, (const T*)act, ld_act, gate, ldg, (T*)dCT, ldt, dgate, ld_dgate, dsum, L, C);
  });
  return drn_launch_status("drn_gate_bwd_t");
}

// ---------------------------------------------------------------- column sum (bias gradients): out[c] (+)= sum_m X[m][c]
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ X, int ld, int M, int C, float* __restrict__ partial,
                                                             int direct) {
  constexpr int N = V16<T>::N;
  const int nvec = C / N;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvec) return;
  float acc[N];
#pragma unroll
  for (int k = 0; k < N; ++k) acc[k] = 0.f;
  int m = r0;
  for (; m + 8 <= r1; m += 8) {                 // 8 independent row loads in flight (one at a time, a 32-row sum took 16 us)
    float x[8][N];
#pragma unroll
    for (int u = 0; u < 8; ++u) V16<T>::load(X + (long)(m + u) * ld + v * N, x[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] += x[u][k];
  }
  for (; m < r1; ++m) {
    float x[N];
    V16<T>::load(X + (long)m * ld + v * N, x);
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] += x[k];
  }
  if (direct) {   // single row block: `partial` is the output itself
#pragma unroll
    for (int k = 0; k < N; ++k) partial[v * N + k] = direct == 2 ? partial[v * N + k] + acc[k] : acc[k];
    return;
  }
#pragma unroll
  for (int k = 0; k < N; ++k) partial[(long)blockIdx.y * C + v * N + k] = acc[k];
}
__global__ void reduce_partials_kernel(const float* __restrict__ partial, int nblk, int n, float* __restrict__ out, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += partial[(long)b * n + i];
  out[i] = accumulate ? out[i] + s : s;
}
extern "C" int drn_colsum(const void* X, int ld, int M, int C, float* out, int accumulate, float* ws /* >= 64*C floats */, int dtype,
                          void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(X && out && ws && M > 0 && C > 0, "drn_colsum: bad args");
  const int nblk = M >= 64 * 16 ? 64 : (M >= 128 ? M / 16 : 1);
  DISPATCH_DT(dtype, "drn_colsum", {
    constexpr int N = V16<T>::N;
    DRN_CHECK_ARG(C % N == 0 && ld % N == 0, "drn_colsum: C/ld must be 16-byte multiples");
    dim3 grid(cdiv(C / N, nblk == 1 ? 64 : 256), nblk);
    colsum_partial_kernel<T><<<grid, nblk == 1 ? 64 : 256, 0, (hipStream_t)stream>>>((const T*)X, ld, M, C, nblk == 1 ? out : ws,
                                                                                      nblk == 1 ? 1 + (accumulate != 0) : 0);
  });
  if (nblk > 1) reduce_partials_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(ws, nblk, C, out, accumulate);
  return drn_launch_status("drn_colsum");
}


// ---------------------------------------------------------------------------------------------------------------------------
// Up to DRN_COPY_MAX byte ranges copied (or zero-filled: src NULL) in ONE launch: the trainer's per-step refill of a captured
// step's static input buffers (tokens, lengths, features, proposal boundaries, ground truth -- five framework copies of 5-8 us each
// between two replays before).  A workgroup moves one 64 KB chunk with 16-byte accesses (ranges whose pointers or sizes are not
// 16-byte multiples take the byte path for their ragged head / tail).
struct CopyMultiParams {
  const char* src[DRN_COPY_MAX];
  char* dst[DRN_COPY_MAX];
  long bytes[DRN_COPY_MAX];
  int row_bytes[DRN_COPY_MAX];     // > 0: a 2-D range -- dst rows of dst_pitch bytes, the first row_bytes of each from src rows of
  int src_pitch[DRN_COPY_MAX];     // src_pitch bytes, the rest of the row zero (a token matrix padded out to the slot's query length)
  int dst_pitch[DRN_COPY_MAX];
  int blk0[DRN_COPY_MAX + 1];      // first workgroup of every range
  int n;
};
#define COPY_CHUNK 65536
__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyMultiParams P) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < DRN_COPY_MAX; ++k)
    if (k < P.n && (int)blockIdx.x >= P.blk0[k]) i = k;
  const long off = (long)((int)blockIdx.x - P.blk0[i]) * COPY_CHUNK;
  const long n = min((long)COPY_CHUNK, P.bytes[i] - off);
  if (P.row_bytes[i] > 0) {           // padded rows (small: byte path)
    const int rb = P.row_bytes[i], sp = P.src_pitch[i], dp = P.dst_pitch[i];
    for (long k = off + threadIdx.x; k < off + n; k += 256) {
      const long row = k / dp;
      const int col = (int)(k - row * dp);
      P.dst[i][k] = col < rb ? P.src[i][row * sp + col] : (char)0;
    }
    return;
  }
  const char* __restrict__ s = P.src[i] ? P.src[i] + off : nullptr;
  char* __restrict__ d = P.dst[i] + off;
  const bool vec = ((((uintptr_t)d) | ((uintptr_t)(s ? s : d))) & 15) == 0;
  const long nv = vec ? n / 16 : 0;
  for (long k = threadIdx.x; k < nv; k += 256) ((uint4*)d)[k] = s ? ((const uint4*)s)[k] : make_uint4(0u, 0u, 0u, 0u);
  for (long k = nv * 16 + threadIdx.x; k < n; k += 256) d[k] = s ? s[k] : (char)0;
}

extern "C" int drn_copy_multi(const void* const* srcs, void* const* dsts, const int64_t* bytes, const int32_t* rows2d, int n, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(srcs && dsts && bytes && n >= 1 && n <= DRN_COPY_MAX, "drn_copy_multi: 1..%d ranges", DRN_COPY_MAX);
  CopyMultiParams P;
  memset(&P, 0, sizeof(P));
  int blocks = 0, m = 0;
  for (int i = 0; i < n; ++i) {
    DRN_CHECK_ARG(bytes[i] >= 0 && (bytes[i] == 0 || dsts[i]), "drn_copy_multi: bad range %d", i);
    if (bytes[i] == 0) continue;
    P.src[m] = (const char*)srcs[i]; P.dst[m] = (char*)dsts[i]; P.bytes[m] = bytes[i]; P.blk0[m] = blocks;
    if (rows2d && rows2d[3 * i] > 0) {
      P.row_bytes[m] = rows2d[3 * i]; P.src_pitch[m] = rows2d[3 * i + 1]; P.dst_pitch[m] = rows2d[3 * i + 2];
      DRN_CHECK_ARG(srcs[i] && P.row_bytes[m] <= P.src_pitch[m] && P.row_bytes[m] <= P.dst_pitch[m] && bytes[i] % P.dst_pitch[m] == 0,
                    "drn_copy_multi: bad 2-D range %d", i);
    }
    blocks += (int)((bytes[i] + COPY_CHUNK - 1) / COPY_CHUNK);
    ++m;
  }
  if (m == 0) return DRN_OK;
  P.blk0[m] = blocks;
  P.n = m;
  copy_multi_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(P);
  return drn_launch_status("drn_copy_multi");
}
