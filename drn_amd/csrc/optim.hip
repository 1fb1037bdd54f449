// Fused gradient-norm clipping + Adam over flat gradient buckets (main.py:238-243: clip_grad_norm_(0.5), Adam.step).
// Gradients (already all-reduced) and the Adam moments are flat fp32 buffers per bucket; parameters stay where
// PyTorch put them and are reached through a small device-resident segment table.  Two launches per bucket:
//   1. drn_sumsq_partials : per-block sums of g^2 (fixed order -> deterministic norm); the first call of a step
//      also advances the device-side step counter (no host scalar changes between steps -> hipGraph friendly);
//   2. drn_sumsq_finalize : ONE workgroup adds all partials of all buckets in a fixed order -> the squared global norm
//      (round 3 tried to fold this 5 us launch into (1) by arrival tickets, three ways: a returning atomic per 4096-element
//      block made the 29 us pass 144 us long -- one word takes ~90 atomics/us --, two-level tickets 64 us, <= 1024 grid-stride
//      workgroups with one ticket each 44 us: the streaming pass is at 6 TB/s only as 10887 independent fire-and-forget blocks);
//   3. drn_adam_bucket    : clip coefficient from that scalar, then m,v,p updates with torch.optim.Adam's formula
//      (bias-corrected, eps outside the sqrt, no weight decay).
#include "common.h"
#include "../../include/drn_hip.h"
#ifndef OPT_NT_TP
#define OPT_NT_TP 1
#endif
#ifndef OPT_NT
#define OPT_NT 1      // Adam moments: read once and written once per step -- non-temporal accesses (experiment: -DOPT_NT=0)
#endif

#define OPT_THREADS 256
#define OPT_ELEMS_PER_BLOCK 4096

__global__ __launch_bounds__(OPT_THREADS) void sumsq_partials_kernel(const float* __restrict__ g, long n, float* __restrict__ partials,
                                                                      int* __restrict__ step_counter) {
  __shared__ float sh[17];
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;
  const long base = (long)blockIdx.x * OPT_ELEMS_PER_BLOCK;
  float s = 0.f;
  for (int i = threadIdx.x * 4; i < OPT_ELEMS_PER_BLOCK; i += OPT_THREADS * 4) {
    const long k = base + i;
    if (k + 4 <= n) {
      const f32x4 v = *(const f32x4*)(g + k);
      s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
    } else {
      for (int e = 0; e < 4 && k + e < n; ++e) s = fmaf(g[k + e], g[k + e], s);
    }
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

extern "C" int64_t drn_opt_nblocks(int64_t n);
// The same pass with up to DRN_SUMSQ_MAX_SKIP element ranges LEFT OUT: gradients whose squared sums were produced by the kernels that
// wrote them (the prop_fc weight gradient's GEMM epilogue, the one-launch reduce of the conv weight gradients) and reach
// drn_sumsq_finalize2 as external partials -- 108 of the 153 MB of a step's gradients are not read a second time.  A block inside a
// range writes 0 and leaves; blocks clear of every range run the statements of sumsq_partials_kernel; the (few) boundary blocks test
// every element.
struct SumsqSkip {
  long lo[DRN_SUMSQ_MAX_SKIP], hi[DRN_SUMSQ_MAX_SKIP];
  int n;
};
__global__ __launch_bounds__(OPT_THREADS) void sumsq_partials_skip_kernel(const float* __restrict__ g, long n, float* __restrict__ partials,
                                                                           int* __restrict__ step_counter, const SumsqSkip K,
                                                                           const unsigned char* __restrict__ cls) {
  __shared__ float sh[17];
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;
  const long base = (long)blockIdx.x * OPT_ELEMS_PER_BLOCK, end = min(n, base + OPT_ELEMS_PER_BLOCK);
  bool inside = false, touches = false;
  if (cls) {                             // host-classified blocks (drn_sumsq_block_classes): 0 clear of every range, 1 inside one, 2 boundary
    const int c = cls[blockIdx.x];       // (walking the ranges in every block put ~1.5 us of dependent scalar loads in front of each:
    inside = c == 1;                     //  37 us for a third of the bytes the plain pass reads in 28)
    touches = c != 0;
  } else {
#pragma unroll
    for (int r = 0; r < DRN_SUMSQ_MAX_SKIP; ++r) {      // (static indices: the ranges are read from the kernel arguments once)
      inside |= r < K.n && K.lo[r] <= base && end <= K.hi[r];
      touches |= r < K.n && K.lo[r] < end && base < K.hi[r];
    }
  }
  if (inside) {
    if (threadIdx.x == 0) partials[blockIdx.x] = 0.f;
    return;
  }
  float s = 0.f;
  if (!touches) {
    for (int i = threadIdx.x * 4; i < OPT_ELEMS_PER_BLOCK; i += OPT_THREADS * 4) {
      const long k = base + i;
      if (k + 4 <= n) {
        const f32x4 v = *(const f32x4*)(g + k);
        s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
      } else {
        for (int e = 0; e < 4 && k + e < n; ++e) s = fmaf(g[k + e], g[k + e], s);
      }
    }
  } else {
    // a boundary block: the (one or two) ranges that reach into it, clipped to block-local offsets -- walking K.lo[r] with a
    // run-time r costs a dependent scalar load per test, which made these 16 blocks a 30 us tail
    int l0 = 0, h0 = 0, l1 = 0, h1 = 0, extra = 0;
#pragma unroll
    for (int r = 0; r < DRN_SUMSQ_MAX_SKIP; ++r) {
      if (r < K.n && K.lo[r] < end && base < K.hi[r]) {
        const int l = (int)(max(K.lo[r], base) - base), h = (int)(min(K.hi[r], end) - base);
        if (h0 == 0) { l0 = l; h0 = h; }
        else if (h1 == 0) { l1 = l; h1 = h; }
        else extra = 1;
      }
    }
    const int len = (int)(end - base);
    for (int k = threadIdx.x; k < len; k += OPT_THREADS) {
      bool skip = (l0 <= k && k < h0) || (l1 <= k && k < h1);
      if (extra) {                       // three or more ranges meet in one block (ranges shorter than a block)
        for (int r = 0; r < K.n; ++r) skip |= K.lo[r] <= base + k && base + k < K.hi[r];
      }
      if (!skip) s = fmaf(g[base + k], g[base + k], s);
    }
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

extern "C" int64_t drn_opt_nblocks(int64_t n) { return (n + OPT_ELEMS_PER_BLOCK - 1) / OPT_ELEMS_PER_BLOCK; }

extern "C" int drn_sumsq_block_classes(int64_t n, const int64_t* skip_lo, const int64_t* skip_hi, int nskip, unsigned char* cls_host) {
  drn_clear_status();
  DRN_CHECK_ARG(n > 0 && cls_host && nskip >= 0 && (nskip == 0 || (skip_lo && skip_hi)), "drn_sumsq_block_classes: bad args");
  const long nb = drn_opt_nblocks(n);
  for (long b = 0; b < nb; ++b) {
    const long base = b * OPT_ELEMS_PER_BLOCK, end = base + OPT_ELEMS_PER_BLOCK < n ? base + OPT_ELEMS_PER_BLOCK : n;
    bool inside = false, touches = false;
    for (int r = 0; r < nskip; ++r) {
      inside |= skip_lo[r] <= base && end <= skip_hi[r];
      touches |= skip_lo[r] < end && base < skip_hi[r];
    }
    cls_host[b] = inside ? 1 : (touches ? 2 : 0);
  }
  return DRN_OK;
}

extern "C" int drn_sumsq_partials_skip(const float* g, int64_t n, float* partials, int* step_counter, const int64_t* skip_lo,
                                       const int64_t* skip_hi, int nskip, const unsigned char* cls_dev, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(g && partials && n > 0 && nskip >= 0 && nskip <= DRN_SUMSQ_MAX_SKIP && (nskip == 0 || (skip_lo && skip_hi)),
                "drn_sumsq_partials_skip: bad args (at most %d ranges)", DRN_SUMSQ_MAX_SKIP);
  SumsqSkip K;
  memset(&K, 0, sizeof(K));
  K.n = nskip;
  for (int r = 0; r < nskip; ++r) {
    DRN_CHECK_ARG(skip_lo[r] >= 0 && skip_lo[r] <= skip_hi[r] && skip_hi[r] <= n, "drn_sumsq_partials_skip: bad range %d", r);
    K.lo[r] = skip_lo[r]; K.hi[r] = skip_hi[r];
  }
  sumsq_partials_skip_kernel<<<(int)drn_opt_nblocks(n), OPT_THREADS, 0, (hipStream_t)stream>>>(g, n, partials, step_counter, K, cls_dev);
  return drn_launch_status("drn_sumsq_partials_skip");
}

struct SumsqExt {
  const float* p[DRN_SUMSQ_MAX_EXT];
  int n[DRN_SUMSQ_MAX_EXT];
  int next;
};
__global__ __launch_bounds__(1024) void sumsq_finalize2_kernel(const float* __restrict__ partials, int n, const SumsqExt E,
                                                               float* __restrict__ out, float grad_scale) {
  __shared__ float sh[17];
  float s = 0.f;
#pragma unroll 8
  for (int i = threadIdx.x; i < n; i += 1024) s += partials[i];
#pragma unroll
  for (int e = 0; e < DRN_SUMSQ_MAX_EXT; ++e) {
    if (e < E.next) {
      const float* __restrict__ q = E.p[e];
      const int ne = E.n[e];
#pragma unroll 8
      for (int i = threadIdx.x; i < ne; i += 1024) s += q[i];
    }
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s * grad_scale * grad_scale;
}

extern "C" int drn_sumsq_finalize2(const float* partials, int npartials, const float* const* ext, const int32_t* ext_n, int next,
                                   float* total_sumsq, float grad_scale, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(partials && total_sumsq && npartials > 0 && next >= 0 && next <= DRN_SUMSQ_MAX_EXT && (next == 0 || (ext && ext_n)),
                "drn_sumsq_finalize2: bad args (at most %d external partial arrays)", DRN_SUMSQ_MAX_EXT);
  SumsqExt E;
  memset(&E, 0, sizeof(E));
  E.next = next;
  for (int e = 0; e < next; ++e) {
    DRN_CHECK_ARG(ext[e] && ext_n[e] > 0, "drn_sumsq_finalize2: bad external array %d", e);
    E.p[e] = ext[e]; E.n[e] = ext_n[e];
  }
  sumsq_finalize2_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(partials, npartials, E, total_sumsq, grad_scale);
  return drn_launch_status("drn_sumsq_finalize2");
}

extern "C" int drn_sumsq_partials(const float* g, int64_t n, float* partials, int* step_counter, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(g && partials && n > 0, "drn_sumsq_partials: bad args");
  sumsq_partials_kernel<<<(int)drn_opt_nblocks(n), OPT_THREADS, 0, (hipStream_t)stream>>>(g, n, partials, step_counter);
  return drn_launch_status("drn_sumsq_partials");
}

__global__ __launch_bounds__(1024) void sumsq_finalize_kernel(const float* __restrict__ partials, int n, float* __restrict__ out,
                                                              float grad_scale) {
  __shared__ float sh[17];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += partials[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s * grad_scale * grad_scale;       // the norm of grad_scale * g
}

extern "C" int drn_sumsq_finalize(const float* partials, int npartials, float* total_sumsq, float grad_scale, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(partials && total_sumsq && npartials > 0, "drn_sumsq_finalize: bad args");
  sumsq_finalize_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(partials, npartials, total_sumsq, grad_scale);
  return drn_launch_status("drn_sumsq_finalize");
}

struct AdamArgs {
  const float* g;           // flat gradients of this bucket
  float* m;
  float* v;
  long n;
  const long* seg_start;    // [nseg+1] prefix offsets of the tensors inside the flat buffers (device)
  float* const* p_ptr;      // [nseg] parameter base pointers (device)
  int nseg;
  const float* total_sumsq; // squared global gradient norm over ALL buckets (drn_sumsq_finalize)
  const int* blk_seg;       // [blocks] tensor index of each block's first element (host-precomputed), or NULL
  bf16_t* const* mirror;    // [nseg] bf16 copy of the tensor in the SAME element order (Linear / 1x1-conv GEMM operand), or NULL
  const int* step_counter;
  float lr, beta1, beta2, eps, max_norm;
  float grad_scale;         // g is read as grad_scale * g (1/world: the buckets hold the all-reduced SUM, see drn_amd.dist)
};

__global__ __launch_bounds__(OPT_THREADS) void adam_bucket_kernel(const AdamArgs A) {
  __shared__ int first_seg;
  const float total_norm = sqrtf(A.total_sumsq[0]);
  const bool have_tab = A.blk_seg != nullptr;
  float clip = A.max_norm > 0.f ? A.max_norm / (total_norm + 1e-6f) : 1.f;   // torch.nn.utils.clip_grad_norm_
  clip = fminf(clip, 1.f);
  const int t = *A.step_counter;
  const float bc1 = 1.f - powf(A.beta1, (float)t), bc2 = 1.f - powf(A.beta2, (float)t);
  const float step_size = A.lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const long base = (long)blockIdx.x * OPT_ELEMS_PER_BLOCK;
  if (!have_tab) {
    if (threadIdx.x == 0) {   // binary search: last segment with start <= base
      int lo = 0, hi = A.nseg - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (A.seg_start[mid] <= base) lo = mid; else hi = mid - 1;
      }
      first_seg = lo;
    }
    __syncthreads();
  }
  int seg = have_tab ? A.blk_seg[blockIdx.x] : first_seg;
  // Fast path (almost every block of the large tensors): the whole 4096-element block lies inside ONE 16-byte-aligned
  // tensor, so there are no per-quad table lookups and all 16 loads of the four trips are issued before the first use.
  {
    const long s0 = A.seg_start[seg], s1 = A.seg_start[seg + 1];
    float* pbase = as_global_v(A.p_ptr[seg]);         // (a pointer read from a table: generic to the compiler -- common.h, as_global)
    if (pbase != nullptr && base + OPT_ELEMS_PER_BLOCK <= s1 && base + OPT_ELEMS_PER_BLOCK <= A.n &&
        ((((uintptr_t)(pbase + (base - s0))) & 15) == 0)) {
      float* p = pbase + (base - s0);
      bf16_t* mp = A.mirror ? as_global_v(A.mirror[seg]) : nullptr;
      if (mp) mp += base - s0;
      f32x4 g4[4], m4[4], v4[4], p4[4];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const int i = threadIdx.x * 4 + t4 * OPT_THREADS * 4;
        g4[t4] = *(const f32x4*)(A.g + base + i);
        m4[t4] = OPT_NT ? __builtin_nontemporal_load((const f32x4*)(A.m + base + i)) : *(const f32x4*)(A.m + base + i);
        v4[t4] = OPT_NT ? __builtin_nontemporal_load((const f32x4*)(A.v + base + i)) : *(const f32x4*)(A.v + base + i);
        p4[t4] = (OPT_NT && mp) ? __builtin_nontemporal_load((const f32x4*)(p + i)) : *(const f32x4*)(p + i);      // (a parameter with a bf16 copy is read by Adam only)
      }
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const int i = threadIdx.x * 4 + t4 * OPT_THREADS * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = g4[t4][e] * A.grad_scale * clip;
          m4[t4][e] = A.beta1 * m4[t4][e] + (1.f - A.beta1) * g;
          v4[t4][e] = A.beta2 * v4[t4][e] + (1.f - A.beta2) * g * g;
          p4[t4][e] -= step_size * m4[t4][e] / (sqrtf(v4[t4][e]) * inv_sqrt_bc2 + A.eps);
        }
        if (OPT_NT) { __builtin_nontemporal_store(m4[t4], (f32x4*)(A.m + base + i)); __builtin_nontemporal_store(v4[t4], (f32x4*)(A.v + base + i)); }
        else { *(f32x4*)(A.m + base + i) = m4[t4]; *(f32x4*)(A.v + base + i) = v4[t4]; }
        if (OPT_NT && mp) __builtin_nontemporal_store(p4[t4], (f32x4*)(p + i)); else *(f32x4*)(p + i) = p4[t4];
        if (mp) {                                   // the GEMM's bf16 operand, refreshed in the same pass
          bf16x4 b;
#pragma unroll
          for (int e = 0; e < 4; ++e) b[e] = (bf16_t)p4[t4][e];
          *(bf16x4*)(mp + i) = b;
        }
      }
      return;
    }
  }
  // 4 elements per thread per trip.  Segment starts are 4-aligned in the flat buffers (drn_amd.dist pads them), so a
  // quad never straddles two tensors; only a tensor's last (partial) quad takes the scalar path.
  for (int i = threadIdx.x * 4; i < OPT_ELEMS_PER_BLOCK; i += OPT_THREADS * 4) {
    const long k = base + i;
    if (k >= A.n) break;
    while (k >= A.seg_start[seg + 1]) ++seg;
    const long s0 = A.seg_start[seg], s1 = A.seg_start[seg + 1];
    float* pbase = as_global_v(A.p_ptr[seg]);
    if (pbase == nullptr) continue;                       // padding between tensors
    float* p = pbase + (k - s0);
    bf16_t* mq = (A.mirror && A.mirror[seg]) ? as_global_v(A.mirror[seg]) + (k - s0) : nullptr;
    if (k + 4 <= s1 && k + 4 <= A.n && (((uintptr_t)p) & 15) == 0) {
      const f32x4 g4 = *(const f32x4*)(A.g + k);
      f32x4 m4 = *(const f32x4*)(A.m + k), v4 = *(const f32x4*)(A.v + k), p4 = *(const f32x4*)p;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = g4[e] * A.grad_scale * clip;
        m4[e] = A.beta1 * m4[e] + (1.f - A.beta1) * g;
        v4[e] = A.beta2 * v4[e] + (1.f - A.beta2) * g * g;
        p4[e] -= step_size * m4[e] / (sqrtf(v4[e]) * inv_sqrt_bc2 + A.eps);
      }
      *(f32x4*)(A.m + k) = m4;
      *(f32x4*)(A.v + k) = v4;
      *(f32x4*)p = p4;
      if (mq)
#pragma unroll
        for (int e = 0; e < 4; ++e) mq[e] = (bf16_t)p4[e];
    } else {
      for (int e = 0; e < 4 && k + e < s1 && k + e < A.n; ++e) {
        const float g = A.g[k + e] * A.grad_scale * clip;
        const float m = A.beta1 * A.m[k + e] + (1.f - A.beta1) * g;
        const float v = A.beta2 * A.v[k + e] + (1.f - A.beta2) * g * g;
        A.m[k + e] = m;
        A.v[k + e] = v;
        p[e] -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + A.eps);
        if (mq) mq[e] = (bf16_t)p[e];
      }
    }
  }
}

extern "C" int drn_adam_bucket(const float* g, float* m, float* v, int64_t n, const int64_t* seg_start_dev, float* const* p_ptr_dev,
                               int nseg, const int* blk_seg, void* const* mirror_dev, const float* total_sumsq,
                               const int* step_counter, float lr, float beta1, float beta2, float eps, float max_norm, float grad_scale,
                               void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(g && m && v && n > 0 && seg_start_dev && p_ptr_dev && nseg > 0 && total_sumsq && step_counter,
                "drn_adam_bucket: bad args");
  AdamArgs A;
  A.g = g; A.m = m; A.v = v; A.n = n; A.seg_start = (const long*)seg_start_dev; A.p_ptr = p_ptr_dev; A.nseg = nseg;
  A.total_sumsq = total_sumsq; A.step_counter = step_counter; A.blk_seg = blk_seg; A.mirror = (bf16_t* const*)mirror_dev;
  A.lr = lr; A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.max_norm = max_norm; A.grad_scale = grad_scale;
  adam_bucket_kernel<<<(int)drn_opt_nblocks(n), OPT_THREADS, 0, (hipStream_t)stream>>>(A);
  return drn_launch_status("drn_adam_bucket");
}


// ---------------------------------------------------------------------------------------------------------------------
// Adam over tensors with re-laid GEMM copies: 64-row x 64-channel tiles (see include/drn_hip.h, drn_adam_tiled).
// ---------------------------------------------------------------------------------------------------------------------
#define TILED_MAXK 3
// Division by a small run-time divisor d as one multiply-high: magic = floor(2^32 / d) + 1 is exact for q * d < 2^32 (d >= 2;
// every quotient here is below 2^16).  The tile index arithmetic below divides by the tile's row length in pieces, by the piece
// count and by the tap count -- ~150 integer divisions per thread and tile as ~40-instruction sequences made this kernel
// VALU-bound (3.8 TB/s where the linear kernel streams 5.4): one real division per divisor and thread instead.
struct FastDiv {
  unsigned magic;
  int d;
  __device__ __forceinline__ explicit FastDiv(int dd) : magic(dd > 1 ? 0xFFFFFFFFu / (unsigned)dd + 1u : 0u), d(dd) {}
  __device__ __forceinline__ int div(int q) const { return d > 1 ? (int)__umulhi((unsigned)q, magic) : q; }
};
template <typename T>
__device__ __forceinline__ void tiled_store_piece(T* dst, const float* v);
template <>
__device__ __forceinline__ void tiled_store_piece<float>(float* dst, const float* v) {
  *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
}
template <>
__device__ __forceinline__ void tiled_store_piece<bf16_t>(bf16_t* dst, const float* v) {
  bf16x8 b;
#pragma unroll
  for (int e = 0; e < 8; ++e) b[e] = (bf16_t)v[e];
  *(bf16x8*)dst = b;
}

// copy 1: [r][tap][c] -- pieces of VEC consecutive channels; copy 2: [c][tap][r] -- pieces of VEC consecutive rows
template <typename T>
__device__ __forceinline__ void tiled_flush(const float (*tile)[64 * TILED_MAXK + 1], const DrnAdamTiledItem& it, int which, int r0, int c0,
                                            int nr, int nc) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int k = it.k;
  if (which == 1) {
    T* out = (T*)it.m1;
    const bool vec = (nc % VEC == 0) && (it.ld1 % VEC == 0) && ((((uintptr_t)out) & 15) == 0);
    const int pcs = (nc + VEC - 1) / VEC;
    const FastDiv dp(pcs), dk(k);
    for (int q = threadIdx.x; q < nr * k * pcs; q += OPT_THREADS) {
      const int rt = dp.div(q), cv = q - rt * pcs, r = dk.div(rt), tap = rt - r * k;
      float vals[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) vals[e] = cv * VEC + e < nc ? tile[r][(cv * VEC + e) * k + tap] : 0.f;
      T* dst = out + ((long)(r0 + r) * k + tap) * it.ld1 + c0 + cv * VEC;
      if (vec) tiled_store_piece<T>(dst, vals);
      else
        for (int e = 0; e < VEC && cv * VEC + e < nc; ++e) DT<T>::st(dst + e, vals[e]);
    }
  } else {
    T* out = (T*)it.m2;
    const bool vec = (nr % VEC == 0) && (it.ld2 % VEC == 0) && ((((uintptr_t)out) & 15) == 0);
    const int pcs = (nr + VEC - 1) / VEC;
    const FastDiv dp(pcs), dk(k);
    for (int q = threadIdx.x; q < nc * k * pcs; q += OPT_THREADS) {
      const int ct = dp.div(q), rv = q - ct * pcs, c = dk.div(ct), tap = ct - c * k;
      float vals[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) vals[e] = rv * VEC + e < nr ? tile[rv * VEC + e][c * k + tap] : 0.f;
      T* dst = out + ((long)(c0 + c) * k + tap) * it.ld2 + r0 + rv * VEC;
      if (vec) tiled_store_piece<T>(dst, vals);
      else
        for (int e = 0; e < VEC && rv * VEC + e < nr; ++e) DT<T>::st(dst + e, vals[e]);
    }
  }
}

__global__ __launch_bounds__(OPT_THREADS) void adam_tiled_kernel(const float* __restrict__ G, float* __restrict__ Mo, float* __restrict__ Vo,
                                                                const DrnAdamTiledItem* __restrict__ items, const int* __restrict__ blk_item,
                                                                const int* __restrict__ blk_tile, const float* __restrict__ total_sumsq,
                                                                const int* __restrict__ step_counter, float lr, float beta1, float beta2,
                                                                float eps, float max_norm, float grad_scale) {
  __shared__ float tile[64][64 * TILED_MAXK + 1];
  DrnAdamTiledItem it = items[blk_item[blockIdx.x]];
  it.p = as_global(it.p); it.m1 = as_global(it.m1); it.m2 = as_global(it.m2);          // (read from a table: common.h, as_global)
  const int t = blk_tile[blockIdx.x];
  constexpr int tcw = 64;       // channels per tile (192-element tiles for k = 1 -- half as many workgroups -- measured 13 % slower)
  const int r0 = (t / it.tiles_c) * 64, c0 = (t % it.tiles_c) * tcw;
  const int nr = min(64, it.R - r0), nc = min(tcw, it.C - c0);
  const int k = it.k, S = it.C * k, ts = nc * k;            // row length of the tensor, of the tile
  const float total_norm = sqrtf(total_sumsq[0]);
  float clip = max_norm > 0.f ? max_norm / (total_norm + 1e-6f) : 1.f;
  clip = fminf(clip, 1.f);
  const int tstep = *step_counter;
  const float bc1 = 1.f - powf(beta1, (float)tstep), bc2 = 1.f - powf(beta2, (float)tstep);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const long tbase = (long)r0 * S + (long)c0 * k;             // first element of the tile inside the tensor
  const bool vec4 = (ts % 4 == 0) && (S % 4 == 0) && ((((uintptr_t)it.p) & 15) == 0) && (it.off % 4 == 0);
  if (vec4) {
    const int qpr = ts / 4, nq = nr * qpr;                    // float4 pieces per tile row, in the tile
    const FastDiv dq(qpr);
    for (int q0 = threadIdx.x; q0 < nq; q0 += 4 * OPT_THREADS) {
      f32x4 g4[4], m4[4], v4[4], p4[4];
      long e0[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                           // 16 loads in flight per trip
        const int q = q0 + u * OPT_THREADS;
        const int r = dq.div(q), c4 = (q - r * qpr) * 4;
        e0[u] = q < nq ? tbase + (long)r * S + c4 : -1;
        if (e0[u] >= 0) {
          g4[u] = *(const f32x4*)(G + it.off + e0[u]);
          m4[u] = OPT_NT ? __builtin_nontemporal_load((const f32x4*)(Mo + it.off + e0[u])) : *(const f32x4*)(Mo + it.off + e0[u]);
          v4[u] = OPT_NT ? __builtin_nontemporal_load((const f32x4*)(Vo + it.off + e0[u])) : *(const f32x4*)(Vo + it.off + e0[u]);
          p4[u] = OPT_NT_TP ? __builtin_nontemporal_load((const f32x4*)(it.p + e0[u])) : *(const f32x4*)(it.p + e0[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (e0[u] < 0) continue;
        const int q = q0 + u * OPT_THREADS;
        const int r = dq.div(q), c4 = (q - r * qpr) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = g4[u][e] * grad_scale * clip;
          m4[u][e] = beta1 * m4[u][e] + (1.f - beta1) * g;
          v4[u][e] = beta2 * v4[u][e] + (1.f - beta2) * g * g;
          p4[u][e] -= step_size * m4[u][e] / (sqrtf(v4[u][e]) * inv_sqrt_bc2 + eps);
          tile[r][c4 + e] = p4[u][e];
        }
        if (OPT_NT) { __builtin_nontemporal_store(m4[u], (f32x4*)(Mo + it.off + e0[u])); __builtin_nontemporal_store(v4[u], (f32x4*)(Vo + it.off + e0[u])); }
        else { *(f32x4*)(Mo + it.off + e0[u]) = m4[u]; *(f32x4*)(Vo + it.off + e0[u]) = v4[u]; }
        if (OPT_NT_TP) __builtin_nontemporal_store(p4[u], (f32x4*)(it.p + e0[u])); else *(f32x4*)(it.p + e0[u]) = p4[u];
      }
    }
  } else {
    const FastDiv dt(ts);
    for (int q = threadIdx.x; q < nr * ts; q += OPT_THREADS) {
      const int r = dt.div(q), c = q - r * ts;
      const long e0 = tbase + (long)r * S + c;
      const float g = G[it.off + e0] * grad_scale * clip;
      const float mm = beta1 * Mo[it.off + e0] + (1.f - beta1) * g;
      const float vv = beta2 * Vo[it.off + e0] + (1.f - beta2) * g * g;
      Mo[it.off + e0] = mm;
      Vo[it.off + e0] = vv;
      const float pp = it.p[e0] - step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
      it.p[e0] = pp;
      tile[r][c] = pp;
    }
  }
  __syncthreads();
  if (it.m1) {
    if (it.code1 == DRN_BF16) tiled_flush<bf16_t>(tile, it, 1, r0, c0, nr, nc); else tiled_flush<float>(tile, it, 1, r0, c0, nr, nc);
  }
  if (it.m2) {
    if (it.code2 == DRN_BF16) tiled_flush<bf16_t>(tile, it, 2, r0, c0, nr, nc); else tiled_flush<float>(tile, it, 2, r0, c0, nr, nc);
  }
}

extern "C" int drn_adam_tiled(const float* g, float* m, float* v, const DrnAdamTiledItem* items_dev, const int32_t* blk_item_dev,
                              const int32_t* blk_tile_dev, int nblocks, const float* total_sumsq, const int* step_counter, float lr,
                              float beta1, float beta2, float eps, float max_norm, float grad_scale, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(g && m && v && items_dev && blk_item_dev && blk_tile_dev && nblocks > 0 && total_sumsq && step_counter,
                "drn_adam_tiled: bad args");
  adam_tiled_kernel<<<nblocks, OPT_THREADS, 0, (hipStream_t)stream>>>(g, m, v, items_dev, blk_item_dev, blk_tile_dev, total_sumsq,
                                                                       step_counter, lr, beta1, beta2, eps, max_norm, grad_scale);
  return drn_launch_status("drn_adam_tiled");
}
