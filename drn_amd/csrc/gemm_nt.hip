// Grouped implicit-GEMM (NT) on CDNA4 MFMA: Conv1d fwd / dgrad and Linear fwd / dgrad.
//
// C[M][N] = A'(M x K) * B[N][K]^T, K = taps*Cin, A' = im2col view of a channels-last
// tensor (see include/drn_hip.h).  Replaces the cuDNN/cuBLAS calls behind
// nn.Conv1d (model/basic_blocks.py:9-18, model/fcos.py:33-69) and nn.Linear
// (model/main_model.py:33,59) of the reference.
//
// Tile: 128x128 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4
// MFMA 16x16 tiles), K-step = 128 bytes per row (64 bf16 / 32 f32).  Both operands
// are staged HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip) into a
// double-buffered, XOR-swizzled image: LDS is written lane-linear, so the swizzle is
// applied to the per-lane SOURCE address and again on the ds_read_b128 (rule 21 of
// the CDNA guide).  Out-of-range rows/taps/columns read a 16-byte zero page, so
// padding, stride-2 gradients and ragged edges need no branches in the MFMA loop.
//   bf16: v_mfma_f32_16x16x32_bf16 (8 bf16 = 16 B per lane per operand)
//   f32 : 4 x v_mfma_f32_16x16x4_f32 per 16-B fragment (exact fp32, parity mode);
//         the k-permutation this implies is applied identically to A and B.
#include "common.h"
#include "../../include/drn_hip.h"

#define TILE 128
#define NT_THREADS 256
#define STAGE_BYTES 32768  // 16 KB A + 16 KB B


struct GemmProb {
  const void* A;
  const void* B;
  void* C;
  void* C2;
  const float* bias;
  const float* gate;
  float* stats;
  int M, N, K;
  int Cin, taps, stride, pad, mode;
  int Lout, Lsrc;
  int lda, ldb, ldc, ldg, ldc2;
  int accumulate;
  int tiles_n, tile_start;
};
struct GemmParams {
  int ngroups;
  GemmProb p[DRN_MAX_GROUPS];
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ void run(const frag (&a)[4], const frag (&b)[4], f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  }
};
template <> struct Mma<float> {
  typedef f32x4 frag;
  static __device__ __forceinline__ void run(const frag (&a)[4], const frag (&b)[4], f32x4 (&acc)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
  }
};

template <typename T>
__global__ __launch_bounds__(NT_THREADS, 2) void conv_gemm_nt_kernel(const GemmParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CH = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int BK = 8 * CH;               // elements per K-step (128 bytes)
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;

  int g = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.ngroups && (int)blockIdx.x >= P.p[i].tile_start) g = i;
  const GemmProb& pr = P.p[g];
  const int t_local = blockIdx.x - pr.tile_start;
  const int tm = t_local / pr.tiles_n, tn = t_local - tm * pr.tiles_n;
  const int m0 = tm * TILE, n0 = tn * TILE;
  const int M = pr.M, N = pr.N, K = pr.K, Cin = pr.Cin, taps = pr.taps;
  const int stride = pr.stride, pad = pr.pad, mode = pr.mode, Lsrc = pr.Lsrc;
  const T* __restrict__ Ag = (const T*)pr.A;
  const T* __restrict__ Bg = (const T*)pr.B;
  const T* zero = (const T*)g_zero_page;

  // ---- per-thread staging state: 4 A rows + 4 B rows (one 16-byte chunk each per K-step)
  int a_base[4], a_t[4];
  long b_off[4];
  const int pch = l & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (w * 4 + i) * 8 + (l >> 3);
    const int m = m0 + row;
    if (m < M) {
      const int seq = m / pr.Lout;
      a_t[i] = m - seq * pr.Lout;
      a_base[i] = seq * Lsrc;
    } else {
      a_t[i] = -(1 << 28);  // never valid
      a_base[i] = 0;
    }
    const int n = n0 + row;
    b_off[i] = n < N ? (long)n * pr.ldb : -1;
  }

  auto stage = [&](int buf, int kt) {
    char* As = smem + buf * STAGE_BYTES;
    char* Bs = As + 16384;
    int tap2[2], cc2[2], kk2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = pch ^ ((h << 2) + (l >> 4));
      const int kk = kt * BK + c * CH;
      kk2[h] = kk;
      if (taps == 1) {
        tap2[h] = 0;
        cc2[h] = kk;
      } else {
        tap2[h] = kk / Cin;
        cc2[h] = kk - tap2[h] * Cin;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int h = i & 1;
      const T* src = zero;
      if (kk2[h] < K) {
        int st;
        bool ok;
        if (mode == 0) {
          st = a_t[i] * stride + tap2[h] - pad;
          ok = st >= 0 && st < Lsrc;
        } else {
          const int num = a_t[i] + pad - tap2[h];
          if (stride == 1) {
            st = num;
            ok = num >= 0 && num < Lsrc;
          } else if (stride == 2) {
            st = num >> 1;
            ok = num >= 0 && (num & 1) == 0 && st < Lsrc;
          } else {
            st = num / stride;
            ok = num >= 0 && st * stride == num && st < Lsrc;
          }
        }
        if (ok) src = Ag + ((long)(a_base[i] + st) * pr.lda + cc2[h]);
      }
      glds16(src, As + (w * 4 + i) * 1024);
      const T* bsrc = zero;
      if (kk2[h] < K && b_off[i] >= 0) bsrc = Bg + (b_off[i] + kk2[h]);
      glds16(bsrc, Bs + (w * 4 + i) * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wr = w >> 1, wc = w & 1;
  const int swz = (l >> 1) & 7;
  const int nkt = (K + BK - 1) / BK;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
    const char* As = smem + cur * STAGE_BYTES;
    const char* Bs = As + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int pc = ((ks * 4 + (l >> 4)) ^ swz) * 16;
      typename Mma<T>::frag a[4], b[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        a[mi] = *(const typename Mma<T>::frag*)(As + (wr * 64 + mi * 16 + (l & 15)) * 128 + pc);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        b[ni] = *(const typename Mma<T>::frag*)(Bs + (wc * 64 + ni * 16 + (l & 15)) * 128 + pc);
      Mma<T>::run(a, b, acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue.  acc[mi][ni][r]: m = wr*64+mi*16+(l>>4)*4+r, n = wc*64+ni*16+(l&15)
  if (pr.stats) {
    // per-tile column sums of the raw fp32 accumulators (rows >= M contribute exact zeros)
    float* sh = (float*)smem;  // [2 wr][2 kind][128 n]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[mi][ni][r];
          s += v;
          q += v * v;
        }
      s += __shfl_xor(s, 16, 64);
      q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (l < 16) {
        sh[(wr * 2 + 0) * 128 + wc * 64 + ni * 16 + l] = s;
        sh[(wr * 2 + 1) * 128 + wc * 64 + ni * 16 + l] = q;
      }
    }
    __syncthreads();
    {
      const int kind = tid >> 7, n = tid & 127;
      if (n0 + n < N) pr.stats[((long)tm * 2 + kind) * N + n0 + n] = sh[(0 * 2 + kind) * 128 + n] + sh[(1 * 2 + kind) * 128 + n];
    }
  }

  T* __restrict__ Cg = (T*)pr.C;
  T* __restrict__ C2g = (T*)pr.C2;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wr * 64 + mi * 16 + (l >> 4) * 4 + r;
      if (m >= M) continue;
      const float* grow = pr.gate ? pr.gate + (long)(m / pr.Lout) * pr.ldg : nullptr;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wc * 64 + ni * 16 + (l & 15);
        if (n >= N) continue;
        float v = acc[mi][ni][r];
        if (pr.bias) v += pr.bias[n];
        const long off = (long)m * pr.ldc + n;
        if (C2g) DT<T>::st(C2g + ((long)m * pr.ldc2 + n), v);
        if (grow) v *= grow[n];
        if (pr.accumulate) v += DT<T>::ld(Cg + off);
        DT<T>::st(Cg + off, v);
      }
    }
  }
}

static int launch_nt(const DrnGemmDesc* d, int ngroups, int dtype, hipStream_t stream) {
  DRN_CHECK_ARG(ngroups >= 1 && ngroups <= DRN_MAX_GROUPS, "drn_gemm_nt: ngroups=%d out of range", ngroups);
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "drn_gemm_nt: bad dtype %d", dtype);
  const int ch = dtype == DRN_BF16 ? 8 : 4;
  GemmParams P;
  memset(&P, 0, sizeof(P));
  P.ngroups = ngroups;
  int total = 0;
  for (int g = 0; g < ngroups; ++g) {
    const DrnGemmDesc& s = d[g];
    GemmProb& p = P.p[g];
    DRN_CHECK_ARG(s.A && s.B && s.C, "drn_gemm_nt: null operand in group %d", g);
    DRN_CHECK_ARG(s.M > 0 && s.N > 0 && s.Cin > 0 && s.taps >= 1 && s.stride >= 1, "drn_gemm_nt: bad dims in group %d", g);
    DRN_CHECK_ARG(s.Cin % ch == 0 && s.lda % ch == 0 && s.ldb % ch == 0,
                  "drn_gemm_nt: Cin/lda/ldb must be multiples of %d elements (16 bytes); got Cin=%d lda=%d ldb=%d", ch,
                  s.Cin, s.lda, s.ldb);
    DRN_CHECK_ARG(((uintptr_t)s.A & 15) == 0 && ((uintptr_t)s.B & 15) == 0, "drn_gemm_nt: A/B must be 16-byte aligned");
    DRN_CHECK_ARG(s.Lout > 0 && s.Lsrc > 0 && s.M % s.Lout == 0, "drn_gemm_nt: M=%d not a multiple of Lout=%d", s.M, s.Lout);
    p.A = s.A; p.B = s.B; p.C = s.C; p.C2 = s.C2; p.bias = s.bias; p.gate = s.gate; p.stats = s.stats;
    p.M = s.M; p.N = s.N; p.K = s.taps * s.Cin; p.Cin = s.Cin; p.taps = s.taps; p.stride = s.stride; p.pad = s.pad;
    p.mode = s.mode; p.Lout = s.Lout; p.Lsrc = s.Lsrc; p.lda = s.lda; p.ldb = s.ldb; p.ldc = s.ldc; p.ldg = s.ldg; p.ldc2 = s.ldc2;
    p.accumulate = s.accumulate;
    p.tiles_n = cdiv(s.N, TILE);
    p.tile_start = total;
    total += cdiv(s.M, TILE) * p.tiles_n;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)conv_gemm_nt_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    hipFuncSetAttribute((const void*)conv_gemm_nt_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    attr_set = true;
  }
  if (dtype == DRN_BF16)
    conv_gemm_nt_kernel<bf16_t><<<total, NT_THREADS, 2 * STAGE_BYTES, stream>>>(P);
  else
    conv_gemm_nt_kernel<float><<<total, NT_THREADS, 2 * STAGE_BYTES, stream>>>(P);
  return drn_launch_status("drn_gemm_nt");
}

extern "C" int drn_gemm_nt(const DrnGemmDesc* descs, int ngroups, int dtype, void* stream) {
  drn_clear_status();
  return launch_nt(descs, ngroups, dtype, (hipStream_t)stream);
}
