// Weight-gradient implicit GEMM (TN) on CDNA4 MFMA.
//
//   dW[n][tap][c] = sum_m dY[m][n] * X[src(m,tap)][c]          (fp32 result)
//
// is the backward-weights of nn.Conv1d (model/basic_blocks.py:9-18, model/fcos.py:33-69) and,
// with taps==1, of nn.Linear (model/main_model.py:33).  Both operands are channels-last, so the
// reduction index m is the strided one: tiles are staged in their natural [rows][channels] order
// with global_load_lds and the MFMA fragments are gathered with the gfx950 hardware transpose
// read (ds_read_b64_tr_b16) for bf16, or plain ds_read_b32 (one k per lane) for exact-f32.
//
// Tile: 128 (n) x 128 (c, inside one tap) per 256-thread workgroup, 4 waves 2x2, 64x64 per wave.
// LDS image per operand and stage (16 KB): [8 column blocks of 16][R rows][16 cols], R = 64 (bf16)
// or 32 (f32) so one wave-instruction of global_load_lds fills 1 KB of it linearly.
// The row range (all groups concatenated) is split over gridDim.z; partial tiles go to an fp32
// workspace and are summed in a fixed order by wgrad_reduce_kernel (deterministic, no atomics).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/drn_hip.h"



struct WgradGroup {
  const void* dY;
  const void* X;
  int M, Lout, Lsrc, ldy, ldx;
  int blk_start;  // first row-block (of R rows) of this group in the concatenated space
};
struct WgradParams {
  int ngroups;
  WgradGroup g[DRN_MAX_GROUPS];
  int total_blks, blks_per_split;
  int N, Cin, taps, stride, pad;
  int ctiles;     // ceil(Cin/128)
  float* out;     // workspace [nsplit][N][taps*Cin] or final dW when nsplit==1
  int direct;     // 1: out is the final dW (nsplit==1)
  int w_layout;   // direct only: 0 = [N][taps][Cin], 1 = [N][Cin][taps]
  int accumulate; // direct only
  // multi != 0: every group is an INDEPENDENT problem of the same N / Cin / taps with its own output (the FPN level convs):
  // group g owns gridDim.z slices [z_start[g], z_start[g+1]), its rows are split into bps[g]-block pieces
  int multi;
  int z_start[DRN_MAX_GROUPS + 1], bps[DRN_MAX_GROUPS], gdirect[DRN_MAX_GROUPS];
  int gcin[DRN_MAX_GROUPS];   // multi: the problem's own input-channel count (the per-tap kernel only; the FPN laterals differ in Cin)
  float* gout[DRN_MAX_GROUPS];
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// LDS-DMA load as an asm statement, NOT the builtin: the compiler tracks builtin LDS-DMA stores and puts an
// `s_waitcnt vmcnt(0)` in front of every later LDS read whose memory operand is in the LDS address space (the transposing
// reads below are) -- which serialises each block's loads with its MFMAs.  The kernels order LDS-DMA against LDS reads
// themselves (counted vmcnt + barrier), exactly as the ring protocol needs.
__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
}

template <typename T> struct TnMma;

template <> struct TnMma<bf16_t> {
  static constexpr int R = 64;
  // image: [cb][r][16 cols] bf16, 32 B per row piece.  One stage = 2 MFMA k-steps of 32 rows.
  static __device__ __forceinline__ bf16x8 frag(const char* img, int cb, int ks, int l) {
    const int g = l >> 4, i = l & 15;
    const char* base = img + cb * 2048 + (i & 3) * 8;
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + (ks * 32 + g * 4 + (i >> 2)) * 32));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(base + (ks * 32 + 16 + g * 4 + (i >> 2)) * 32));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
  }
  template <int MI, int NI>
  static __device__ __forceinline__ void compute(const char* Ys, const char* Xs, int wr, int wc, int l, f32x4 (&acc)[MI][NI]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = frag(Ys, wr * MI + mi, ks, l);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = frag(Xs, wc * NI + ni, ks, l);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
  }
};

template <> struct TnMma<float> {
  static constexpr int R = 32;
  // image: [cb][r][16 cols] f32, 64 B per row piece.  One stage = 8 MFMA k-steps of 4 rows.
  template <int MI, int NI>
  static __device__ __forceinline__ void compute(const char* Ys, const char* Xs, int wr, int wc, int l, f32x4 (&acc)[MI][NI]) {
    const int g = l >> 4, i = l & 15;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float a[MI], b[NI];
      const int off = (ks * 4 + g) * 64 + i * 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *(const float*)(Ys + (wr * MI + mi) * 2048 + off);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *(const float*)(Xs + (wc * NI + ni) * 2048 + off);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
  }
};

// Tile: WM x WN waves of MI x NI MFMA tiles -> TNn = WM*MI*16 output channels x TC = WN*NI*16 input channels.
//   <2,2,4,4> 128x128 (4 waves, 2 workgroups/CU)   <2,4,8,4> 256x256 (8 waves) for the 4096x4096 prop_fc gradient.
template <typename T, int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_wgrad_tn_kernel(const WgradParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = TnMma<T>::R;
  constexpr int NW = WM * WN, TNn = WM * MI * 16, TC = WN * NI * 16;
  constexpr int IMG_Y = TNn * 128, IMG_X = TC * 128, STAGE_B = IMG_Y + IMG_X;   // bytes (R rows x 2 B or 32 rows x 4 B = 128 B/col)
  constexpr int PMAX = TNn * 128 / (NW * 1024);   // one-KB staging pieces per wave and operand
  static_assert(TNn == TC && PMAX * NW * 1024 == TNn * 128 && (PMAX == 4 || PMAX == 2), "square tiles, 4 or 2 staging pieces per wave and operand");
  constexpr int CH = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int CPR = (int)sizeof(T);       // 16-byte chunks per 16-column row piece
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;   // wave index as a scalar: LDS-DMA bases (M0) stay in SGPRs
  const int tn = blockIdx.x;                // 128-wide block of output channels n
  int split = blockIdx.z;
  int blk_lo = split * P.blks_per_split;
  int blk_hi = min(blk_lo + P.blks_per_split, P.total_blks);
  float* out_ptr = P.out;
  int direct = P.direct;
  int Cin = P.Cin, ctiles = P.ctiles;
  if (P.multi) {
    int pg = 0;
#pragma unroll
    for (int i = 1; i < DRN_MAX_GROUPS; ++i)
      if (i < P.ngroups && (int)blockIdx.z >= P.z_start[i]) pg = i;
    split = blockIdx.z - P.z_start[pg];
    const int blk_end = pg + 1 < P.ngroups ? P.g[pg + 1].blk_start : P.total_blks;
    blk_lo = P.g[pg].blk_start + split * P.bps[pg];
    blk_hi = min(blk_lo + P.bps[pg], blk_end);
    out_ptr = P.gout[pg];
    direct = P.gdirect[pg];
    Cin = P.gcin[pg];
    ctiles = (Cin + TC - 1) / TC;
    if ((int)blockIdx.y >= P.taps * ctiles) return;       // the grid is sized for the widest problem
  }
  const int tap = blockIdx.y / ctiles;
  const int c0 = (blockIdx.y - tap * ctiles) * TC;
  const int n0 = tn * TNn;
  const T* zero = (const T*)g_zero_page;

  // lane-constant piece of the staging map: instr q = w*4+i covers chunks p = q*64 + l
  int s_r[PMAX];
  long y_off[PMAX], x_off[PMAX];          // column offsets inside a row (or -1 when the 16-column block is out of range)
#pragma unroll
  for (int i = 0; i < PMAX; ++i) {
    const int q = w * PMAX + i;
    const int within = (q & 1) * 64 + l;
    const int col = (q >> 1) * 16 + (within % CPR) * CH;
    s_r[i] = within / CPR;
    y_off[i] = n0 + col < P.N ? n0 + col : -1;
    x_off[i] = c0 + col < Cin ? c0 + col : -1;
  }

  // Row blocks are consumed strictly in order; (group, first row of the block) advance incrementally and the
  // (sequence, position) of each of the thread's 4 rows is carried along, so the K-loop has no integer division.
  int s_blk = blk_lo, s_g = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.ngroups && blk_lo >= P.g[i].blk_start) s_g = i;
  int s_seq[PMAX], s_t[PMAX];             // of row (block base + s_r[i]) in the current group
  // the current group's fields live in registers (see conv_wgrad3_tn_kernel: run-time indexing of P.g[] inside the block
  // loop costs scalar loads whose lgkmcnt(0) waits also drain the LDS reads in flight)
  const T* g_Y;
  const T* g_X;
  int g_M, g_Lout, g_Lsrc, g_ldy, g_ldx, g_start, g_next;
  auto locate = [&]() {
    const WgradGroup& G = P.g[s_g];
    g_Y = (const T*)G.dY;
    g_X = (const T*)G.X;
    g_M = G.M; g_Lout = G.Lout; g_Lsrc = G.Lsrc; g_ldy = G.ldy; g_ldx = G.ldx; g_start = G.blk_start;
    g_next = s_g + 1 < P.ngroups ? P.g[s_g + 1].blk_start : 0x7fffffff;
    const int mbase = (s_blk - G.blk_start) * R;
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
      const int m = mbase + s_r[i];
      s_seq[i] = m / G.Lout;
      s_t[i] = m - s_seq[i] * G.Lout;
    }
  };
  locate();

  // branch-free: masked lanes read the zero page; every thread issues exactly 8 global_load_lds per block
  auto stage = [&](int buf) {
    char* Ys = smem + buf * STAGE_B;
    char* Xs = Ys + IMG_Y;
    const bool live = s_blk < blk_hi;
    const int mbase = (s_blk - g_start) * R;
    const T* __restrict__ Yg = g_Y;
    const T* __restrict__ Xg = g_X;
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
      const int m = mbase + s_r[i];
      const bool min_ = live & (m < g_M);
      const bool oky = min_ & (y_off[i] >= 0);
      const T* ys = oky ? Yg + ((long)m * g_ldy + y_off[i]) : zero;
      const int st = s_t[i] * P.stride + tap - P.pad;
      const bool okx = min_ & (x_off[i] >= 0) & (st >= 0) & (st < g_Lsrc);
      const T* xs = okx ? Xg + ((long)(s_seq[i] * g_Lsrc + st) * g_ldx + x_off[i]) : zero;
      glds16(ys, Ys + (w * PMAX + i) * 1024);
      glds16(xs, Xs + (w * PMAX + i) * 1024);
    }
    // advance to the next row block
    ++s_blk;
    if (s_blk >= g_next) {
      ++s_g;
      locate();
    } else {
      const int Lout = g_Lout;
#pragma unroll
      for (int i = 0; i < PMAX; ++i) {
        s_t[i] += R;
        while (s_t[i] >= Lout) {
          s_t[i] -= Lout;
          ++s_seq[i];
        }
      }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wr = w / WN, wc = w % WN;

  // 2-deep ring, one barrier per row block: wait for block b, barrier (everyone also finished block b-1), issue b+1
  // into the slot b-1 used, compute b.
  stage(0);
  int cur = 0;
  for (int blk = blk_lo; blk < blk_hi; ++blk) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage(cur ^ 1);
    const char* Ys = smem + cur * STAGE_B;
    TnMma<T>::template compute<MI, NI>(Ys, Ys + IMG_Y, wr, wc, l, acc);
    cur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // epilogue: acc[mi][ni][r] -> n = n0 + wr*MI*16 + mi*16 + (l>>4)*4 + r ; c = c0 + wc*NI*16 + ni*16 + (l&15)
  const int KW = P.taps * Cin;
  const bool rowmajor = !direct || (P.w_layout == 0 && !P.accumulate);   // destination rows contiguous along c
  float* obase = direct ? out_ptr : out_ptr + (long)split * P.N * KW;
  if (rowmajor && (KW % 4 == 0) && (Cin % 4 == 0) && (((uintptr_t)obase & 15) == 0)) {
    // coalesced: each wave transposes its 64-column slab through a private LDS patch (32 rows at a time) and writes
    // 16-byte row segments instead of 64 scattered 4-byte stores per lane
    constexpr int WCOLS = NI * 16;            // columns owned by one wave
    constexpr int PITCH = WCOLS * 4 + 16;
    constexpr int LPR = WCOLS / 4, RPI = 64 / LPR;
    __syncthreads();
    char* wbuf = smem + w * (32 * PITCH);
#pragma unroll
    for (int ch = 0; ch < MI / 2; ++ch) {
      const int nrow0 = n0 + wr * (MI * 16) + ch * 32;
#pragma unroll
      for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            *((float*)(wbuf + (mi2 * 16 + (l >> 4) * 4 + r) * PITCH) + ni * 16 + (l & 15)) = acc[ch * 2 + mi2][ni][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // per-wave patch: in-order LDS, no workgroup barrier needed
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {   // LPR lanes per row, RPI rows per instruction
        const int rl = it * RPI + l / LPR, cv = l % LPR;
        const int n = nrow0 + rl, c = c0 + wc * (NI * 16) + cv * 4;
        if (n < P.N && c < Cin) {             // Cin % 4 == 0: a quad never crosses the row end
          const f32x4 v = *(const f32x4*)(wbuf + rl * PITCH + cv * 16);
          *(f32x4*)(obase + ((long)n * KW + tap * Cin + c)) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // per-wave patch: in-order LDS, no workgroup barrier needed
    }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wr * (MI * 16) + mi * 16 + (l >> 4) * 4 + r;
      if (n >= P.N) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int c = c0 + wc * (NI * 16) + ni * 16 + (l & 15);
        if (c >= Cin) continue;
        float v = acc[mi][ni][r];
        if (direct) {
          float* dst = P.w_layout == 0 ? out_ptr + ((long)n * KW + tap * Cin + c)
                                       : out_ptr + ((long)n * KW + (long)c * P.taps + tap);
          if (P.accumulate) v += *dst;
          *dst = v;
        } else {
          out_ptr[((long)split * P.N + n) * KW + tap * Cin + c] = v;
        }
      }
    }
}

// ---- fused 3-tap weight gradient (k = 3, stride 1, pad 1, bf16) ---------------------------------------------------------
// The per-tap kernel above re-reads dY for every tap and X three times (each tap is the same rows shifted by one).  Here one
// workgroup owns a 128 (n) x 128 (c) tile for ALL THREE taps: per 64-row block it stages dY once (16 KB) and X once plus one
// halo row on each side (16.5 KB) and feeds 3 x the MFMAs -- a third of the staged bytes per MAC.
// Sequence ends need no masks: the reduction runs over a PADDED row space with Lout+1 rows per sequence whose extra row is
// zero in both operands (staged from the zero page), so the +-1 row shifts of taps 0 / 2 meet a zero row at either end of
// a sequence and the padding row itself contributes nothing.  p = seq*(Lout+1) + t  <->  source row m = p - seq.
// 8 waves as 2 (n) x 4 (c): 64 x 32 per wave and tap = 4 x 2 MFMA tiles x 3 taps = 96 accumulator registers.
// LDS per stage: dY [2 halves][64 rows][128 B] | X [2 halves][64 rows][128 B] | X halo [2 rows][2 halves][128 B] (row -1, 64).
// A half is 64 columns = four 32-byte column-block pieces per row; piece cb of row r sits at position cb ^ ((r>>1)&3), so
// the 8 rows x 4 eight-byte pieces a transposing read touches per 32-lane group fall on 64 distinct banks.  Staging:
// one global_load_lds covers 8 rows x 128 contiguous bytes (whole cache lines; the swizzle permutes the SOURCE chunks of a
// row among its 8 lanes).
// What bounds it (ablation on the conv0 shape, us per 64-row block and workgroup): MFMAs + fragment reads alone 0.89,
// the block's 33 KB of global_load_lds alone 0.71, together 1.6 -- the LDS-DMA issue (~46 GB/s per CU) does not overlap
// the matrix pipe when all eight waves enter it together; ring depth, register double-buffering of the fragments and
// cheaper address arithmetic each changed nothing.  Spreading the loads between the MFMA groups is worth 5 %.
__device__ __forceinline__ bf16x8 tr_frag(const char* lo_addr, const char* hi_addr) {
  typedef __attribute__((address_space(3))) s16x4* lp;
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)lo_addr);
  u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)hi_addr);
  return u.v;
}

template <int NSTG>
__global__ __launch_bounds__(512, 2) void conv_wgrad3_tn_kernel(const WgradParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int R = 64, MI = 4, NI = 2;
  constexpr int IMG = 16384, HALO_B = 512, STAGE_B = 2 * IMG + HALO_B;
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;   // wave index as a scalar: LDS-DMA bases (M0) stay in SGPRs
  const int n0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
  int split = blockIdx.z;
  int blk_lo = split * P.blks_per_split;
  int blk_hi = min(blk_lo + P.blks_per_split, P.total_blks);
  float* out_ptr = P.out;
  int direct = P.direct;
  if (P.multi) {
    int pg = 0;
#pragma unroll
    for (int i = 1; i < DRN_MAX_GROUPS; ++i)
      if (i < P.ngroups && (int)blockIdx.z >= P.z_start[i]) pg = i;
    split = blockIdx.z - P.z_start[pg];
    const int blk_end = pg + 1 < P.ngroups ? P.g[pg + 1].blk_start : P.total_blks;
    blk_lo = P.g[pg].blk_start + split * P.bps[pg];
    blk_hi = min(blk_lo + P.bps[pg], blk_end);
    out_ptr = P.gout[pg];
    direct = P.gdirect[pg];
  }
  const T* zero = (const T*)g_zero_page;

  // staging map.  Main pieces: wave w owns rows w*8 + (l>>3) of both images, instruction j = column half j; lane chunk
  // position pc = l&7 inside the 128-byte row holds source column block (pc>>1) ^ ((row>>1)&3), 16-byte half pc&1.
  // Halo: lanes 0..31 of wave 0: halo row l>>4 (row -1 / row 64), half (l>>3)&1, chunk l&7, unswizzled.
  const int srow = w * 8 + (l >> 3);
  const int scol = (((l & 7) >> 1) ^ ((srow >> 1) & 3)) * 16 + (l & 1) * 8;     // inside a 64-column half
  long y_off[2], x_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    y_off[j] = n0 + j * 64 + scol < P.N ? n0 + j * 64 + scol : -1;
    x_off[j] = c0 + j * 64 + scol < P.Cin ? c0 + j * 64 + scol : -1;
  }
  const bool is_halo = (w == 0) & (l < 32);
  const int hcol = ((l >> 3) & 1) * 64 + (l & 7) * 8;
  const long xh_off = c0 + hcol < P.Cin ? c0 + hcol : -1;
  const int rowoff[2] = {srow, (l >> 4) ? R : -1};          // [1]: the halo row of this lane
  int s_blk = blk_lo, s_g = 0;
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i)
    if (i < P.ngroups && blk_lo >= P.g[i].blk_start) s_g = i;
  int s_seq[2], s_t[2];             // padded coordinates of the thread's rows in the current group (floor division: p = -1 -> (-1, Lout))
  // the current group's fields live in registers: indexing P.g[] with a run-time group number inside the block loop costs
  // scalar loads whose lgkmcnt(0) waits also drain the LDS fragment reads in flight
  const T* g_Y;
  const T* g_X;
  int g_L, g_nseq, g_ldy, g_ldx, g_next;     // g_next: first row block of the next group (or past the end)
  auto locate = [&]() {
    const WgradGroup& G = P.g[s_g];
    g_Y = (const T*)G.dY;
    g_X = (const T*)G.X;
    g_L = G.Lout;
    g_nseq = G.M / G.Lout;
    g_ldy = G.ldy;
    g_ldx = G.ldx;
    g_next = s_g + 1 < P.ngroups ? P.g[s_g + 1].blk_start : 0x7fffffff;
    const int Lp = G.Lout + 1;
    const int pbase = (s_blk - G.blk_start) * R;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = pbase + rowoff[j] + Lp;       // >= 0
      const int q = p / Lp;
      s_seq[j] = q - 1;
      s_t[j] = p - q * Lp;
    }
  };
  locate();

  // Source row m = seq*Lout + t and element offsets m*ld fit 32 bits (checked on the host), so a row address is one 24-bit
  // multiply-add on top of a scalar base; rows outside the sequence / group read the zero page (a select, no branch).
  auto row_ptr = [&](const T* base, int m, int ld, long coff, bool ok) -> const T* {
    const unsigned off = __umul24((unsigned)m, (unsigned)ld) + (unsigned)coff;
    const T* p = base + off;
    return ok ? p : zero;
  };
  // part 0..3: one global_load_lds each (the caller spreads them between the MFMA groups); part 3 also carries the halo
  // rows; part 4 advances to the next block.
  auto stage_part = [&](int buf, int part) {
    char* Ys = smem + buf * STAGE_B;
    char* Xs = Ys + IMG;
    char* Hs = Xs + IMG;
    const bool live = s_blk < blk_hi;
    const int L = g_L;
    const bool ok = live & ((unsigned)s_seq[0] < (unsigned)g_nseq) & (s_t[0] < L);
    const int m = s_seq[0] * L + s_t[0];
    if (part == 0) glds16(row_ptr(g_Y, m, g_ldy, y_off[0], ok & (y_off[0] >= 0)), Ys + w * 1024);
    if (part == 1) glds16(row_ptr(g_X, m, g_ldx, x_off[0], ok & (x_off[0] >= 0)), Xs + w * 1024);
    if (part == 2) glds16(row_ptr(g_Y, m, g_ldy, y_off[1], ok & (y_off[1] >= 0)), Ys + 8192 + w * 1024);
    if (part == 3) {
      glds16(row_ptr(g_X, m, g_ldx, x_off[1], ok & (x_off[1] >= 0)), Xs + 8192 + w * 1024);
      if (is_halo) {
        const bool okh = live & ((unsigned)s_seq[1] < (unsigned)g_nseq) & (s_t[1] < L) & (xh_off >= 0);
        glds16(row_ptr(g_X, s_seq[1] * L + s_t[1], g_ldx, xh_off, okh), Hs);
      }
    }
    if (part != 4) return;
    ++s_blk;
    if (s_blk >= g_next) {
      ++s_g;
      locate();
    } else {
      const int Lp = L + 1;
      if (Lp > R) {                 // at most one sequence end per step
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          s_t[j] += R;
          const bool wrap = s_t[j] >= Lp;
          s_t[j] -= wrap ? Lp : 0;
          s_seq[j] += wrap ? 1 : 0;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          s_t[j] += R;
          while (s_t[j] >= Lp) {
            s_t[j] -= Lp;
            ++s_seq[j];
          }
        }
      }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int part = 0; part < 5; ++part) stage_part(buf, part);
  };

  f32x4 acc[3][MI][NI];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[t][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wr = w >> 2, wc = w & 3;

  // fragment addresses (bytes from the stage base).  Lane (g = l>>4, i = l&15) of a transposing read supplies the 8-byte
  // piece (i&3) of row ks*32 + [16 +] fr, fr = g*4 + (i>>2), of one 16-column block; a tap shifts the row by -1 / 0 / +1,
  // and the two rows that leave the block (row -1: ks 0 "lo" part, row 64: ks 1 "hi" part) come from the halo image.
  // ks / hi add multiples of 16 rows, which leave the swizzle term ((row>>1)&3) unchanged -> immediates.
  const int fr = (l >> 4) * 4 + ((l & 15) >> 2);            // 0..15
  const int fb = (l & 3) * 8;
  int ya[MI], xa[3][NI], x_first[NI], x_last[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) ya[mi] = wr * 8192 + fr * 128 + ((mi ^ ((fr >> 1) & 3)) * 32) + fb;   // n-blocks wr*4 + mi
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int half = wc >> 1, cb = (wc & 1) * 2 + ni;       // c-block wc*2 + ni
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int rr = fr + t - 1;
      xa[t][ni] = IMG + half * 8192 + rr * 128 + ((cb ^ ((rr >> 1) & 3)) * 32) + fb;
    }
    x_first[ni] = fr == 0 ? 2 * IMG + half * 128 + cb * 32 + fb : xa[0][ni];                       // tap 0 / ks 0 / lo
    x_last[ni] = fr == 15 ? 2 * IMG + 256 + half * 128 + cb * 32 + fb : xa[2][ni] + 4096 + 2048;   // tap 2 / ks 1 / hi
  }

  struct Frags {
    bf16x8 a[MI], b[3][NI];
  };
  auto load_frags = [&](Frags& F, const char* S, auto ks_) {
    constexpr int ks = decltype(ks_)::value;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const char* q = S + ya[mi] + ks * 4096;
      F.a[mi] = tr_frag(q, q + 2048);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const char* q0 = S + xa[0][ni] + ks * 4096;
      const char* q1 = S + xa[1][ni] + ks * 4096;
      const char* q2 = S + xa[2][ni] + ks * 4096;
      F.b[1][ni] = tr_frag(q1, q1 + 2048);
      F.b[0][ni] = tr_frag(ks == 0 ? S + x_first[ni] : q0, q0 + 2048);
      F.b[2][ni] = tr_frag(q2, ks == 1 ? S + x_last[ni] : q2 + 2048);
    }
  };
  auto mfma_frags = [&](const Frags& F, int mi0) {      // rows [mi0, mi0 + 2) of the 4 x 2 x 3 grid: 12 MFMAs
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[t][mi0 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.a[mi0 + mi], F.b[t][ni], acc[t][mi0 + mi][ni], 0, 0, 0);
  };
  typedef std::integral_constant<int, 0> K0;
  typedef std::integral_constant<int, 1> K1;

  // NSTG-deep ring + register double buffering of the fragments.  Iteration b: blocks b and b+1 have landed (NSTG-3 younger
  // ones stay in flight: 4 loads per block and wave, wave 0 one more for the halo rows), one barrier, refill the slot block
  // b-1 used; the fragments of (b, k-step 1) are read while the MFMAs of (b, k-step 0) run, and those of (b+1, k-step 0)
  // while the MFMAs of (b, k-step 1) run.
  auto wait_landed = [&](int younger) {       // `younger` whole blocks may stay in flight
    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (younger == 1) { if (w == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else { if (w == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
  };
#pragma unroll
  for (int st = 0; st < NSTG - 1; ++st) stage(st);
  wait_landed(NSTG - 2);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  Frags F0, F1;
  load_frags(F0, smem, K0());
  int cur = 0;
  for (int blk = blk_lo; blk < blk_hi; ++blk) {
    wait_landed(NSTG - 3);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int nxt = cur + NSTG - 1;
    if (nxt >= NSTG) nxt -= NSTG;
    const int cn = cur + 1 == NSTG ? 0 : cur + 1;
    load_frags(F1, smem + cur * STAGE_B, K1());
    stage_part(nxt, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_frags(F0, 0);
    __builtin_amdgcn_sched_barrier(0);
    stage_part(nxt, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_frags(F0, 2);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(F0, smem + cn * STAGE_B, K0());
    stage_part(nxt, 2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_frags(F1, 0);
    __builtin_amdgcn_sched_barrier(0);
    stage_part(nxt, 3);
    __builtin_amdgcn_sched_barrier(0);
    mfma_frags(F1, 2);
    __builtin_amdgcn_sched_barrier(0);
    stage_part(nxt, 4);
    cur = cn;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // epilogue: acc[t][mi][ni][r] -> n = n0 + wr*64 + mi*16 + (l>>4)*4 + r ; c = c0 + wc*32 + ni*16 + (l&15)
  const int KW = 3 * P.Cin;
  const bool rowmajor = !direct || (P.w_layout == 0 && !P.accumulate);
  float* obase = direct ? out_ptr : out_ptr + (long)split * P.N * KW;
  if (rowmajor && (P.Cin % 4 == 0) && (((uintptr_t)obase & 15) == 0)) {
    // each wave transposes 32 rows x 32 columns of one tap at a time through a private LDS patch and writes 16-byte segments
    constexpr int PITCH = 32 * 4 + 16;
    __syncthreads();
    char* wbuf = smem + w * (32 * PITCH);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int ch = 0; ch < MI / 2; ++ch) {
        const int nrow0 = n0 + wr * 64 + ch * 32;
#pragma unroll
        for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              *((float*)(wbuf + (mi2 * 16 + (l >> 4) * 4 + r) * PITCH) + ni * 16 + (l & 15)) = acc[t][ch * 2 + mi2][ni][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // per-wave patch: in-order LDS, no workgroup barrier needed
#pragma unroll
        for (int it = 0; it < 4; ++it) {          // 8 lanes per 32-float row, 8 rows per instruction
          const int rl = it * 8 + (l >> 3), cv = l & 7;
          const int n = nrow0 + rl, c = c0 + wc * 32 + cv * 4;
          if (n < P.N && c < P.Cin) {
            const f32x4 v = *(const f32x4*)(wbuf + rl * PITCH + cv * 16);
            *(f32x4*)(obase + ((long)n * KW + t * P.Cin + c)) = v;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    return;
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wr * 64 + mi * 16 + (l >> 4) * 4 + r;
        if (n >= P.N) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int c = c0 + wc * 32 + ni * 16 + (l & 15);
          if (c >= P.Cin) continue;
          float v = acc[t][mi][ni][r];
          if (direct) {
            float* dst = P.w_layout == 0 ? out_ptr + ((long)n * KW + t * P.Cin + c) : out_ptr + ((long)n * KW + (long)c * 3 + t);
            if (P.accumulate) v += *dst;
            *dst = v;
          } else {
            out_ptr[((long)split * P.N + n) * KW + t * P.Cin + c] = v;
          }
        }
      }
}

struct WgradReduceMulti {
  const float* ws[DRN_MAX_GROUPS];
  float* out[DRN_MAX_GROUPS];
  int nsplit[DRN_MAX_GROUPS];
  int cin[DRN_MAX_GROUPS];
};
// out[n][...] = (accumulate ? out : 0) + sum_z ws[z][n][tap*Cin+c], in the requested parameter layout; blockIdx.y = problem
__global__ void wgrad_reduce_multi_kernel(const WgradReduceMulti R, int N, int taps, int w_layout, int accumulate) {
  const int Cin = R.cin[blockIdx.y];
  const long KW = (long)taps * Cin;
  const long total = (long)N * KW;
  const float* __restrict__ ws = R.ws[blockIdx.y];
  float* __restrict__ out = R.out[blockIdx.y];
  const int nsplit = R.nsplit[blockIdx.y];
  if (nsplit <= 1) return;                       // written directly by the GEMM
  if (w_layout == 1 && taps == 3) {
    const long pairs = (long)N * Cin;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < pairs; idx += (long)gridDim.x * blockDim.x) {
      const long n = idx / Cin;
      const int c = (int)(idx - n * Cin);
      const float* p = ws + n * KW + c;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 4
      for (int z = 0; z < nsplit; ++z) {
        s0 += p[(long)z * total];
        s1 += p[(long)z * total + Cin];
        s2 += p[(long)z * total + 2 * Cin];
      }
      float* o = out + n * KW + (long)c * 3;
      if (accumulate) {
        s0 += o[0];
        s1 += o[1];
        s2 += o[2];
      }
      o[0] = s0;
      o[1] = s1;
      o[2] = s2;
    }
    return;
  }
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += ws[(long)z * total + idx];
    long o = idx;
    if (w_layout == 1) {
      const long n = idx / KW;
      const int rem = (int)(idx - n * KW);
      const int tap = rem / Cin, c = rem - tap * Cin;
      o = n * KW + (long)c * taps + tap;
    }
    if (accumulate) s += out[o];
    out[o] = s;
  }
}

// out[n][...] = (accumulate ? out : 0) + sum_z ws[z][n][tap*Cin+c], in the requested parameter layout
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int nsplit, int N, int Cin,
                                    int taps, int w_layout, int accumulate) {
  const long KW = (long)taps * Cin;
  const long total = (long)N * KW;
  if (w_layout == 1 && taps == 3) {
    // nn.Conv1d layout (Cout, Cin, 3): one thread per (n, c) sums its three taps (each read coalesced across the wave) and
    // writes 12 contiguous bytes -- the generic loop below writes every element at a 12-byte stride
    const long pairs = (long)N * Cin;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < pairs; idx += (long)gridDim.x * blockDim.x) {
      const long n = idx / Cin;
      const int c = (int)(idx - n * Cin);
      const float* p = ws + n * KW + c;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 4
      for (int z = 0; z < nsplit; ++z) {
        s0 += p[(long)z * total];
        s1 += p[(long)z * total + Cin];
        s2 += p[(long)z * total + 2 * Cin];
      }
      float* o = out + n * KW + (long)c * 3;
      if (accumulate) {
        s0 += o[0];
        s1 += o[1];
        s2 += o[2];
      }
      o[0] = s0;
      o[1] = s1;
      o[2] = s2;
    }
    return;
  }
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += ws[(long)z * total + idx];
    long o = idx;
    if (w_layout == 1) {
      const long n = idx / KW;
      const int rem = (int)(idx - n * KW);
      const int tap = rem / Cin, c = rem - tap * Cin;
      o = n * KW + (long)c * taps + tap;
    }
    if (accumulate) s += out[o];
    out[o] = s;
  }
}

// ---- deferred reduce passes.  Inside a training step six weight-gradient launches each ended with their own reduce launch
// (wgrad_reduce*: 7-13 us apiece, most of it the launch floor and the ramp of a 2000-workgroup streaming pass over 10-50 MB).
// Given a DrnWgradPending list (caller-owned HOST memory, include/drn_hip.h) the GEMM launches only record what their reduce would have
// been; drn_wgrad_reduce_pending() then runs ALL of them as ONE launch -- same per-element summation order over the splits, so the
// same bits.  The caller keeps the workspaces alive until then and must flush before anything reads the gradients
// (drn_amd.dist.GradReducer.collect does).  No state lives in the library: two models (or two threads) hand over two lists.
typedef DrnWgradPendItem WgradPendItem;
#define WGRAD_PEND_MAX DRN_WGRAD_PEND_MAX
typedef DrnWgradPending WgradPendParams;      // (plain data: goes to the kernel by value)

// 1 = recorded; 0 = not deferrable (no list, list full, or an accumulating reduce: it reads `out`, which nobody may have pending);
// < 0 = error: `out` already has a reduce recorded -- running either first would be wrong, the caller has to flush in between
static int wgrad_pend_push(DrnWgradPending* L, const float* ws, float* out, int nsplit, int N, int Cin, int taps, int w_layout, int accumulate) {
  if (!L || L->n < 0 || L->n >= WGRAD_PEND_MAX) return 0;
  for (int i = 0; i < L->n; ++i)
    if (L->it[i].out == out) {
      drn_set_error("drn_gemm_wgrad: the output already has a deferred reduce pending in this DrnWgradPending (flush with drn_wgrad_reduce_pending first)");
      return DRN_ERR_ARG;
    }
  if (accumulate) return 0;
  L->it[L->n++] = WgradPendItem{ws, out, nsplit, N, Cin, taps, w_layout, accumulate};
  return 1;
}

__global__ __launch_bounds__(256) void wgrad_reduce_all_kernel(const WgradPendParams P) {
  __shared__ float sh_sq[17];
  float sq = 0.f;
  int i = 0;
  for (int k = 1; k < P.n; ++k)
    if ((int)blockIdx.x >= P.blk_start[k]) i = k;
  const WgradPendItem& I = P.it[i];
  const int nb = P.blk_start[i + 1] - P.blk_start[i], b = blockIdx.x - P.blk_start[i];
  const float* __restrict__ ws = I.ws;
  float* __restrict__ out = I.out;
  const int N = I.N, Cin = I.Cin, taps = I.taps, nsplit = I.nsplit, accumulate = I.accumulate, w_layout = I.w_layout;
  const long KW = (long)taps * Cin;
  const long total = (long)N * KW;
  // (16-byte accesses -- four channels per thread -- were measured and bought nothing: 61.5 vs 60.5 us at the training shapes)
  if (w_layout == 1 && taps == 3) {               // the arithmetic of wgrad_reduce_kernel, statement for statement
    const long pairs = (long)N * Cin;
    for (long idx = (long)b * 256 + threadIdx.x; idx < pairs; idx += (long)nb * 256) {
      const long n = idx / Cin;
      const int c = (int)(idx - n * Cin);
      const float* p = ws + n * KW + c;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 4
      for (int z = 0; z < nsplit; ++z) {
        s0 += __builtin_nontemporal_load(p + (long)z * total);          // (partials are read once)
        s1 += __builtin_nontemporal_load(p + (long)z * total + Cin);
        s2 += __builtin_nontemporal_load(p + (long)z * total + 2 * Cin);
      }
      float* o = out + n * KW + (long)c * 3;
      if (accumulate) {
        s0 += o[0];
        s1 += o[1];
        s2 += o[2];
      }
      o[0] = s0;
      o[1] = s1;
      o[2] = s2;
      sq = fmaf(s0, s0, sq); sq = fmaf(s1, s1, sq); sq = fmaf(s2, s2, sq);
    }
  } else {
    for (long idx = (long)b * 256 + threadIdx.x; idx < total; idx += (long)nb * 256) {
      float s = 0.f;
      for (int z = 0; z < nsplit; ++z) s += __builtin_nontemporal_load(ws + (long)z * total + idx);
      long o = idx;
      if (w_layout == 1) {
        const long n = idx / KW;
        const int rem = (int)(idx - n * KW);
        const int tap = rem / Cin, c = rem - tap * Cin;
        o = n * KW + (long)c * taps + tap;
      }
      if (accumulate) s += out[o];
      out[o] = s;
      sq = fmaf(s, s, sq);
    }
  }
  if (P.sumsq) {                       // (wave-uniform: a kernel argument)
    sq = block_sum(sq, sh_sq);
    if (threadIdx.x == 0) P.sumsq[blockIdx.x] = sq;
  }
}

static int wgrad_pend_plan(DrnWgradPending* L) {
  int blocks = 0;
  for (int i = 0; i < L->n; ++i) {
    const WgradPendItem& I = L->it[i];
    const long work = (I.w_layout == 1 && I.taps == 3) ? (long)I.N * I.Cin : (long)I.N * I.taps * I.Cin;
    int nb = (int)((work + 255) / 256);
    if (nb > 2048) nb = 2048;
    L->blk_start[i] = blocks;
    blocks += nb;
  }
  L->blk_start[L->n] = blocks;
  return blocks;
}

extern "C" int drn_wgrad_pending_blocks(DrnWgradPending* L) {
  drn_clear_status();
  DRN_CHECK_ARG(L && L->n >= 0 && L->n <= WGRAD_PEND_MAX, "drn_wgrad_pending_blocks: bad list");
  return wgrad_pend_plan(L);
}

// bytes the pending launch will move (partials read + gradients written [+ read when accumulating]): its roofline denominator
extern "C" int64_t drn_wgrad_pending_bytes(const DrnWgradPending* L) {
  int64_t b = 0;
  if (!L || L->n < 0 || L->n > WGRAD_PEND_MAX) return 0;
  for (int i = 0; i < L->n; ++i) {
    const WgradPendItem& I = L->it[i];
    b += (int64_t)I.N * I.taps * I.Cin * 4 * (I.nsplit + 1 + (I.accumulate ? 1 : 0));
  }
  return b;
}

extern "C" int drn_wgrad_reduce_pending(DrnWgradPending* L, float* sumsq, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(L && L->n >= 0 && L->n <= WGRAD_PEND_MAX, "drn_wgrad_reduce_pending: bad list");
  if (L->n == 0) return DRN_OK;
  const int blocks = wgrad_pend_plan(L);
  L->sumsq = sumsq;
  wgrad_reduce_all_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(*L);
  L->n = 0;
  L->sumsq = nullptr;
  return drn_launch_status("drn_wgrad_reduce_pending");
}

// 256x256 tiles only when they alone fill half the chip (the 4096x4096 prop_fc gradient: 256 tiles)
static int wgrad_tile(int N, int Cin, int taps) {
  if (const char* e = drn_exp_env("DRN_TN_TILE")) return atoi(e) == 256 ? 256 : 128;
  return (long)cdiv(N, 256) * taps * cdiv(Cin, 256) >= 128 ? 256 : 128;
}

static int wgrad_nsplit(int total_blks, int N, int Cin, int taps) {
  const int tile = wgrad_tile(N, Cin, taps);
  const int tiles = cdiv(N, tile) * taps * cdiv(Cin, tile);
  const int tgt = drn_tuning(DRN_TUNE_EXP0 + 2) > 0 ? drn_tuning(DRN_TUNE_EXP0 + 2) : 768;      // (exp2: experiment override)
  int ns = cdiv(tile == 256 ? 256 : tgt, tiles);   // ~1 (8-wave) or ~3 (4-wave) workgroups per CU
  // every split must own enough row blocks to amortise its fp32 partial tile (write + re-read in the reduce pass)
  const int cap = total_blks / 12;
  if (ns > cap) ns = cap;
  if (ns < 1) ns = 1;
  if (ns > 64) ns = 64;
  return ns;
}

static int total_blocks_upper(int M_total, int ngroups_max) {
  // row blocks use R = 32 in the worst case (f32); each group adds at most one partial block
  return cdiv(M_total, 32) + ngroups_max;
}

// fused 3-tap kernel (conv_wgrad3_tn_kernel): one 8-wave workgroup per CU; row splits fill the chip once.
static int wgrad3_nsplit(int m_total, int N, int Cin) {
  int target = drn_tuning(DRN_TUNE_EXP0 + 1) > 0 ? drn_tuning(DRN_TUNE_EXP0 + 1) : 256;      // (exp1: experiment override)
  if (const char* e = drn_exp_env("DRN_TN3_TARGET")) target = atoi(e);
  const int tiles = cdiv(N, 128) * cdiv(Cin, 128);
  int ns = target / tiles;
  const int cap = cdiv(m_total, 64) / 8;      // every split owns >= 8 row blocks
  if (ns > cap) ns = cap;
  if (ns < 1) ns = 1;
  if (ns > 64) ns = 64;
  return ns;
}
static bool wgrad3_enabled() {
  return drn_tuning(DRN_TUNE_TN_FUSED) != 0;
}
// stride-1 geometry, enough rows to amortise the wider tile, 32-bit element offsets (rows are addressed with a 24-bit multiply)
static bool wgrad3_group_ok(const DrnWgradDesc& s) {
  return s.Lsrc == s.Lout && s.M < (1 << 24) && s.ldy < (1 << 24) && s.ldx < (1 << 24) &&
         (long)s.M * s.ldy < (1L << 31) && (long)s.M * s.ldx < (1L << 31);
}
static int wgrad3_min_rows() {
  return drn_tuning(DRN_TUNE_TN3_MINROWS);        // 4096 unless a test forces the kernel onto small shapes (drn_tune)
}
static void wgrad3_attr() {
  static bool set = false;
  if (!set) {
#ifdef DRN_EXPERIMENTS
    (void)hipFuncSetAttribute((const void*)conv_wgrad3_tn_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
#endif
    (void)hipFuncSetAttribute((const void*)conv_wgrad3_tn_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    set = true;
  }
}
#define WGRAD3_STAGE (2 * 16384 + 512)
static int wgrad3_stages() {
  int st = 4;
  if (const char* e = drn_exp_env("DRN_TN3_STAGES")) st = atoi(e);
  return st < 3 ? 3 : (st > 4 ? 4 : st);
}
#ifdef DRN_EXPERIMENTS
#define WGRAD3_LAUNCH(GRID, P) do { const int st_ = wgrad3_stages(); \
    if (st_ == 3) conv_wgrad3_tn_kernel<3><<<GRID, 512, 3 * WGRAD3_STAGE, stream>>>(P); \
    else conv_wgrad3_tn_kernel<4><<<GRID, 512, 4 * WGRAD3_STAGE, stream>>>(P); } while (0)
#else
#define WGRAD3_LAUNCH(GRID, P) conv_wgrad3_tn_kernel<4><<<GRID, 512, 4 * WGRAD3_STAGE, stream>>>(P)      // 4-deep ring
#endif

extern "C" int64_t drn_wgrad_ws_elems(int M_total, int N, int Cin, int taps) {
  int ns = wgrad_nsplit(total_blocks_upper(M_total, DRN_MAX_GROUPS), N, Cin, taps);
  if (taps == 3) {
    const int ns3 = wgrad3_nsplit(M_total, N, Cin);
    if (ns3 > ns) ns = ns3;
  }
  return ns > 1 ? (int64_t)ns * N * taps * Cin : 0;
}

extern "C" int drn_gemm_wgrad(const DrnWgradDesc* d, int ngroups, float* dW, int N, int Cin, int taps, int stride, int pad,
                              int w_layout, int accumulate, float* ws, int dtype, DrnWgradPending* pend, void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(ngroups >= 1 && ngroups <= DRN_MAX_GROUPS, "drn_gemm_wgrad: ngroups=%d out of range", ngroups);
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "drn_gemm_wgrad: bad dtype %d", dtype);
  DRN_CHECK_ARG(dW && N > 0 && Cin > 0 && taps >= 1 && stride >= 1, "drn_gemm_wgrad: bad dims");
  if (pend)      // (checked BEFORE the GEMM launch: an output that already has a reduce recorded cannot take a second one)
    for (int i = 0; i < pend->n && i < WGRAD_PEND_MAX; ++i)
      DRN_CHECK_ARG(pend->it[i].out != dW, "drn_gemm_wgrad: the output already has a deferred reduce pending in this DrnWgradPending (flush first)");
  const int ch = dtype == DRN_BF16 ? 8 : 4;
  const int R = dtype == DRN_BF16 ? 64 : 32;
  WgradParams P;
  memset(&P, 0, sizeof(P));
  P.ngroups = ngroups;
  int blks = 0, m_total = 0;
  for (int g = 0; g < ngroups; ++g) {
    const DrnWgradDesc& s = d[g];
    DRN_CHECK_ARG(s.dY && s.X && s.M > 0 && s.Lout > 0 && s.Lsrc > 0 && s.M % s.Lout == 0, "drn_gemm_wgrad: bad group %d", g);
    DRN_CHECK_ARG(s.ldy % ch == 0 && s.ldx % ch == 0 && N % ch == 0 && Cin % ch == 0,
                  "drn_gemm_wgrad: N/Cin/ldy/ldx must be multiples of %d elements", ch);
    DRN_CHECK_ARG(((uintptr_t)s.dY & 15) == 0 && ((uintptr_t)s.X & 15) == 0, "drn_gemm_wgrad: operands must be 16-byte aligned");
    P.g[g].dY = s.dY; P.g[g].X = s.X; P.g[g].M = s.M; P.g[g].Lout = s.Lout; P.g[g].Lsrc = s.Lsrc;
    P.g[g].ldy = s.ldy; P.g[g].ldx = s.ldx; P.g[g].blk_start = blks;
    blks += cdiv(s.M, R);
    m_total += s.M;
  }
  bool fused = dtype == DRN_BF16 && taps == 3 && stride == 1 && pad == 1 && wgrad3_enabled() && m_total >= wgrad3_min_rows();
  for (int g = 0; g < ngroups; ++g) fused = fused && wgrad3_group_ok(d[g]);
  if (fused) {
    // row blocks of the PADDED row space (Lout + 1 rows per sequence), see conv_wgrad3_tn_kernel
    blks = 0;
    for (int g = 0; g < ngroups; ++g) {
      P.g[g].blk_start = blks;
      blks += cdiv((d[g].M / d[g].Lout) * (d[g].Lout + 1), 64);
    }
    int ns = wgrad3_nsplit(m_total, N, Cin);
    if (ns > blks) ns = blks;
    DRN_CHECK_ARG(ns == 1 || ws, "drn_gemm_wgrad: workspace required (drn_wgrad_ws_elems)");
    P.total_blks = blks;
    P.blks_per_split = cdiv(blks, ns);
    ns = cdiv(blks, P.blks_per_split);
    P.N = N; P.Cin = Cin; P.taps = taps; P.stride = stride; P.pad = pad;
    P.ctiles = cdiv(Cin, 128);
    P.direct = ns == 1;
    P.out = P.direct ? dW : ws;
    P.w_layout = w_layout;
    P.accumulate = accumulate;
    wgrad3_attr();
    WGRAD3_LAUNCH(dim3(cdiv(N, 128), cdiv(Cin, 128), ns), P);
    int rc = drn_launch_status("drn_gemm_wgrad(fused taps)");
    if (rc) return rc;
    const int pushed = P.direct ? 0 : wgrad_pend_push(pend, ws, dW, ns, N, Cin, taps, w_layout, accumulate);
    if (pushed < 0) return pushed;
    if (!P.direct && !pushed) {
      const long total = (long)N * taps * Cin;
      int nb = (int)((total + 255) / 256);
      if (nb > 2048) nb = 2048;
      wgrad_reduce_kernel<<<nb, 256, 0, stream>>>(ws, dW, ns, N, Cin, taps, w_layout, accumulate);
      rc = drn_launch_status("drn_gemm_wgrad(reduce)");
    }
    return rc;
  }
  // the split count must not exceed what drn_wgrad_ws_elems() promised for this M_total
  int ns = wgrad_nsplit(total_blocks_upper(m_total, DRN_MAX_GROUPS), N, Cin, taps);
  if (ns > blks) ns = blks;
  DRN_CHECK_ARG(ns == 1 || ws, "drn_gemm_wgrad: workspace required (drn_wgrad_ws_elems)");
  P.total_blks = blks;
  P.blks_per_split = cdiv(blks, ns);
  ns = cdiv(blks, P.blks_per_split);
  P.N = N; P.Cin = Cin; P.taps = taps; P.stride = stride; P.pad = pad;
  const int tile = wgrad_tile(N, Cin, taps);
  P.ctiles = cdiv(Cin, tile);
  P.direct = ns == 1;
  P.out = P.direct ? dW : ws;
  P.w_layout = w_layout;
  P.accumulate = accumulate;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tn_kernel<float, 2, 2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tn_kernel<bf16_t, 2, 2, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tn_kernel<float, 2, 4, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tn_kernel<bf16_t, 2, 4, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tn_kernel<float, 2, 4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tn_kernel<bf16_t, 2, 4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    attr_set = true;
  }
  dim3 grid(cdiv(N, tile), taps * P.ctiles, ns);
  if (tile == 256) {
    if (dtype == DRN_BF16) conv_wgrad_tn_kernel<bf16_t, 2, 4, 8, 4><<<grid, 512, 2 * 65536, stream>>>(P);
    else conv_wgrad_tn_kernel<float, 2, 4, 8, 4><<<grid, 512, 2 * 65536, stream>>>(P);
  } else {
#ifdef DRN_EXPERIMENTS
    if (drn_exp_env("DRN_TN_WAVES") && atoi(drn_exp_env("DRN_TN_WAVES")) == 4) {
      if (dtype == DRN_BF16) conv_wgrad_tn_kernel<bf16_t, 2, 2, 4, 4><<<grid, 256, 2 * 32768, stream>>>(P);
      else conv_wgrad_tn_kernel<float, 2, 2, 4, 4><<<grid, 256, 2 * 32768, stream>>>(P);
    } else
#endif
    if (dtype == DRN_BF16) conv_wgrad_tn_kernel<bf16_t, 2, 4, 4, 2><<<grid, 512, 2 * 32768, stream>>>(P);      // 8 waves per 128x128 tile
    else conv_wgrad_tn_kernel<float, 2, 4, 4, 2><<<grid, 512, 2 * 32768, stream>>>(P);
  }
  int rc = drn_launch_status("drn_gemm_wgrad");
  if (rc) return rc;
  const int pushed = P.direct ? 0 : wgrad_pend_push(pend, ws, dW, ns, N, Cin, taps, w_layout, accumulate);
  if (pushed < 0) return pushed;
  if (!P.direct && !pushed) {
    const long total = (long)N * taps * Cin;
    int nb = (int)((total + 255) / 256);
    if (nb > 2048) nb = 2048;
    wgrad_reduce_kernel<<<nb, 256, 0, stream>>>(ws, dW, ns, N, Cin, taps, w_layout, accumulate);
    rc = drn_launch_status("drn_gemm_wgrad(reduce)");
  }
  return rc;
}

// n INDEPENDENT weight gradients of equal N / Cin / taps / stride (the three FPN level convs: different weights, different
// row counts) in ONE launch + one reduce launch; problem i: dWs[i] = dY_i^T x im2col(X_i).  ws >= n * drn_wgrad_ws_elems(max M).
extern "C" int drn_gemm_wgrad_multi(const DrnWgradDesc* d, int n, float* const* dWs, int N, int Cin, const int32_t* Cins, int taps,
                                    int stride, int pad, int w_layout, int accumulate, float* ws, int dtype, DrnWgradPending* pend,
                                    void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(d && dWs && n >= 1 && n <= DRN_MAX_GROUPS, "drn_gemm_wgrad_multi: n=%d out of range", n);
  if (pend)      // (checked BEFORE the GEMM launch: an output that already has a reduce recorded cannot take a second one)
    for (int g = 0; g < n; ++g)
      for (int i = 0; i < pend->n && i < WGRAD_PEND_MAX; ++i)
        DRN_CHECK_ARG(pend->it[i].out != dWs[g], "drn_gemm_wgrad_multi: output %d already has a deferred reduce pending in this DrnWgradPending", g);
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "drn_gemm_wgrad_multi: bad dtype %d", dtype);
  DRN_CHECK_ARG(N > 0 && Cin > 0 && taps >= 1 && stride >= 1, "drn_gemm_wgrad_multi: bad dims");
  const int ch = dtype == DRN_BF16 ? 8 : 4;
  const int R = dtype == DRN_BF16 ? 64 : 32;
  const int tile = 128;
  WgradParams P;
  WgradReduceMulti RM;
  memset(&P, 0, sizeof(P));
  memset(&RM, 0, sizeof(RM));
  P.ngroups = n;
  P.multi = 1;
  // Cins (host, or NULL = every problem has Cin input channels): the problems' own channel counts -- the FPN laterals share N,
  // taps = 1 and differ in Cin; Cin is then the LARGEST of them (workspace slices, grid)
  bool varied = false;
  for (int g = 0; g < n && Cins; ++g) {
    DRN_CHECK_ARG(Cins[g] > 0 && Cins[g] <= Cin && Cins[g] % ch == 0, "drn_gemm_wgrad_multi: bad Cins[%d]", g);
    varied |= Cins[g] != Cin;
  }
  int blks = 0, z = 0, mmax = 0;
  bool any_split = false;
  for (int g = 0; g < n; ++g) mmax = d[g].M > mmax ? d[g].M : mmax;
  const long ws_per = drn_wgrad_ws_elems(mmax, N, Cin, taps);
  bool fused = !varied && dtype == DRN_BF16 && taps == 3 && stride == 1 && pad == 1 && wgrad3_enabled() && mmax >= wgrad3_min_rows();
  for (int g = 0; g < n; ++g) fused = fused && wgrad3_group_ok(d[g]);
  for (int g = 0; g < n; ++g) {
    const DrnWgradDesc& s = d[g];
    DRN_CHECK_ARG(s.dY && s.X && dWs[g] && s.M > 0 && s.Lout > 0 && s.Lsrc > 0 && s.M % s.Lout == 0, "drn_gemm_wgrad_multi: bad problem %d", g);
    DRN_CHECK_ARG(s.ldy % ch == 0 && s.ldx % ch == 0 && N % ch == 0 && Cin % ch == 0,
                  "drn_gemm_wgrad_multi: N/Cin/ldy/ldx must be multiples of %d elements", ch);
    DRN_CHECK_ARG(((uintptr_t)s.dY & 15) == 0 && ((uintptr_t)s.X & 15) == 0, "drn_gemm_wgrad_multi: operands must be 16-byte aligned");
    P.g[g].dY = s.dY; P.g[g].X = s.X; P.g[g].M = s.M; P.g[g].Lout = s.Lout; P.g[g].Lsrc = s.Lsrc;
    P.g[g].ldy = s.ldy; P.g[g].ldx = s.ldx; P.g[g].blk_start = blks;
    const int gb = fused ? cdiv((s.M / s.Lout) * (s.Lout + 1), 64) : cdiv(s.M, R);
    blks += gb;
    const int cin_g = Cins ? Cins[g] : Cin;
    const long per = (long)N * taps * cin_g;
    P.gcin[g] = cin_g; RM.cin[g] = cin_g;
    int ns = wgrad_nsplit(total_blocks_upper(s.M, 1), N, cin_g, taps);
    if (fused) {
      // the problems share the chip: each gets its share of the one-workgroup-per-CU budget by row count
      long m_all = 0;
      for (int k = 0; k < n; ++k) m_all += d[k].M;
      ns = (int)((long)wgrad3_nsplit(mmax, N, Cin) * s.M / m_all);
      const int cap = gb / 4;
      if (ns > cap) ns = cap;
      if (ns < 1) ns = 1;
    }
    if (ns > gb) ns = gb;
    P.bps[g] = cdiv(gb, ns);
    ns = cdiv(gb, P.bps[g]);
    DRN_CHECK_ARG(ns == 1 || ws, "drn_gemm_wgrad_multi: workspace required");
    P.z_start[g] = z;
    z += ns;
    P.gdirect[g] = ns == 1;
    P.gout[g] = ns == 1 ? dWs[g] : ws + (long)g * ws_per;
    RM.ws[g] = ws + (long)g * ws_per; RM.out[g] = dWs[g]; RM.nsplit[g] = ns;
    any_split |= ns > 1;
    DRN_CHECK_ARG(ns == 1 || (long)ns * per <= ws_per, "drn_gemm_wgrad_multi: workspace slice too small");
  }
  P.z_start[n] = z;
  P.total_blks = blks;
  P.blks_per_split = blks;
  P.N = N; P.Cin = Cin; P.taps = taps; P.stride = stride; P.pad = pad;
  P.ctiles = cdiv(Cin, tile);
  P.w_layout = w_layout;
  P.accumulate = accumulate;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tn_kernel<float, 2, 4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tn_kernel<bf16_t, 2, 4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    attr_set = true;
  }
  dim3 grid(cdiv(N, tile), taps * P.ctiles, z);
  if (fused) {
    wgrad3_attr();
    WGRAD3_LAUNCH(dim3(cdiv(N, 128), cdiv(Cin, 128), z), P);
  } else if (dtype == DRN_BF16) conv_wgrad_tn_kernel<bf16_t, 2, 4, 4, 2><<<grid, 512, 2 * 32768, stream>>>(P);      // 8 waves per 128x128 tile
  else conv_wgrad_tn_kernel<float, 2, 4, 4, 2><<<grid, 512, 2 * 32768, stream>>>(P);
  int rc = drn_launch_status("drn_gemm_wgrad_multi");
  if (rc || !any_split) return rc;
  if (pend && !accumulate && pend->n >= 0 && pend->n + n <= WGRAD_PEND_MAX) {
    for (int g = 0; g < n; ++g)
      if (RM.nsplit[g] > 1) (void)wgrad_pend_push(pend, RM.ws[g], RM.out[g], RM.nsplit[g], N, RM.cin[g], taps, w_layout, accumulate);
    return rc;
  }
  int nb = (int)(((long)N * taps * Cin + 255) / 256);
  if (nb > 1024) nb = 1024;
  wgrad_reduce_multi_kernel<<<dim3(nb, n), 256, 0, stream>>>(RM, N, taps, w_layout, accumulate);
  return drn_launch_status("drn_gemm_wgrad_multi(reduce)");
}
