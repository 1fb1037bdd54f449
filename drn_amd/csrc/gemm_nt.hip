// Launchers of the grouped implicit-GEMM (NT) kernel (gemm_nt_kernel.h): drn_gemm_nt, drn_gemm_nt_splitk(_grouped).
#include "gemm_nt_kernel.h"

bool drn_nt_w4_eligible(const DrnGemmDesc* d, int ngroups, int dtype);            // gemm_nt_w4.hip
int drn_nt_w4_launch(const GemmParams& P, int total, hipStream_t stream);
bool drn_nt_w4c_eligible(const DrnGemmDesc* d, int ngroups, int dtype);
int drn_nt_w4c_launch(const GemmParams& P, int total, hipStream_t stream, int ksplit);
bool drn_nt_w4h_eligible(const DrnGemmDesc* d, int ngroups, int dtype, bool* conv_out);  // gemm_nt_w4h.hip
int drn_nt_w4h_launch(const GemmParams& P, int total, bool conv, hipStream_t stream, int ksplit);

// Which kernel a launch runs on, given the tile size launch_nt chose (drn_gemm_nt_plan reports it to callers that schedule
// around a launch -- functional.input_prep's weight pre-touch -- instead of re-deriving the rule on their side).
static int nt_kind(const DrnGemmDesc* d, int ngroups, int dtype, int tile, int ksplit, bool planes256) {
  if (tile == 256 && ksplit == 1 && drn_tuning(DRN_TUNE_NT_W4) > 0 && drn_nt_w4_eligible(d, ngroups, dtype)) return DRN_NT_KIND_W4;
  if (tile == 256 && (planes256 || (ksplit == 1 && drn_tuning(DRN_TUNE_NT_W4C) > 0)) && drn_nt_w4c_eligible(d, ngroups, dtype))
    return DRN_NT_KIND_W4C;
  return tile == 256 ? DRN_NT_KIND_TILE256 : DRN_NT_KIND_TILE128;
}

static int launch_nt(const DrnGemmDesc* d, int ngroups, int dtype, hipStream_t stream, int ksplit_arg = 1, float* ws = nullptr,
                     int* counters = nullptr, bool planes256 = false, bool plan_only = false) {
  // (the public `ksplit` argument: split count in the low 16 bits + DRN_KSPLIT_CONFIRM_* request bits, include/drn_hip.h)
  const int ksplit = ksplit_arg & 0xffff;
  const int xchg = ksplit_arg & (DRN_XCHG_CONFIRM | DRN_XCHG_NONE);
  DRN_CHECK_ARG(ngroups >= 1 && ngroups <= DRN_MAX_GROUPS, "drn_gemm_nt: ngroups=%d out of range", ngroups);
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "drn_gemm_nt: bad dtype %d", dtype);
  const int ch = dtype == DRN_BF16 ? 8 : 4;
  // Tile choice: 256x256 (8 waves) when every group is large enough to keep the chip busy with such tiles,
  // else 128x128.  DRN_NT_TILE=128|256 overrides (tests exercise both).
  long big_tiles = 0;
  bool fast = true;
  for (int g = 0; g < ngroups; ++g) {
    const DrnGemmDesc& s = d[g];
    DRN_CHECK_ARG(s.A && s.B && s.C, "drn_gemm_nt: null operand in group %d", g);
    DRN_CHECK_ARG(s.M > 0 && s.N > 0 && s.Cin > 0 && s.taps >= 1 && s.stride >= 1, "drn_gemm_nt: bad dims in group %d", g);
    DRN_CHECK_ARG(s.Cin % ch == 0 && s.lda % ch == 0 && s.ldb % ch == 0,
                  "drn_gemm_nt: Cin/lda/ldb must be multiples of %d elements (16 bytes); got Cin=%d lda=%d ldb=%d", ch,
                  s.Cin, s.lda, s.ldb);
    DRN_CHECK_ARG(((uintptr_t)s.A & 15) == 0 && ((uintptr_t)s.B & 15) == 0, "drn_gemm_nt: A/B must be 16-byte aligned");
    DRN_CHECK_ARG(s.Lout > 0 && s.Lsrc > 0 && s.M % s.Lout == 0, "drn_gemm_nt: M=%d not a multiple of Lout=%d", s.M, s.Lout);
    big_tiles += (long)cdiv(s.M, 256) * cdiv(s.N, 256);
    if (s.Cin % (8 * ch) != 0 || (s.mode == 1 && s.stride > 2)) fast = false;
  }
  const int big_min = drn_tuning(DRN_TUNE_EXP0) > 0 ? drn_tuning(DRN_TUNE_EXP0) : 200;     // (exp0: experiment override)
  int tile = big_tiles >= big_min ? 256 : 128;
  if (const char* e = drn_exp_env("DRN_NT_TILE")) tile = atoi(e) == 256 ? 256 : 128;
  if (drn_exp_env("DRN_NT_GENERIC")) fast = false;
  GemmParams P;
  memset(&P, 0, sizeof(P));
  P.ngroups = ngroups;
  // in-launch exchanges confirm their stores by read-back unless the CALL says otherwise (per launch: no process-wide switch)
  P.ksplit = ksplit | (ksplit > 1 && counters ? (xchg & DRN_XCHG_NONE ? 0 : xchg & DRN_XCHG_CONFIRM ? DRN_XCHG_CONFIRM : DRN_XCHG_READBACK) : 0);
  P.ws = ws;
  P.counters = counters;
  P.xcd_swizzle = drn_exp_env("DRN_NO_XCD_SWIZZLE") ? 0 : 3;
  if (drn_tuning(DRN_TUNE_EXP0 + 3) > 0) P.xcd_swizzle = drn_tuning(DRN_TUNE_EXP0 + 3) - 1;   // (exp3: experiment override, value - 1)
  if (const char* e = drn_exp_env("DRN_NT_ORDER")) P.xcd_swizzle = atoi(e);      // bit 0: XCD-contiguous runs, bit 1: 8-row grouped order
  if (ksplit > 1) tile = planes256 ? 256 : 128;
  // launches on 128x128 tiles whose problems make enough 256x128 tiles: the 4-wave loop on half-width tiles (gemm_nt_w4h.hip)
  bool w4h = false, w4h_conv = false;
  if (tile == 128 && !planes256 && drn_tuning(DRN_TUNE_NT_W4H) > 0 && drn_nt_w4h_eligible(d, ngroups, dtype, &w4h_conv)) {
    long th = 0;
    int min_ksteps = 1 << 30;
    for (int g = 0; g < ngroups; ++g) {
      th += (long)(d[g].M / 256) * (d[g].N / 128);
      const int ksteps = d[g].taps * d[g].Cin / 64, per = cdiv(ksteps, ksplit);
      min_ksteps = min(min_ksteps, ksteps - (cdiv(ksteps, per) - 1) * per);          // the last split's share
      if (cdiv(ksteps, per) != ksplit) min_ksteps = 0;                               // (a split would be empty)
    }
    // one launch: enough tiles; split (drn_gemm_nt_splitk*: the caller sized ksplit for this kernel -- every split keeps at least two
    // K-steps and the tiles have arrival counters)
    w4h = ksplit == 1 ? th >= drn_tuning(DRN_TUNE_NT_W4H)
                      : (ngroups == 1 && th * ksplit >= drn_tuning(DRN_TUNE_NT_W4H) && min_ksteps >= 2 && th <= DRN_QD_COUNTERS && ws && counters);
  }
  const int tile_m = w4h ? 256 : tile, tile_n = w4h ? 128 : tile;
  int total = 0;
  for (int g = 0; g < ngroups; ++g) {
    const DrnGemmDesc& s = d[g];
    GemmProb& p = P.p[g];
    p.A = s.A; p.B = s.B; p.C = s.C; p.C2 = s.C2; p.bias = s.bias; p.gate = s.gate; p.stats = s.stats;
    p.M = s.M; p.N = s.N; p.K = s.taps * s.Cin; p.Cin = s.Cin; p.taps = s.taps; p.stride = s.stride; p.pad = s.pad;
    p.mode = s.mode; p.Lout = s.Lout; p.Lsrc = s.Lsrc; p.lda = s.lda; p.ldb = s.ldb; p.ldc = s.ldc; p.ldg = s.ldg; p.ldc2 = s.ldc2;
    p.accumulate = s.accumulate;
    p.out_f32 = s.out_f32;
    p.sumsq = s.sumsq;
    p.gb_act = s.gb_act; p.gb_dct = s.gb_dct; p.gb_dgate = s.gb_dgate; p.gb_dsum = s.gb_dsum; p.gb_ld_act = s.gb_ld_act; p.gb_ldt = s.gb_ldt;
    p.tiles_n = cdiv(s.N, tile_n);
    p.tile_start = total;
    total += cdiv(s.M, tile_m) * p.tiles_n;
  }
  // Pipeline depth for the 128x128 tile: 2 stages leave room for two workgroups per CU (best when the grid
  // oversubscribes the chip); 4 stages (one workgroup per CU) otherwise.  DRN_NT_STAGES overrides for experiments.
  int stages = (long)total * ksplit > 256 ? 2 : 4;
  if (const char* e = drn_exp_env("DRN_NT_STAGES")) stages = atoi(e) == 2 ? 2 : 4;
  if (tile == 256) stages = 2;
  // 128x128 tiles: 8 waves per workgroup, 2-slot ring (measured 20-30 % faster than 4 waves at one workgroup per CU, 8 % at
  // two; DRN_NT_WAVES=4 brings the 4-wave variants back for experiments)
  bool waves8 = tile == 128;
  if (const char* e = drn_exp_env("DRN_NT_WAVES")) waves8 = tile == 128 && atoi(e) == 8;
  if (ksplit > 1) waves8 = true;       // the in-launch split-K exchange exists for the 8-wave 128x128 tile only
  if (waves8 && !drn_exp_env("DRN_NT_STAGES")) stages = 2;
  // launches that leave CUs idle anyway (<= 256 workgroups: most of the pyramid GEMMs at T = 256, everything at Charades-STA's
  // T = 32) run one workgroup per CU with a 4-slot ring -- three K-tiles of loads in flight against the cold operands instead of
  // one: step 1.51 -> 1.435 ms at T = 32, 2.448 -> 2.434 at T = 256 (512 as the bound: 2.465)
  int max_ksteps = 0;
  for (int g = 0; g < ngroups; ++g) max_ksteps = max(max_ksteps, cdiv(cdiv(d[g].taps * d[g].Cin, 8 * ch), ksplit));
  const bool deep8 = waves8 && stages == 2 &&
                     ((drn_tuning(DRN_TUNE_NT_DEEP) > 0 && (long)total * ksplit <= drn_tuning(DRN_TUNE_NT_DEEP)) ||
                      (drn_tuning(DRN_TUNE_NT_DEEP2) > 0 && (long)total * ksplit <= drn_tuning(DRN_TUNE_NT_DEEP2) &&
                       max_ksteps <= drn_tuning(DRN_TUNE_NT_DEEP_KS)));
  static bool attr_set = false;
  if (!attr_set) {
#define NT_ATTR(TT, SS, ...) \
    (void)hipFuncSetAttribute((const void*)conv_gemm_nt_kernel<TT, SS, true, __VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); \
    (void)hipFuncSetAttribute((const void*)conv_gemm_nt_kernel<TT, SS, false, __VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)
    NT_ATTR(float, 2, 2, 4, 8, 4); NT_ATTR(bf16_t, 2, 2, 4, 8, 4);
    NT_ATTR(float, 2, 2, 4, 4, 2); NT_ATTR(bf16_t, 2, 2, 4, 4, 2);
    NT_ATTR(float, 4, 2, 4, 4, 2); NT_ATTR(bf16_t, 4, 2, 4, 4, 2);
#ifdef DRN_EXPERIMENTS
    NT_ATTR(float, 2, 2, 2, 4, 4); NT_ATTR(bf16_t, 2, 2, 2, 4, 4); NT_ATTR(float, 4, 2, 2, 4, 4); NT_ATTR(bf16_t, 4, 2, 2, 4, 4);
#endif
#undef NT_ATTR
    attr_set = true;
  }
#define NT_LAUNCH(TT, SS, THREADS, LDS, ...) do { \
    P.nblocks = total; \
    if (fast) conv_gemm_nt_kernel<TT, SS, true, __VA_ARGS__><<<dim3(total, ksplit), THREADS, LDS, stream>>>(P); \
    else conv_gemm_nt_kernel<TT, SS, false, __VA_ARGS__><<<dim3(total, ksplit), THREADS, LDS, stream>>>(P); } while (0)
  const int kind = w4h ? DRN_NT_KIND_W4H : nt_kind(d, ngroups, dtype, tile, ksplit, planes256);
  if (plan_only) return kind;
  if (kind == DRN_NT_KIND_W4H) {
    P.nblocks = total;
    return drn_nt_w4h_launch(P, total, w4h_conv, stream, ksplit);
  }
  for (int g = 0; g < ngroups; ++g)
    DRN_CHECK_ARG(!d[g].sumsq || (kind == DRN_NT_KIND_W4 && d[g].out_f32 && !d[g].bias && !d[g].accumulate && ngroups == 1),
                  "drn_gemm_nt: DrnGemmDesc::sumsq needs a single fp32-output launch on gemm_nt_w4_kernel without bias / accumulate "
                  "(ask drn_gemm_nt_plan first)");
  for (int g = 0; g < ngroups; ++g) {
    const DrnGemmDesc& s = d[g];
    if (!s.gb_act) continue;
    DRN_CHECK_ARG(kind == DRN_NT_KIND_W4C && ngroups == 1 && ksplit == 1 && s.mode == 1 && s.gate && s.gb_dct && s.gb_dgate && s.gb_dsum &&
                      !s.bias && !s.C2 && !s.stats && !s.accumulate && (s.Lout == 32 || s.Lout == 64 || s.Lout == 128 || s.Lout == 256) &&
                      s.gb_ld_act % 8 == 0 && s.gb_ldt % 8 == 0 && !((uintptr_t)s.gb_act & 15) && !((uintptr_t)s.gb_dct & 15) && s.ldg >= s.N,
                  "drn_gemm_nt: DrnGemmDesc::gb_* needs a single bf16 data-gradient launch on gemm_nt_w4c_kernel (ask drn_gemm_nt_plan "
                  "first) with gate and all three outputs set, clips of 32 / 64 / 128 / 256 rows, no bias / C2 / stats / accumulate");
  }
  if (kind == DRN_NT_KIND_W4) {
    P.nblocks = total;
    return drn_nt_w4_launch(P, total, stream);
  }
  if (kind == DRN_NT_KIND_W4C) {
    P.nblocks = total;
    return drn_nt_w4c_launch(P, total, stream, ksplit);
  }
  if (planes256) {
    drn_set_error("drn_gemm_nt_splitk256: the problem is not one gemm_nt_w4c_kernel runs (bf16, k = 3 / stride 1 / pad 1, M and N multiples of 256, Cin of 64)");
    return DRN_ERR_UNSUPPORTED;
  }
  if (tile == 256) {
    if (dtype == DRN_BF16) NT_LAUNCH(bf16_t, 2, 512, 2 * 65536, 2, 4, 8, 4); else NT_LAUNCH(float, 2, 512, 2 * 65536, 2, 4, 8, 4);
  } else if (deep8) {     // few workgroups, cold operands: three tiles of loads in flight instead of one
    if (dtype == DRN_BF16) NT_LAUNCH(bf16_t, 4, 512, 4 * 32768, 2, 4, 4, 2); else NT_LAUNCH(float, 4, 512, 4 * 32768, 2, 4, 4, 2);
  }
#ifdef DRN_EXPERIMENTS
  else if (waves8 && stages == 4) {
    if (dtype == DRN_BF16) NT_LAUNCH(bf16_t, 4, 512, 4 * 32768, 2, 4, 4, 2); else NT_LAUNCH(float, 4, 512, 4 * 32768, 2, 4, 4, 2);
  } else if (!waves8 && dtype == DRN_BF16) {
    if (stages == 2) NT_LAUNCH(bf16_t, 2, 256, 2 * 32768, 2, 2, 4, 4); else NT_LAUNCH(bf16_t, 4, 256, 4 * 32768, 2, 2, 4, 4);
  } else if (!waves8) {
    if (stages == 2) NT_LAUNCH(float, 2, 256, 2 * 32768, 2, 2, 4, 4); else NT_LAUNCH(float, 4, 256, 4 * 32768, 2, 2, 4, 4);
  }
#endif
  else {      // 128x128 tile, 8 waves, 2-slot ring
    if (dtype == DRN_BF16) NT_LAUNCH(bf16_t, 2, 512, 2 * 32768, 2, 4, 4, 2); else NT_LAUNCH(float, 2, 512, 2 * 32768, 2, 4, 4, 2);
  }
#undef NT_LAUNCH
  return drn_launch_status("drn_gemm_nt");
}

extern "C" int drn_gemm_nt_plan(const DrnGemmDesc* descs, int ngroups, int dtype) {
  drn_clear_status();
  return launch_nt(descs, ngroups, dtype, nullptr, 1, nullptr, nullptr, false, true);
}

extern "C" int drn_gemm_nt_splitk_plan(const DrnGemmDesc* descs, int ngroups, int ksplit, int dtype) {
  drn_clear_status();
  ksplit &= 0xffff;
  DRN_CHECK_ARG(ksplit >= 1 && ksplit <= 64, "drn_gemm_nt_splitk_plan: bad ksplit");
  static float dummy_ws;           // (plan only: non-null workspace / counters so that the split kernels' preconditions read as met)
  static int dummy_cnt;
  return launch_nt(descs, ngroups, dtype, nullptr, ksplit, ksplit > 1 ? &dummy_ws : nullptr, ksplit > 1 ? &dummy_cnt : nullptr, false, true);
}

extern "C" int drn_gemm_nt(const DrnGemmDesc* descs, int ngroups, int dtype, void* stream) {
  drn_clear_status();
  return launch_nt(descs, ngroups, dtype, (hipStream_t)stream);
}

extern "C" int64_t drn_gemm_nt_splitk_ws_elems(int M, int N, int ksplit) {
  return (int64_t)ksplit * cdiv(M, 128) * cdiv(N, 128) * 128 * 128;
}

// The same for a GROUPED launch (pyramid levels / independent problems of one launch): workspace and counters are indexed by the
// launch-wide tile number, every problem splits its own K range `ksplit` ways.  Short sequences (Charades-STA's 32 proposals:
// 14-56 tiles per grouped launch on 256 CUs) are where this pays.  ws >= ksplit * (sum of 128x128 tiles) * 16384 floats.
extern "C" int drn_gemm_nt_splitk_grouped(const DrnGemmDesc* descs, int ngroups, int ksplit_arg, float* ws, int32_t* counters, int dtype,
                                          void* stream) {
  drn_clear_status();
  const int ksplit = ksplit_arg & 0xffff;
  DRN_CHECK_ARG(!(ksplit_arg & ~(0xffff | DRN_XCHG_CONFIRM | DRN_XCHG_NONE)), "drn_gemm_nt_splitk_grouped: unknown bits in ksplit");
  DRN_CHECK_ARG(descs && ngroups >= 1 && ngroups <= DRN_MAX_GROUPS && ksplit >= 1 && ksplit <= 64 && (ksplit == 1 || (ws && counters)),
                "drn_gemm_nt_splitk_grouped: bad groups/ksplit/workspace/counters");
  long tiles = 0;
  for (int g = 0; g < ngroups; ++g) tiles += (long)cdiv(descs[g].M, 128) * cdiv(descs[g].N, 128);
  DRN_CHECK_ARG(ksplit == 1 || (tiles <= DRN_QD_COUNTERS && (((uintptr_t)ws) & 15) == 0),
                "drn_gemm_nt_splitk_grouped: more than %d output tiles or unaligned workspace", DRN_QD_COUNTERS);
  return launch_nt(descs, ngroups, dtype, (hipStream_t)stream, ksplit_arg, ws, (int*)counters);
}

extern "C" int drn_gemm_nt_splitk(const DrnGemmDesc* desc, int ksplit_arg, float* ws, int32_t* counters, int dtype, void* stream) {
  drn_clear_status();
  const int ksplit = ksplit_arg & 0xffff;
  DRN_CHECK_ARG(!(ksplit_arg & ~(0xffff | DRN_XCHG_CONFIRM | DRN_XCHG_NONE)), "drn_gemm_nt_splitk: unknown bits in ksplit");
  DRN_CHECK_ARG(desc && ksplit >= 1 && ksplit <= 64 && (ksplit == 1 || (ws && counters)), "drn_gemm_nt_splitk: bad ksplit/workspace/counters");
  DRN_CHECK_ARG(ksplit == 1 || (cdiv(desc->M, 128) * cdiv(desc->N, 128) <= DRN_QD_COUNTERS && (((uintptr_t)ws) & 15) == 0),
                "drn_gemm_nt_splitk: more than %d output tiles or unaligned workspace", DRN_QD_COUNTERS);
  return launch_nt(desc, 1, dtype, (hipStream_t)stream, ksplit_arg, ws, (int*)counters);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Split-K on 256x256 tiles in TWO launches, for ONE long-K k = 3 convolution with few output tiles (conv0's forward: 8192 x 256 x
// 13056 = 32 tiles): with 128x128 tiles (the in-launch split above) every A panel is staged once per 128 output columns -- 856 MB
// from the L2s into LDS for conv0, which is what its 81 us were; full-width 256-column tiles halve that, and gemm_nt_w4c_kernel's
// five-slot ring keeps the loads coming (the same split on the 8-wave 256x256 kernel measured 94 us: one K-step in flight against
// operands that come straight from HBM).  Launch 1: grid (tiles, ksplit), split y writes its fp32 partial product as plane y of
// ws[ksplit][M][N].  Launch 2 (below): adds the planes in split order
// (deterministic), + bias, writes the output in `dtype` and the per-128-row-slab BatchNorm statistics of the fp32 sums -- what the
// one-launch epilogue would have written.  No gate / pre-gate copy / accumulate here (conv -> BN callers have none).
template <typename T>
__global__ __launch_bounds__(256) void splitk256_reduce_kernel(const float* __restrict__ ws, int ksplit, T* __restrict__ C, int ldc,
                                                               const float* __restrict__ bias, float* __restrict__ stats, int M, int N) {
  // one workgroup = one 128-row slab x 64 columns; thread = column tid % 64, rows (tid / 64) * 32 .. + 32
  __shared__ float sh[4][64];
  const int c = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int n = blockIdx.y * 64 + c, slab = blockIdx.x;
  const int r0 = slab * 128 + j * 32;
  const long plane = (long)M * N;
  const bool ncol = n < N;
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = 0.f;
  for (int s = 0; s < ksplit; ++s) {
    const float* p = ws + s * plane + (long)r0 * N + n;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] += (ncol && r0 + i < M) ? p[(long)i * N] : 0.f;
  }
  const float b = (bias && ncol) ? bias[n] : 0.f;
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    sm += v[i];                                                   // statistics of the RAW sums (rows >= M hold exact zeros)
    if (ncol && r0 + i < M) DT<T>::st(C + (long)(r0 + i) * ldc + n, v[i] + b);
  }
  if (!stats) return;
  sh[j][c] = sm;
  __syncthreads();
  const float tot = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
  const int rows = min(128, M - slab * 128);
  const float mean = tot / (float)rows;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const float dlt = v[i] - mean;
    q += r0 + i < M ? dlt * dlt : 0.f;
  }
  __syncthreads();
  sh[j][c] = q;
  __syncthreads();
  if (j == 0 && ncol) {
    stats[((long)slab * 2 + 0) * N + n] = tot;
    stats[((long)slab * 2 + 1) * N + n] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
  }
}

extern "C" int64_t drn_gemm_nt_splitk256_ws_elems(int M, int N, int ksplit) { return (int64_t)ksplit * M * N; }

extern "C" int drn_gemm_nt_splitk256(const DrnGemmDesc* desc, int ksplit, float* ws, int dtype, void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(desc && ksplit >= 2 && ksplit <= 64 && ws && (((uintptr_t)ws) & 15) == 0, "drn_gemm_nt_splitk256: bad ksplit / workspace");
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "drn_gemm_nt_splitk256: bad dtype %d", dtype);
  DRN_CHECK_ARG(!desc->gate && !desc->C2 && !desc->accumulate && !desc->out_f32 && desc->C,
                "drn_gemm_nt_splitk256: gate / pre-gate copy / accumulate / fp32 destination are not supported here");
  DRN_CHECK_ARG(desc->N % 4 == 0, "drn_gemm_nt_splitk256: N must be a multiple of 4 (16-byte rows of the fp32 planes)");
  {
    const int ksteps = desc->taps * desc->Cin / 64, per = cdiv(ksteps, ksplit);
    DRN_CHECK_ARG(ksteps - (cdiv(ksteps, per) - 1) * per >= 2 && cdiv(ksteps, per) == ksplit,
                  "drn_gemm_nt_splitk256: %d K-steps do not split %d ways with at least 2 per split", ksteps, ksplit);
  }
  DrnGemmDesc part = *desc;
  part.C = ws;                       // (gemm_nt_w4c_kernel addresses plane `split` of P.ws itself)
  part.ldc = desc->N;
  part.bias = nullptr;
  part.stats = nullptr;
  const int rc = launch_nt(&part, 1, dtype, stream, ksplit, ws, nullptr, true);
  if (rc != DRN_OK) return rc;
  const dim3 grid(cdiv(desc->M, 128), cdiv(desc->N, 64));
  if (dtype == DRN_BF16)
    splitk256_reduce_kernel<bf16_t><<<grid, 256, 0, stream>>>(ws, ksplit, (bf16_t*)desc->C, desc->ldc, desc->bias, desc->stats, desc->M, desc->N);
  else
    splitk256_reduce_kernel<float><<<grid, 256, 0, stream>>>(ws, ksplit, (float*)desc->C, desc->ldc, desc->bias, desc->stats, desc->M, desc->N);
  return drn_launch_status("drn_gemm_nt_splitk256(reduce)");
}
