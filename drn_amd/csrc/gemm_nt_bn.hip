// drn_conv_bn_train: Conv1d -> BatchNorm1d (training) -> ReLU as ONE launch of the implicit-GEMM kernel
// (gemm_nt_kernel.h: conv_gemm_nt_kernel<..., BNF = true> / nt_epilogue_bn).  Replaces drn_gemm_nt + drn_bn_train_apply for
// the conv blocks of model/basic_blocks.py:9-31, the FPN laterals / output convs (model/FPN.py:54-69) and the head towers
// (model/fcos.py:33-69, statistics per level call: fcos.py:93-102).
#define DRN_NT_PHASES_NAME drn_debug_nt_phases_bn
#include "gemm_nt_kernel.h"

// Workgroups of a kernel variant the chip holds at once (the column wait needs the whole grid resident).
template <typename KernelT>
static int nt_resident_capacity(KernelT kernel, int threads, int lds_bytes) {
  int dev = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  static int cus[64];
  if (dev < 0 || dev >= 64) return 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cus[dev] = n;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, threads, (size_t)lds_bytes) != hipSuccess) return 0;
  // A variant that spills to scratch cannot count on the occupancy the register / LDS arithmetic promises: the waves that may
  // hold scratch at once are limited per shader engine (measured: ~415 of 448 eight-wave workgroups resident, the rest starting
  // only after the first had given up waiting).  One workgroup per CU is what such a variant is trusted with.
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, (const void*)kernel) != hipSuccess) return 0;
  if (fa.localSizeBytes > 0 && per_cu > 1) per_cu = 1;
  return per_cu * cus[dev];
}

extern "C" int64_t drn_conv_bn_train_ws_bytes(const DrnGemmDesc* d, int ngroups) {
  int64_t n = 0;
  for (int g = 0; g < ngroups; ++g) n += (int64_t)cdiv(d[g].M, 128) * 2 * d[g].N * 8;
  return n;
}

extern "C" int drn_conv_bn_train(const DrnGemmDesc* d, const DrnBnTrainDesc* bn, int ngroups, int relu, const int32_t* up_group,
                                 void* tagged_ws, int64_t ws_bytes, int32_t* generation, int dtype, void* stream_) {
  drn_clear_status();
  const char* who = "drn_conv_bn_train";
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(d && bn && tagged_ws && generation && ngroups >= 1 && ngroups <= DRN_MAX_GROUPS, "%s: bad arguments", who);
  DRN_CHECK_ARG(((uintptr_t)tagged_ws & 7) == 0 && ws_bytes >= drn_conv_bn_train_ws_bytes(d, ngroups),
                "%s: statistics workspace too small or misaligned (drn_conv_bn_train_ws_bytes)", who);
  DRN_CHECK_ARG(dtype == DRN_F32 || dtype == DRN_BF16, "%s: bad dtype %d", who, dtype);
  const int ch = dtype == DRN_BF16 ? 8 : 4;
  long big_tiles = 0;
  bool fast = true;
  const int N = d[0].N;
#define BN_UNSUPPORTED(cond, ...) do { if (cond) { drn_set_error(__VA_ARGS__); return DRN_ERR_UNSUPPORTED; } } while (0)
  for (int g = 0; g < ngroups; ++g) {
    const DrnGemmDesc& s = d[g];
    const DrnBnTrainDesc& b = bn[g];
    DRN_CHECK_ARG(s.A && s.B && s.C, "%s: null operand in group %d", who, g);
    DRN_CHECK_ARG(s.M > 0 && s.N > 0 && s.Cin > 0 && s.taps >= 1 && s.stride >= 1, "%s: bad dims in group %d", who, g);
    DRN_CHECK_ARG(s.Cin % ch == 0 && s.lda % ch == 0 && s.ldb % ch == 0, "%s: Cin/lda/ldb must be multiples of %d elements", who, ch);
    DRN_CHECK_ARG(((uintptr_t)s.A & 15) == 0 && ((uintptr_t)s.B & 15) == 0, "%s: A/B must be 16-byte aligned", who);
    DRN_CHECK_ARG(s.Lout > 0 && s.Lsrc > 0 && s.M % s.Lout == 0, "%s: M=%d not a multiple of Lout=%d", who, s.M, s.Lout);
    DRN_CHECK_ARG(b.raw == s.C && b.M == s.M && b.L == s.Lout && b.tiles == cdiv(s.M, 128) && b.ld_raw == s.ldc,
                  "%s: group %d: the BatchNorm descriptor does not describe the GEMM's output", who, g);
    DRN_CHECK_ARG(b.out && b.scale_shift && b.save && b.gamma && b.beta, "%s: group %d: null BatchNorm operand", who, g);
    DRN_CHECK_ARG((b.gate != nullptr) == (b.gated != nullptr), "%s: gate and gated must come together", who);
    BN_UNSUPPORTED(s.C2 || s.bias || s.gate || s.accumulate || s.out_f32 || s.mode != 0,
                   "%s: group %d: GEMM epilogue options (C2 / bias / gate / accumulate / out_f32 / mode 1) do not combine with the fused BatchNorm", who, g);
    BN_UNSUPPORTED(s.N != N || N % 128 != 0, "%s: all groups need the same N, a multiple of 128 (got %d / %d)", who, s.N, N);
    BN_UNSUPPORTED(s.ldc % ch != 0 || b.ld_out % ch != 0 || ((uintptr_t)s.C & 15) || ((uintptr_t)b.out & 15) ||
                   (b.gated && (b.ld_gated % ch != 0 || b.ldg % 4 != 0 || ((uintptr_t)b.gated & 15) || ((uintptr_t)b.gate & 15))),
                   "%s: group %d: outputs must be 16-byte aligned with 16-byte row strides", who, g);
    big_tiles += (long)cdiv(s.M, 256) * cdiv(s.N, 256);
    if (s.Cin % (8 * ch) != 0) fast = false;
  }
  // the same tile choice as drn_gemm_nt (the per-slab statistics are summed in tile order: equal tiles = equal bits)
  const int big_min = drn_tuning(DRN_TUNE_EXP0) > 0 ? drn_tuning(DRN_TUNE_EXP0) : 200;
  int tile = big_tiles >= big_min ? 256 : 128;
  if (const char* e = drn_exp_env("DRN_NT_TILE")) tile = atoi(e) == 256 ? 256 : 128;
  BN_UNSUPPORTED(N % tile != 0, "%s: N=%d is not a multiple of the %d-wide tile", who, N, tile);
  GemmParamsBn P;
  memset(&P, 0, sizeof(P));
  P.ngroups = ngroups;
  P.ksplit = 1;
  P.xcd_swizzle = drn_exp_env("DRN_NO_XCD_SWIZZLE") ? 0 : 3;
  if (drn_tuning(DRN_TUNE_EXP0 + 3) > 0) P.xcd_swizzle = drn_tuning(DRN_TUNE_EXP0 + 3) - 1;
  if (const char* e = drn_exp_env("DRN_NT_ORDER")) P.xcd_swizzle = atoi(e);
  int total = 0;
  bool chain = false;
  unsigned long long* tws = (unsigned long long*)tagged_ws;
  for (int g = 0; g < ngroups; ++g) {
    const DrnGemmDesc& s = d[g];
    const DrnBnTrainDesc& b = bn[g];
    GemmProb& p = P.p[g];
    p.A = s.A; p.B = s.B; p.C = s.C; p.gate = b.gate;
    p.stats = (float*)tws;                       // this group's TAGGED statistics pairs [slab][2][N] x 8 bytes
    tws += (long)cdiv(s.M, 128) * 2 * s.N;
    p.M = s.M; p.N = s.N; p.K = s.taps * s.Cin; p.Cin = s.Cin; p.taps = s.taps; p.stride = s.stride; p.pad = s.pad;
    p.mode = 0; p.Lout = s.Lout; p.Lsrc = s.Lsrc; p.lda = s.lda; p.ldb = s.ldb; p.ldc = s.ldc; p.ldg = b.ldg;
    p.tiles_n = cdiv(s.N, tile);
    p.tile_start = total;
    total += cdiv(s.M, tile) * p.tiles_n;
    BnFuse& F = P.bn[g];
    F.out = b.out; F.gated = b.gated; F.ss = b.scale_shift; F.save = b.save; F.gamma = b.gamma; F.beta = b.beta; F.cbias = b.conv_bias;
    F.rm = b.running_mean; F.rv = b.running_var; F.momentum = b.momentum; F.eps = b.eps;
    F.ld_out = b.ld_out; F.ld_gated = b.ld_gated; F.slabs = b.tiles;
    F.up_group = up_group ? up_group[g] : -1;
    if (F.up_group >= 0) {
      // out_g += nearest_x2(out_h): h is a later (coarser) group with half the sequence length and the same clips
      const int h = F.up_group;
      DRN_CHECK_ARG(h > g && h < ngroups && d[h].Lout * 2 == s.Lout && d[h].M * 2 == s.M && bn[g].up == bn[h].out,
                    "%s: group %d: bad upsample source %d", who, g, h);
      chain = true;
    } else {
      BN_UNSUPPORTED(b.up != nullptr, "%s: group %d: an upsample source outside the launch is not supported", who, g);
    }
    // running statistics: the first group that carries a module's buffers updates them for every later group that shares them
    F.rs_owner = 0;
    F.rs_mask = 0;
    if (b.running_mean || b.running_var) {
      bool first = true;
      for (int h = 0; h < g; ++h)
        if (bn[h].running_mean == b.running_mean && bn[h].running_var == b.running_var) first = false;
      if (first) {
        F.rs_owner = 1;
        for (int h = g + 1; h < ngroups; ++h)
          if (bn[h].running_mean == b.running_mean && bn[h].running_var == b.running_var) F.rs_mask |= 1 << h;
      }
    }
  }
  for (int g = 0; g < ngroups; ++g) {          // the chain kernel normalises at most three levels per workgroup (the FPN has three)
    int depth = 1;
    for (int h = P.bn[g].up_group; h >= 0; h = P.bn[h].up_group) ++depth;
    BN_UNSUPPORTED(depth > 3, "%s: upsample chain deeper than three levels", who);
  }
  P.counters = (int*)generation;               // the launch generation word (read by every workgroup, advanced by workgroup 0)
  P.bn_relu = relu;
  P.bn_chain = chain ? 1 : 0;
  P.nblocks = total;
  const bool deep8 = tile == 128 && drn_tuning(DRN_TUNE_NT_DEEP) > 0 && total <= drn_tuning(DRN_TUNE_NT_DEEP) && !drn_exp_env("DRN_NT_STAGES");

  static bool attr_set = false;
  if (!attr_set) {
#define BN_ATTR(TT, SS, ...) \
    (void)hipFuncSetAttribute((const void*)conv_gemm_nt_kernel<TT, SS, true, __VA_ARGS__, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); \
    (void)hipFuncSetAttribute((const void*)conv_gemm_nt_kernel<TT, SS, false, __VA_ARGS__, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)
    BN_ATTR(float, 2, 2, 4, 8, 4); BN_ATTR(bf16_t, 2, 2, 4, 8, 4);
    BN_ATTR(float, 2, 2, 4, 4, 2); BN_ATTR(bf16_t, 2, 2, 4, 4, 2);
    BN_ATTR(float, 4, 2, 4, 4, 2); BN_ATTR(bf16_t, 4, 2, 4, 4, 2);
#undef BN_ATTR
#define BN_ATTRC(TT, SS) \
    (void)hipFuncSetAttribute((const void*)conv_gemm_nt_kernel<TT, SS, true, 2, 4, 4, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)
    BN_ATTRC(float, 2); BN_ATTRC(bf16_t, 2); BN_ATTRC(float, 4); BN_ATTRC(bf16_t, 4);
#undef BN_ATTRC
    attr_set = true;
  }
  // capacity[variant]: resident workgroups of the variant (0 = not asked yet); the wait inside the kernel needs total <= capacity
  static int capacity[2][3][2][2];
  const int vi = tile == 256 ? 0 : (deep8 ? 2 : 1);
  int& cap = capacity[dtype == DRN_BF16][vi][fast][chain];
  // the chain variant exists for the 128x128 tile on channel counts that are multiples of a K-step (the FPN laterals)
  BN_UNSUPPORTED(chain && (tile == 256 || !fast), "%s: the upsample chain needs 128x128 tiles and Cin %% %d == 0", who, 8 * ch);
#define BN_LAUNCHC(TT, SS, LDS) do { \
    if (!cap) cap = nt_resident_capacity(conv_gemm_nt_kernel<TT, SS, true, 2, 4, 4, 2, true, true>, 512, LDS); \
    BN_UNSUPPORTED(total > cap, "%s: %d workgroups exceed the %d the chip holds at once", who, total, cap); \
    conv_gemm_nt_kernel<TT, SS, true, 2, 4, 4, 2, true, true><<<dim3(total, 1), 512, LDS, stream>>>(P); } while (0)
#define BN_LAUNCH(TT, SS, LDS, ...) do { \
    if (fast) { \
      if (!cap) cap = nt_resident_capacity(conv_gemm_nt_kernel<TT, SS, true, __VA_ARGS__, true>, 512, LDS); \
      BN_UNSUPPORTED(total > cap, "%s: %d workgroups exceed the %d the chip holds at once", who, total, cap); \
      conv_gemm_nt_kernel<TT, SS, true, __VA_ARGS__, true><<<dim3(total, 1), 512, LDS, stream>>>(P); \
    } else { \
      if (!cap) cap = nt_resident_capacity(conv_gemm_nt_kernel<TT, SS, false, __VA_ARGS__, true>, 512, LDS); \
      BN_UNSUPPORTED(total > cap, "%s: %d workgroups exceed the %d the chip holds at once", who, total, cap); \
      conv_gemm_nt_kernel<TT, SS, false, __VA_ARGS__, true><<<dim3(total, 1), 512, LDS, stream>>>(P); \
    } } while (0)
  if (chain) {
    if (deep8) { if (dtype == DRN_BF16) BN_LAUNCHC(bf16_t, 4, 4 * 32768); else BN_LAUNCHC(float, 4, 4 * 32768); }
    else { if (dtype == DRN_BF16) BN_LAUNCHC(bf16_t, 2, 2 * 32768); else BN_LAUNCHC(float, 2, 2 * 32768); }
  } else if (tile == 256) {
    if (dtype == DRN_BF16) BN_LAUNCH(bf16_t, 2, 2 * 65536, 2, 4, 8, 4); else BN_LAUNCH(float, 2, 2 * 65536, 2, 4, 8, 4);
  } else if (deep8) {
    if (dtype == DRN_BF16) BN_LAUNCH(bf16_t, 4, 4 * 32768, 2, 4, 4, 2); else BN_LAUNCH(float, 4, 4 * 32768, 2, 4, 4, 2);
  } else {
    if (dtype == DRN_BF16) BN_LAUNCH(bf16_t, 2, 2 * 32768, 2, 4, 4, 2); else BN_LAUNCH(float, 2, 2 * 32768, 2, 4, 4, 2);
  }
#undef BN_LAUNCH
#undef BN_LAUNCHC
#undef BN_UNSUPPORTED
  return drn_launch_status(who);
}

// Watchdog of the column wait: number of workgroups that gave up after 2 s (a co-residency assumption broken, e.g. another
// process holding CUs while also waiting).  Results of such a launch are invalid.  Synchronises the device.
extern "C" int drn_conv_bn_train_timeouts(int reset) {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_bn_fuse_timeouts), sizeof(int)) != hipSuccess) return -1;
  if (reset && v) {
    const int z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bn_fuse_timeouts), &z, sizeof(int));
  }
  return v;
}
