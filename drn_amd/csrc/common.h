// Shared device/host helpers for libdrn_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define DRN_OK 0
#define DRN_ERR_ARG (-1)
#define DRN_ERR_LAUNCH (-2)
#define DRN_ERR_UNSUPPORTED (-3)

#define DRN_F32 0
#define DRN_BF16 1

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// A pointer the compiler cannot trace to a kernel argument (assembled from lanes, loaded from a table in memory) is a GENERIC address to
// it and every access through it a FLAT instruction: those count on the LDS counter as well as the memory one, so each wait for an LDS
// access also waits for the loads and stores in flight (an epilogue that stages tiles through LDS serialises against its own global
// stores), and they cannot use the scalar-base addressing mode.  as_global() states what the host guarantees -- it is device memory.
template <typename T>
__device__ __forceinline__ T* as_global(T* p) {
  typedef __attribute__((address_space(1))) T* g_t;
  g_t g = (g_t)p;
  asm("" : "+s"(g));        // (opaque, or the two casts fold back into the generic pointer; the addresses this is used for are wave-uniform)
  return (T*)g;
}
template <typename T>
__device__ __forceinline__ T* as_global_v(T* p) {           // the same for an address that differs from lane to lane
  typedef __attribute__((address_space(1))) T* g_t;
  g_t g = (g_t)p;
  asm("" : "+v"(g));
  return (T*)g;
}


void drn_set_error(const char* fmt, ...);

#define DRN_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      drn_set_error(__VA_ARGS__);         \
      return DRN_ERR_ARG;                 \
    }                                     \
  } while (0)

// HIP's "last error" is sticky per thread: clear whatever an unrelated earlier runtime call left behind
// before launching, so drn_launch_status() reports only this entry point's own launches.
static inline void drn_clear_status() { (void)hipGetLastError(); }

static inline int drn_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    drn_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return DRN_ERR_LAUNCH;
  }
  return DRN_OK;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Experiment switches: the shipped library carries the measured-best kernel variants only and never reads the environment.
// `make EXPERIMENTS=1` (-DDRN_EXPERIMENTS) compiles the alternative variants back in and lets DRN_* variables select them
// (DESIGN.md section 9).
#ifdef DRN_EXPERIMENTS
#include <stdlib.h>
static inline const char* drn_exp_env(const char* name) { return getenv(name); }
#else
static inline const char* drn_exp_env(const char*) { return nullptr; }
#endif
// Two tuning values tests need (force the fused-tap weight-gradient kernel on small shapes / switch it off): set explicitly
// through drn_tune(), process-wide, not read from the environment.
int drn_tuning(int key);
// Flags in GemmParams::ksplit (= the public DRN_KSPLIT_CONFIRM_* bits of the `ksplit` argument of drn_gemm_nt_splitk*, include/drn_hip.h;
// launch_nt maps "no bit" to READBACK): how a split confirms its partial-tile stores before it takes its ticket.  CONFIRM: a returning
// agent-scope read-modify-write of every stored address (~6 us per launch, +76 us per step); READBACK (the default): an sc1 load of every
// 64-byte request (+12 us per step) -- needed whenever a kernel of ANOTHER queue may run beside the launch (qdense.hip,
// skinny_group_kernel, has the measurements), and a caller cannot prove a single queue from where it stands, so it is what ships.
#define DRN_XCHG_CONFIRM 0x20000
#define DRN_XCHG_READBACK 0x40000
#define DRN_XCHG_NONE 0x80000     // (request bit only: never reaches a kernel)
#define DRN_TUNE_TN3_MINROWS 0
#define DRN_TUNE_TN_FUSED 1
#define DRN_TUNE_EXP0 3         // exp0..exp4: experiment overrides, 0 = shipped behaviour (scripts/experiments/ab_tune.sh A/Bs them inside
                                // one process): exp0 = 256x256-tile threshold of the NT launches (200 big tiles), exp1 = workgroup
                                // target of the fused-tap weight gradient (256), exp2 = of the per-tap one (768), exp3 = NT tile order + 1, exp4 = of the K-split skinny launches (256)
#define DRN_TUNE_NT_W4C 9        // 1: eligible k = 3 / stride 1 bf16 convolutions on 256x256 tiles run gemm_nt_w4c_kernel (gemm_nt_w4.hip)
#define DRN_TUNE_NT_W4 8         // 1: eligible large bf16 products run the 4-wave hand-scheduled kernel (gemm_nt_w4.hip); 0: the general kernel
#define DRN_TUNE_NT_DEEP 2       // > 0: 128x128 NT launches of at most that many workgroups run the 4-slot ring (one workgroup per CU)
#define DRN_TUNE_NT_W4H 12       // > 0 (160; 128 loses at T = 32, where prop_fc makes 128 such tiles: 1.234 vs 1.220 ms): eligible bf16 launches that would run 128x128 tiles and make at least that many 256x128 tiles run gemm_nt_w4h_kernel
#define DRN_TUNE_BN1_MAXWG 13   // drn_bn_bwd_one: the smallest row block whose grid is at most this many workgroups (512)
#define DRN_TUNE_W4H_TAPIL 14   // > 0: split k = 3 launches of gemm_nt_w4h_kernel with at least that many input channels walk K as (channel block, tap)
#define DRN_TUNE_W4H_HALO 15    // 1: k = 3 launches of gemm_nt_w4h_kernel whose sequences are multiples of 64 rows stage a channel block ONCE for its three taps (W4HX_LOOP_ASM)
#define DRN_TUNE_NT_DEEP2 10     // > 0: ... and launches of at most THAT many workgroups too when no tile has more than DRN_TUNE_NT_DEEP_KS K-steps (the FPN
#define DRN_TUNE_NT_DEEP_KS 11   // laterals: 448 tiles of 4-16 K-steps each, where a tile is its own load latency: three K-steps in flight instead of one)

// ---- device helpers -------------------------------------------------------
// zero-initialised source for masked 16-byte global_load_lds (one copy per translation unit)
static __device__ uint4 g_zero_page[4];

template <typename T> struct DT;
template <> struct DT<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float round(float v) { return v; }          // value as a T would hold it
};
template <> struct DT<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)v; }
  static __device__ __forceinline__ float round(float v) { return (float)(bf16_t)v; }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sh /* >= 17 floats */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  if (w == 0) {
    float t = l < nw ? sh[l] : 0.f;
    t = wave_sum(t);
    if (l == 0) sh[16] = t;
  }
  __syncthreads();
  return sh[16];
}
