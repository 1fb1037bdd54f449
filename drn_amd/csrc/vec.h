// 16-byte vector access for the HBM-bound kernels: 8 bf16 or 4 f32 per lane per load.
#pragma once
#include "common.h"

template <typename T> struct V16;

template <> struct V16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const f32x4 t = *(const f32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = t[i];
  }
  // the 16 bytes as loaded (conversion deferred: a prefetched vector costs 4 registers, whatever T is)
  typedef f32x4 raw_t;
  static __device__ __forceinline__ raw_t ldraw(const float* p) { return *(const f32x4*)p; }
  static __device__ __forceinline__ raw_t ldraw_nt(const float* p) { return __builtin_nontemporal_load((const f32x4*)p); }      // read once
  static __device__ __forceinline__ void cvt(const raw_t& t, float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = t[i];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    f32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = v[i];
    *(f32x4*)p = t;
  }
};

template <> struct V16<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const bf16x8 t = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  typedef bf16x8 raw_t;
  static __device__ __forceinline__ raw_t ldraw(const bf16_t* p) { return *(const bf16x8*)p; }
  static __device__ __forceinline__ raw_t ldraw_nt(const bf16_t* p) { return __builtin_nontemporal_load((const bf16x8*)p); }
  static __device__ __forceinline__ void cvt(const raw_t& t, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    *(bf16x8*)p = t;
  }
};

static inline int ew_blocks(long work_items, int threads, int cap = 4096) {
  long b = (work_items + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

#define DISPATCH_DT(dtype, NAME, ...)                               \
  do {                                                              \
    if ((dtype) == DRN_BF16) { typedef bf16_t T; __VA_ARGS__; }     \
    else if ((dtype) == DRN_F32) { typedef float T; __VA_ARGS__; }  \
    else { drn_set_error(NAME ": bad dtype %d", (int)(dtype)); return DRN_ERR_ARG; } \
  } while (0)
