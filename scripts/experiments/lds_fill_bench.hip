// Per-CU fill rate of LDS from L2-resident global memory on gfx950: LDS-DMA (global_load_lds, 16 B/lane) vs
// global_load_dwordx4 -> VGPR -> ds_write_b128 vs the loads alone.  One 512-thread workgroup per CU, 32 KB per iteration,
// each workgroup cycling over its own 64 KB window (2 MB per XCD: L2-resident).
//   hipcc --offload-arch=gfx950 -O3 lds_fill_bench.hip -o lds_fill_bench && ./lds_fill_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 2) void fill_kernel(const char* __restrict__ src, float* sink, int iters, long window) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const char* base = src + (long)blockIdx.x * window;
  f32x4 accv = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const long off = ((long)it * 32768) % window;
    char* stage = smem + (it % DEPTH) * 32768;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // 8 rows x 128 contiguous bytes per instruction (whole lines)
        const char* g = base + off + (w * 4 + j) * 1024 + l * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(stage + (w * 4 + j) * 1024), 16, 0, 0);
      }
      if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *(const f32x4*)(base + off + (w * 4 + j) * 1024 + l * 16);
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(stage + (w * 4 + j) * 1024 + l * 16) = v[j];
        __builtin_amdgcn_s_barrier();
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) accv += v[j];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 2) accv = *(const f32x4*)(smem + tid * 16);
  if (accv[0] == 123.456f) sink[tid] = accv[1];
}

template <int MODE, int DEPTH>
static void run(const char* name, const char* src, float* sink, int grid, long window) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)fill_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  fill_kernel<MODE, DEPTH><<<grid, 512, DEPTH * 32768>>>(src, sink, 50, window);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  fill_kernel<MODE, DEPTH><<<grid, 512, DEPTH * 32768>>>(src, sink, iters, window);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = (double)iters * 32768;
  printf("%-44s grid %3d window %6ld KB: %7.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip  (%.3f us / 32 KB)\n", name, grid, window >> 10, ms,
         bytes_per_cu / ms / 1e6, bytes_per_cu * grid / ms / 1e9, ms * 1e3 / iters);
}

int main() {
  char* src; float* sink;
  const long total = 256L * (4 << 20);
  hipMalloc(&src, total); hipMalloc(&sink, 4096);
  hipMemset(src, 1, total);
  for (int grid : {64, 256}) {
    for (long window : {65536L, 4L << 20}) {
      run<0, 1>("LDS-DMA, wait each iteration", src, sink, grid, window);
      run<0, 4>("LDS-DMA, one iteration in flight (4 slots)", src, sink, grid, window);
      run<1, 1>("global_load_dwordx4 + ds_write_b128", src, sink, grid, window);
      run<2, 1>("global_load_dwordx4 only", src, sink, grid, window);
    }
  }
  return 0;
}
