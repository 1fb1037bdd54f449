// Does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` write ZEROS into its LDS cell (or leave it alone)?
// build: hipcc --offload-arch=gfx950 -O2 oob_lds.hip -o oob_lds ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint32_t* src, uint32_t* out, int nbytes) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xDEADBEEF;
  __syncthreads();
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rsrc;
  rsrc.x = (uint32_t)(uintptr_t)src;
  rsrc.y = (uint32_t)((uintptr_t)src >> 32);          // stride 0
  rsrc.z = (uint32_t)nbytes;                           // num_records (bytes)
  rsrc.w = 0x00020000;
  // lanes 0..31 in range, lane 5 and lanes 32..63 out of range
  uint32_t voff = threadIdx.x * 16;
  if (threadIdx.x == 5 || threadIdx.x >= 32) voff = 0x80000000u + threadIdx.x * 16;
  uint32_t soff = 64;                                  // scalar offset: NOT part of the range check?
  uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds;
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_waitcnt vmcnt(0)"
               : : "v"(voff), "s"(rsrc), "s"(soff), "s"(ldsbase) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  uint32_t *src, *out, h[256], hs[1024];
  for (int i = 0; i < 1024; ++i) hs[i] = 0x1000 + i;
  hipMalloc(&src, 4096); hipMalloc(&out, 1024);
  hipMemcpy(src, hs, 4096, hipMemcpyHostToDevice);
  k<<<1, 64, 4096>>>(src, out, 2048);
  hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  printf("err=%d\n", (int)hipGetLastError());
  for (int lane = 0; lane < 64; lane += 1) if (lane < 8 || lane == 31 || lane == 32 || lane == 63) printf("lane %2d: %08x %08x %08x %08x\n", lane, h[lane*4], h[lane*4+1], h[lane*4+2], h[lane*4+3]);
  return 0;
}
