// what do s_memtime and s_memrealtime count on gfx950?  One wave spins on dependent VALU adds for a known number of instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long* out, int n) {
  const long long t0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
  float x = (float)threadIdx.x;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (long long)x; }
}
int main() {
  long long* d; hipMalloc(&d, 64);
  for (int n : {1000, 10000, 100000}) {
    probe<<<1, 64>>>(d, n); hipDeviceSynchronize();
    long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("n=%d  instr=%lld  memtime=%lld  realtime=%lld  memtime/realtime=%.3f  instr/realtime_tick=%.2f\n", n, 64LL * n, h[0], h[1], (double)h[0] / h[1], 64.0 * n / h[1]);
  }
  return 0;
}
