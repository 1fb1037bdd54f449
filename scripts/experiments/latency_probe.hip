// Where do 15 us go in a 32-workgroup kernel with one memory round trip?  Stamps wall_clock64() (100 MHz) inside workgroup 0:
//   t0 entry, t1 after a dependent scalar load (lengths[b]), t2 after a vector load round trip, t3 after a barrier + LDS,
//   t4 after a second dependent vector load, t5 after the stores drained.  Host: HIP events around the launch, hot and after a
// cache flush.  build+run (GPU box): hipcc --offload-arch=gfx950 -O3 scripts/experiments/latency_probe.hip -o /tmp/lp && /tmp/lp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256) void probe(const long long* lens, const float* a, const float* b, float* out, long long* stamps, int C) {
  __shared__ float sh[256];
  const long long t0 = wall_clock64();
  const int len = (int)lens[blockIdx.x];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = wall_clock64();
  const float4 v = *(const float4*)(a + ((long)blockIdx.x * 8 + (len & 7)) * C + threadIdx.x * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = wall_clock64();
  sh[threadIdx.x] = v.x + v.y + v.z + v.w;
  __syncthreads();
  const float s = sh[(threadIdx.x + 64) & 255];
  const long long t3 = wall_clock64();
  const int idx = ((int)(s * 0.f) + threadIdx.x) * 4;
  const float4 u = *(const float4*)(b + (long)blockIdx.x * C + idx);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t4 = wall_clock64();
  *(float4*)(out + (long)blockIdx.x * C + threadIdx.x * 4) = make_float4(u.x + s, u.y, u.z, u.w);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t5 = wall_clock64();
  if (threadIdx.x == 0) {
    long long* st = stamps + blockIdx.x * 8;
    st[0] = t0; st[1] = t1; st[2] = t2; st[3] = t3; st[4] = t4; st[5] = t5;
  }
}
__global__ void flush(float* p, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] += 1.f;
}
__global__ void empty() {}
int main() {
  const int B = 32, C = 1024;
  long long* lens; float *a, *b, *out, *big; long long* stamps;
  hipMalloc(&lens, B * 8); hipMalloc(&a, (size_t)B * 8 * C * 4); hipMalloc(&b, (size_t)B * C * 4); hipMalloc(&out, (size_t)B * C * 4);
  hipMalloc(&stamps, B * 64); hipMalloc(&big, 1ul << 30);
  hipMemset(lens, 0, B * 8); hipMemset(a, 0, (size_t)B * 8 * C * 4); hipMemset(b, 0, (size_t)B * C * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    double tot = 0; std::vector<long long> h(B * 8); double d[5] = {0, 0, 0, 0, 0}; double span = 0;
    const int reps = 30;
    for (int r = 0; r < reps + 3; ++r) {
      if (mode == 1) flush<<<4096, 256>>>(big, 1l << 28);
      if (mode == 2) for (int k = 0; k < 20; ++k) empty<<<1, 64>>>();     // a train of tiny kernels in front (keeps the queue busy)
      hipEventRecord(e0);
      probe<<<B, 256>>>(lens, a, b, out, stamps, C);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), stamps, B * 64, hipMemcpyDeviceToHost);
      if (r >= 3) {
        tot += ms;
        long long lo = h[0], hi = h[5];
        for (int w = 0; w < B; ++w) { lo = h[w * 8] < lo ? h[w * 8] : lo; hi = h[w * 8 + 5] > hi ? h[w * 8 + 5] : hi; }
        span += (hi - lo) * 10.0;
        for (int k = 0; k < 5; ++k) d[k] += (h[k + 1] - h[k]) * 10.0;
      }
    }
    printf("%-28s events %.1f us | in-kernel span over all wgs %.2f us | wg0: scalar load %.0f ns, vector load %.0f ns, barrier+LDS %.0f ns, 2nd load %.0f ns, store drain %.0f ns\n",
           mode == 0 ? "hot" : (mode == 1 ? "after a 1 GB flush" : "after 20 empty kernels"), tot / reps * 1e3, span / reps / 1e3, d[0] / reps, d[1] / reps, d[2] / reps, d[3] / reps, d[4] / reps);
  }
  hipEventRecord(e0); empty<<<1, 64>>>(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); printf("empty kernel between events: %.1f us\n", ms * 1e3);
  return 0;
}
