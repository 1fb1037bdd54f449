// Which MFMA shape gets more work out of the power budget?  256 workgroups x 4 waves, operands in registers (random bf16 or zeros), no memory
// traffic in the loop: v_mfma_f32_16x16x32_bf16 (16 accumulator tiles) against v_mfma_f32_32x32x16_bf16 (4 tiles) -- the same FLOPs per cycle
// on paper, half the operand-register reads per FLOP for the second.  Prints TFLOP/s, the effective clock (s_memtime / wall) and cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void burn(const bf16x8* __restrict__ src, float* out, long long* clk, int iters) {
  const int l = threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(blockIdx.x * 8 + i) * 256 + l]; b[i] = src[(blockIdx.x * 8 + 4 + i) * 256 + l]; }
  const long long t0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)wall_clock64();
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i & 1) + 2 * rep], b[(i >> 1) + 2 * rep], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)wall_clock64();
  out[blockIdx.x * 256 + l] = s;
  if (l == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  const int nb = 256, iters = 20000;
  bf16x8* src; float* out; long long* clk;
  hipMalloc(&src, (size_t)nb * 8 * 256 * 16); hipMalloc(&out, nb * 256 * 4); hipMalloc(&clk, nb * 16);
  unsigned short* h = (unsigned short*)malloc((size_t)nb * 8 * 256 * 16);
  for (int mode = 0; mode < 2; ++mode) {
    srand(1);
    for (size_t i = 0; i < (size_t)nb * 8 * 256 * 8; ++i) {
      // random bf16 in (-2, 2): random sign and mantissa, exponent 125..127
      h[i] = mode == 0 ? (unsigned short)(((rand() & 1) << 15) | ((125 + rand() % 3) << 7) | (rand() & 127)) : 0;
    }
    hipMemcpy(src, h, (size_t)nb * 8 * 256 * 16, hipMemcpyHostToDevice);
    for (int shape : {16, 32, 16, 32}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 16) burn<16><<<nb, 256>>>(src, out, clk, iters); else burn<32><<<nb, 256>>>(src, out, clk, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long hc[2 * 256]; hipMemcpy(hc, clk, nb * 16, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0; for (int i = 0; i < nb; ++i) { cyc += hc[2 * i]; wall += hc[2 * i + 1]; }
      const double nm = shape == 16 ? 16.0 * iters : 8.0 * iters;             // MFMAs per wave
      const double flops = (shape == 16 ? 16384.0 : 32768.0) * nm * 4 * nb;
      printf("%s operands, %s: %.3f ms, %.0f TFLOP/s, clock %.2f GHz, %.1f cycles per MFMA\n", mode == 0 ? "random" : "zero  ",
             shape == 16 ? "16x16x32" : "32x32x16", ms, flops / ms / 1e9, cyc / wall / 10.0, cyc / nb / nm);
    }
  }
  return 0;
}
