// Measurement infrastructure, not on any product path: what the chip SUSTAINS on bf16 MFMA with operands that toggle like real data.
// MI355X clocks to its power budget.  A register-only loop of v_mfma_f32_32x32x16_bf16 at the issue floor (32.0 cycles per MFMA and SIMD, no
// LDS, no memory) runs at 2.3 GHz on zeros and at 1.7-1.8 GHz on random bf16 operands: ~1.75-1.85 PFLOP/s, not the 2.5 PFLOP/s of the
// datasheet clock (scripts/experiments/clk/mfma_power.hip; profiles/HISTORY.md, round 6).  bench.py runs this next to the step and reports
// the GEMMs against BOTH numbers: a kernel cannot beat the second one by scheduling.
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/drn_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned diag_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// a bf16 pair in (-2, 2): random signs and mantissas, exponents 125..127 (zero_operands: all bits zero)
__device__ __forceinline__ unsigned diag_bf16_pair(unsigned h) {
  const unsigned lo = ((h & 1u) << 15) | ((125u + ((h >> 1) % 3u)) << 7) | ((h >> 3) & 127u);
  const unsigned hi = (((h >> 10) & 1u) << 15) | ((125u + ((h >> 11) % 3u)) << 7) | ((h >> 13) & 127u);
  return lo | (hi << 16);
}

__global__ __launch_bounds__(256, 1) void mfma_sustained_kernel(float* sink, long long* clk, int iters, int zero_operands) {
  const unsigned l = threadIdx.x, seed = (blockIdx.x * 256u + l) * 64u;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned wa[4], wb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wa[j] = zero_operands ? 0u : diag_bf16_pair(diag_hash(seed + i * 8 + j));
      wb[j] = zero_operands ? 0u : diag_bf16_pair(diag_hash(seed + i * 8 + 4 + j));
    }
    a[i] = __builtin_bit_cast(bf16x8, wa);
    b[i] = __builtin_bit_cast(bf16x8, wb);
  }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i & 1) + 2 * rep], b[(i >> 1) + 2 * rep], acc[i], 0, 0, 0);
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  sink[blockIdx.x * 256 + l] = s;
  if (l == 0) {
    clk[blockIdx.x * 2] = t1 - t0;            // shader cycles
    clk[blockIdx.x * 2 + 1] = r1 - r0;        // 100 MHz ticks
  }
}

// workspace: 256 workgroups x (256 floats + 2 int64) = 266,240 bytes (drn_diag_mfma_ws_bytes)
extern "C" long drn_diag_mfma_ws_bytes(void) { return 256L * (256 * 4 + 16); }

extern "C" int drn_diag_mfma_sustained(void* ws, int iters, int zero_operands, double* tflops, double* clock_ghz, double* cycles_per_mfma,
                                       void* stream) {
  DRN_CHECK_ARG(ws && iters >= 1 && tflops && clock_ghz && cycles_per_mfma, "drn_diag_mfma_sustained: null pointer or iters < 1");
  constexpr int NB = 256;
  float* sink = (float*)ws;
  long long* clk = (long long*)((char*)ws + (size_t)NB * 256 * 4);
  hipStream_t st = (hipStream_t)stream;
  drn_clear_status();
  hipLaunchKernelGGL(mfma_sustained_kernel, dim3(NB), dim3(256), 0, st, sink, clk, iters, zero_operands);
  const int rc = drn_launch_status("drn_diag_mfma_sustained");
  if (rc != DRN_OK) return rc;
  long long h[2 * NB];
  if (hipMemcpyAsync(h, clk, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    drn_set_error("drn_diag_mfma_sustained: read-back failed");
    return DRN_ERR_LAUNCH;
  }
  double cyc = 0.0, ticks = 0.0;
  for (int i = 0; i < NB; ++i) { cyc += (double)h[2 * i]; ticks += (double)h[2 * i + 1]; }
  if (ticks <= 0.0) {
    drn_set_error("drn_diag_mfma_sustained: the clocks did not advance");
    return DRN_ERR_LAUNCH;
  }
  const double mfmas = 8.0 * iters;                                   // per wave
  const double secs = ticks / NB * 1e-8;                              // mean loop time of a workgroup
  *tflops = 32768.0 * mfmas * 4 * NB / secs / 1e12;
  *clock_ghz = cyc / ticks / 10.0;
  *cycles_per_mfma = cyc / NB / mfmas;
  return DRN_OK;
}
