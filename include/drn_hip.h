/* libdrn_hip.so -- C-ABI of the MI355X (gfx950) DRN hot path.
 *
 * The reference (Alvin-Zeng/DRN) has no FFI on this path except the third-party
 * pybind extension fcos_core._C (model/layers/sigmoid_focal_loss.py:18,31); every
 * other entry point below replaces a stock ATen/cuDNN call made from the
 * reference's Python modules (file:line cited per function).  Conventions:
 *   - extern "C", plain pointers/ints, no torch types; caller owns every buffer
 *     (including workspaces); raw DEVICE pointers unless a parameter says "host";
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream);
 *     kernels are asynchronous on it, never synchronise, never allocate;
 *   - return 0 on success, negative on error (see drn_last_error()); never
 *     throws or aborts across the ABI; re-entrant (one process per GPU).
 *   - dtype: 0 = float32 (exact f32 MFMA, parity mode), 1 = bfloat16 storage
 *     with fp32 accumulation.  Activations are channels-last ("NLC"): row
 *     m = (sequence, t), C contiguous, explicit row stride in elements.
 */
#ifndef DRN_HIP_H
#define DRN_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRN_ABI_VERSION 1
#define DRN_MAX_GROUPS 4

int drn_abi_version(void);
const char* drn_last_error(void); /* thread-local, valid until the next failing call on this thread */

/* One problem of a grouped implicit-GEMM launch:  C[M][N] (+)= A'(M x K) * B[N][K]^T
 * where K = taps*Cin and A' is the im2col view of a channels-last tensor A:
 *   mode 0 (forward, model/basic_blocks.py:9-18 Conv1d / nn.Linear when taps==1):
 *       row m=(seq,t), column (tap,c) reads A[seq*Lsrc + t*stride + tap - pad][c]
 *   mode 1 (data gradient of the same conv):
 *       reads A[seq*Lsrc + (t + pad - tap)/stride][c] when divisible and in range
 *   out-of-range taps read zero.  B is [N][taps*Cin] (K contiguous).
 * Epilogue: + bias[n];  optional second output C2 = value before gating;
 *   * gate[(m / Lout)*ldg + n]  (query gating, model/backbone.py:28-30);
 *   per-tile column sums / sums of squares for train-mode BatchNorm
 *   (stats[(tile_m*2 + {0,1})*N + n], tile_m = m/128) -- deterministic, no atomics.
 */
typedef struct DrnGemmDesc {
  const void* A;
  const void* B;
  void* C;
  void* C2;          /* optional pre-gate copy of C (same ldc), or NULL */
  const float* bias; /* [N] fp32 or NULL */
  const float* gate; /* [(M/Lout)][ldg] fp32 or NULL */
  float* stats;      /* [ceil(M/128)][2][N] fp32 or NULL */
  int32_t M, N;
  int32_t Cin, taps, stride, pad, mode;
  int32_t Lout, Lsrc;
  int32_t lda, ldb, ldc, ldg;
  int32_t accumulate; /* 1: C += result */
} DrnGemmDesc;

/* Grouped NT implicit GEMM on MFMA (conv1d fwd / dgrad, linear fwd / dgrad).
 * Replaces nn.Linear (model/main_model.py:59), nn.Conv1d (model/basic_blocks.py:9,
 * model/fcos.py:33,37,59,65) forward and their input gradients. */
int drn_gemm_nt(const DrnGemmDesc* descs /*host*/, int ngroups, int dtype, void* stream);

/* Weight gradient:  dW[n][tap][c] (fp32) = sum_m dY[m][n] * X[src(m,tap)][c]   (mode-0 addressing of X).
 * dW is written as [N][taps][Cin] when w_layout==0 or [N][Cin][taps] (the nn.Conv1d parameter layout) when 1.
 * Grouped: the same dW accumulates over all groups (shared-weight heads, model/fcos.py:93-102).
 * `ws` is an fp32 workspace of drn_wgrad_ws_elems(...) elements (split over rows, reduced deterministically). */
typedef struct DrnWgradDesc {
  const void* dY; /* [M][N] */
  const void* X;  /* channels-last source */
  int32_t M, Lout, Lsrc;
  int32_t ldy, ldx;
} DrnWgradDesc;
int64_t drn_wgrad_ws_elems(int M_total, int N, int Cin, int taps);
int drn_gemm_wgrad(const DrnWgradDesc* descs /*host*/, int ngroups, float* dW, int N, int Cin, int taps, int stride,
                   int pad, int w_layout, int accumulate, float* ws, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
