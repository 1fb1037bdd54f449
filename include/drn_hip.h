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

#define DRN_ABI_VERSION 9
#define DRN_MAX_GROUPS 4

int drn_abi_version(void);
/* TEST / EXPERIMENT switches, process-wide; nothing on the product path (drn_amd/*.py outside tests) calls this.  They let tests reach
 * kernels their shapes would not select: "tn3_minrows" (fewest rows for the fused-tap weight-gradient kernel, default 4096), "tn_fused"
 * (0 switches that kernel off), "nt_w4" (0: drn_gemm_nt runs the general 8-wave kernel also for the large plain bf16 products that
 * gemm_nt_w4_kernel takes by default; results are bit-identical either way), "nt_w4c" / "nt_w4h" (the same for the k = 3 convolutions on
 * 256x256 tiles and the 256x128-tile kernel), "nt_deep*", "bn1_maxwg", "w4h_tapil", "exp0".."exp4" (launch heuristics, 0 = shipped).
 * Per-launch behaviour (the split-K exchange's confirmation, deferred weight-gradient reduces) is NOT here: it travels in the call's own
 * arguments (DRN_KSPLIT_CONFIRM_*, DrnWgradPending).  The library never reads the environment (the experiment build
 * `make EXPERIMENTS=1` does). */
int drn_tune(const char* key, int value);
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
 *   per-128-row-slab BatchNorm statistics in (sum, M2) form: stats[(slab*2 + 0)*N + n] = sum of the slab's rows,
 *   stats[(slab*2 + 1)*N + n] = sum of squared deviations from the SLAB mean (merged by drn_bn_finalize with the
 *   parallel-variance formula: no E[x^2]-E[x]^2 cancellation) -- deterministic, no atomics.
 */
typedef struct DrnGemmDesc {
  const void* A;
  const void* B;
  void* C;
  void* C2;          /* optional pre-gate copy of C (row stride ldc2), or NULL */
  const float* bias; /* [N] fp32 or NULL */
  const float* gate; /* [(M/Lout)][ldg] fp32 or NULL */
  float* stats;      /* [ceil(M/128)][2][N] fp32 or NULL */
  int32_t M, N;
  int32_t Cin, taps, stride, pad, mode;
  int32_t Lout, Lsrc;
  int32_t lda, ldb, ldc, ldg;
  int32_t accumulate; /* 1: C += result */
  int32_t ldc2;       /* row stride of C2 */
  int32_t out_f32;    /* 1: C is fp32 [M][ldc] regardless of dtype (no C2 / gate / stats): weight gradients as an NT product */
  float* sumsq;       /* or NULL.  out_f32 launches on gemm_nt_w4_kernel only (drn_gemm_nt_plan says DRN_NT_KIND_W4; no bias / accumulate):
                       * sumsq[t] = sum of the squares of output tile t's values (256x256 tiles in launch order, (M/256)*(N/256) floats):
                       * the gradient's contribution to the global norm without reading it again (drn_sumsq_finalize2) */
  /* gb_act != NULL: a data gradient (mode 1, k = 3 / stride 1) whose consumer is the input stage's gate backward (drn_gate_bwd_t;
   * model/backbone.py:28-30 on prop_fc's output): C is NOT written; with g = the product rounded to dtype, s = m / Lout,
   *   gb_dct[c][m] = dtype(g[m][c] * gate[s][c])   (row stride gb_ldt: the K-major operand of prop_fc's weight gradient),
   *   gb_dgate[s][c] = sum_t g * gb_act[m][c]       ([M / Lout][N] fp32; gb_act: the pre-gate activation, row stride gb_ld_act),
   *   gb_dsum[s][c]  = sum_t g * gate[s][c]         ([M / Lout][N] fp32: per-clip column sums, the bias gradient's partials).
   * bf16 launches on gemm_nt_w4c_kernel only (drn_gemm_nt_plan says DRN_NT_KIND_W4C), Lout in {32, 64, 128, 256}, gate set,
   * no bias / C2 / stats / accumulate; one problem per launch. */
  const void* gb_act;
  void* gb_dct;
  float* gb_dgate;
  float* gb_dsum;
  int32_t gb_ld_act, gb_ldt;
} DrnGemmDesc;

/* Grouped NT implicit GEMM on MFMA (conv1d fwd / dgrad, linear fwd / dgrad).
 * Replaces nn.Linear (model/main_model.py:59), nn.Conv1d (model/basic_blocks.py:9,
 * model/fcos.py:33,37,59,65) forward and their input gradients. */
int drn_gemm_nt(const DrnGemmDesc* descs /*host*/, int ngroups, int dtype, void* stream);
/* Which kernel drn_gemm_nt would run these problems on (no launch; arguments validated the same way): >= 0 one of the kinds
 * below, < 0 an error code.  For callers that schedule around a launch (functional.input_prep pre-touches the prop_fc weight
 * only when the general kernel runs) instead of re-deriving the library's rule. */
#define DRN_NT_KIND_TILE128 0  /* conv_gemm_nt_kernel, 128x128 tiles */
#define DRN_NT_KIND_TILE256 1  /* conv_gemm_nt_kernel, 256x256 tiles */
#define DRN_NT_KIND_W4 2       /* gemm_nt_w4_kernel: large plain bf16 products, 4 waves, 5-slot ring */
#define DRN_NT_KIND_W4C 3      /* gemm_nt_w4c_kernel: the same loop for k = 3 / stride 1 convolutions */
#define DRN_NT_KIND_W4H 4      /* gemm_nt_w4h_kernel: the same loop on 256x128 tiles (N too narrow to fill the chip with 256x256) */
int drn_gemm_nt_plan(const DrnGemmDesc* descs /*host*/, int ngroups, int dtype);
/* ... and which kernel drn_gemm_nt_splitk / _grouped would run them on with the K loop split ksplit ways. */
int drn_gemm_nt_splitk_plan(const DrnGemmDesc* descs /*host*/, int ngroups, int ksplit, int dtype);
/* Same for ONE problem with the K loop split `ksplit` ways (few output tiles: conv0, the coarse pyramid levels), in ONE
 * launch: every split publishes its fp32 partial tile in ws (drn_gemm_nt_splitk_ws_elems floats, 16-byte aligned), the
 * split that arrives last at a tile adds them in split order (deterministic) and runs the epilogue.  counters: >= one int32
 * per 128x128 output tile (<= DRN_QD_COUNTERS), zero on entry, left zero (the buffer drn_skinny_group uses will do). */
/* `ksplit` = the split count (1..64) in the low 16 bits, optionally OR-ed with ONE of the bits below: how a split confirms its
 * write-through partial-tile stores before it takes the tile's ticket.  No bit (what drn_amd.ops passes): an sc1 load of every stored
 * 64-byte request -- needed whenever a kernel of another queue may run beside the launch (a copy stream, a second graph branch, RCCL),
 * which the library cannot rule out; + 12 us per training step at T = 256.  _ATOMIC: a returning agent-scope read-modify-write per
 * request instead (+ 76 us per step).  _NONE: nothing -- only for a caller that owns the device's every queue (experiments, stress
 * tests).  Values are identical in every mode. */
#define DRN_KSPLIT_CONFIRM_ATOMIC 0x20000
#define DRN_KSPLIT_CONFIRM_NONE   0x80000
int64_t drn_gemm_nt_splitk_ws_elems(int M, int N, int ksplit);
int drn_gemm_nt_splitk(const DrnGemmDesc* desc /*host*/, int ksplit, float* ws, int32_t* counters, int dtype, void* stream);
/* ... and for a grouped launch: every problem's K loop is split ksplit ways; ws >= ksplit * (sum over the problems of their
 * 128x128 output tiles) * 16384 floats, one counter per tile of the launch. */
int drn_gemm_nt_splitk_grouped(const DrnGemmDesc* descs /*host*/, int ngroups, int ksplit, float* ws, int32_t* counters, int dtype,
                               void* stream);

/* ... and for ONE long-K k = 3 / stride 1 convolution with few output tiles (conv0's forward) on full-width 256x256 tiles in two
 * launches: every split writes its fp32 partial product as plane `split` of ws (drn_gemm_nt_splitk256_ws_elems floats), a second
 * kernel adds the planes in split order (+ bias), writes C and the per-128-row-slab BatchNorm statistics.  bf16 only, M and N
 * multiples of 256, Cin of 64, at least 2 K-steps of 64 per split; no gate / C2 / accumulate / out_f32 (DRN_ERR_UNSUPPORTED). */
int64_t drn_gemm_nt_splitk256_ws_elems(int M, int N, int ksplit);
int drn_gemm_nt_splitk256(const DrnGemmDesc* desc /*host*/, int ksplit, float* ws, int dtype, void* stream);

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
/* bf16, taps == 3, stride == 1, pad == 1, Lsrc == Lout and >= 4096 rows run the fused-tap kernel (one staged X block
 * feeds all three taps); everything else the per-tap kernel.  Same results to fp32 rounding (the split points differ).
 * Environment (experiments): DRN_TN_FUSED=0 disables, DRN_TN3_MINROWS, DRN_TN3_TARGET, DRN_TN3_STAGES=3|4. */
int64_t drn_wgrad_ws_elems(int M_total, int N, int Cin, int taps);

/* Deferred reduce passes.  A weight-gradient launch that splits its rows ends with a reduce launch of its own -- unless the caller hands
 * it a DrnWgradPending list (HOST memory the caller owns; all-zero = empty; nothing about it lives in the library, so two models or two
 * threads use two lists): the launch then only RECORDS its reduce, and drn_wgrad_reduce_pending() runs every recorded one in ONE launch
 * (same summation order over the splits: same bits) and empties the list.  The caller keeps the workspaces alive until then and flushes
 * before anything reads the gradients.  Not recorded (reduced at once, as with pend = NULL): an accumulating reduce (it reads `out`), and
 * anything once the list holds DRN_WGRAD_PEND_MAX items.  An output that ALREADY has a reduce recorded in the list is an error
 * (DRN_ERR_ARG before anything is launched): flush in between. */
#define DRN_WGRAD_PEND_MAX 24
typedef struct DrnWgradPendItem {
  const float* ws; /* [nsplit][N][taps][Cin] partials */
  float* out;      /* the gradient */
  int32_t nsplit, N, Cin, taps, w_layout, accumulate;
} DrnWgradPendItem;
typedef struct DrnWgradPending {
  DrnWgradPendItem it[DRN_WGRAD_PEND_MAX];
  int32_t blk_start[DRN_WGRAD_PEND_MAX + 1]; /* filled by the library when it plans the flush */
  int32_t n;                                  /* items recorded */
  float* sumsq;                               /* set for the duration of a flush */
} DrnWgradPending;

int drn_gemm_wgrad(const DrnWgradDesc* descs /*host*/, int ngroups, float* dW, int N, int Cin, int taps, int stride,
                   int pad, int w_layout, int accumulate, float* ws, int dtype, DrnWgradPending* pend /*host or NULL*/, void* stream);

/* n independent weight gradients of equal N / taps / stride / pad (different weights, different row counts: the FPN level
 * convs) in one launch; dWs is a HOST array of n device pointers; ws >= n * drn_wgrad_ws_elems(max M, N, Cin, taps).
 * Cins (host, or NULL): the problems' own input-channel counts when they differ (the FPN 1x1 laterals: 256 / 512 / 1024 -> 512);
 * Cin is then the largest of them. */
int drn_gemm_wgrad_multi(const DrnWgradDesc* problems /*host*/, int n, float* const* dWs /*host*/, int N, int Cin,
                         const int32_t* Cins /*host*/, int taps, int stride, int pad, int w_layout, int accumulate, float* ws,
                         int dtype, DrnWgradPending* pend /*host or NULL*/, void* stream);
/* The flush: one launch on `stream`; sumsq (device, drn_wgrad_pending_blocks(pend) floats, or NULL) receives one partial per workgroup of
 * the squared sums of what it wrote.  drn_wgrad_pending_bytes: partials read + gradients written by that flush (its HBM roofline
 * denominator). */
int drn_wgrad_pending_blocks(DrnWgradPending* pend);
int64_t drn_wgrad_pending_bytes(const DrnWgradPending* pend);
int drn_wgrad_reduce_pending(DrnWgradPending* pend, float* sumsq, void* stream);

/* ---- HBM-bound helpers (drn_amd/csrc/elementwise.hip) -------------------------------------------------- */
/* fp32 -> dtype cast of n contiguous elements (feature tensor / weights; the reference is fp32-only). */
int drn_cast(const float* in, void* out, int64_t n, int dtype, void* stream);
/* out[a][b][c] (dtype) = in[a*sa + b*sb + c*sc] (fp32): re-lays nn.Conv1d weights (Cout,Cin,k) as the GEMM's
 * B operands [Cout][k][Cin] (forward) and [Cin][k][Cout] (data gradient). */
/* out[k][m] = in[m][k] for a row-major M x K matrix of dtype elements (row strides ld_in / ld_out): K-major copies of the
 * operands let the largest weight gradient run as an NT product (see drn_amd/functional.py, _InputStageFn). */
/* out[m][k] = outT[k][m] = (dtype) in[m][k] for a contiguous fp32 M x K matrix: cast and K-major copy in one pass. */
int drn_cast_transpose(const float* in, void* out, void* outT, int M, int K, int dtype, void* stream);
/* The same pass with at most max_workgroups workgroups resident (a grid-stride loop over the tiles): for schedules that run it
 * BESIDE latency-bound launches of another branch (drn_amd.graph.ForkedStep: the query encoder's forward), which a full-rate
 * streaming pass next to them stretches 2-3 x. */
int drn_cast_transpose_throttled(const float* in, void* out, void* outT, int M, int K, int dtype, int max_workgroups, void* stream);
int drn_transpose2d(const void* in, int ld_in, void* out, int ld_out, int M, int K, int dtype, void* stream);
int drn_pack_weight(const float* in, void* out, int A, int B, int C, int64_t sa, int64_t sb, int64_t sc, int dtype, void* stream);
/* The same for n weights in one launch (all GEMM operands of the model after an optimizer step; the reference's cuDNN
 * re-lays its filters inside every call). */
typedef struct {
  const float* in;
  void* out;
  int64_t sa, sb, sc;
  int32_t A, B, C;
  int64_t ldo; /* elements between consecutive (a,b) rows of out; 0 = contiguous (C) */
} DrnPackDesc;
int drn_pack_weights(const DrnPackDesc* items /*host*/, int n, int dtype, void* stream);
/* Read `bytes` (16-byte aligned buffer) through the caches and discard: warms an operand that was written long ago (the
 * bf16 weight copy of prop_fc) right before the GEMM that streams it. */
int drn_touch(const void* p, int64_t bytes, void* stream);
/* n <= DRN_COPY_MAX device byte ranges copied dst[i] <- src[i] (src[i] NULL: zero-filled) in ONE launch: the refill of a captured
 * step's static input buffers between two hipGraph replays (drn_amd.trainer; replaces the reference loop's per-tensor .cuda()
 * copies, main.py:214-217, on device-resident batches).  Ranges must not overlap. */
#define DRN_COPY_MAX 8
/* rows2d (host, 3 ints per range, or NULL): {row_bytes, src_pitch, dst_pitch} > 0 makes range i two-dimensional -- bytes[i] =
 * rows * dst_pitch bytes of dst are written, the first row_bytes of every row from src rows src_pitch apart, the rest zero (a
 * token matrix padded out to the captured step's query length). */
int drn_copy_multi(const void* const* srcs /*host array of device ptrs*/, void* const* dsts, const int64_t* bytes /*host*/,
                   const int32_t* rows2d /*host or NULL*/, int n, void* stream);
/* feat[m][3] = (float)[start, end, end - start] from props_start_end (M x 2, fp64 or fp32): the position features of
 * model/main_model.py:51-55 in one launch (the reference: subtraction, torch.cat, .float()). */
int drn_pos_feat(const void* start_end, int is_f64, float* feat, int M, void* stream);
/* position_transform = nn.Linear(3,256) on [start,end,duration] (model/main_model.py:34,51-55), written straight
 * into the channel slice of conv0's input (replaces torch.cat at model/backbone.py:31-32). */
int drn_pos_embed_fwd(const float* feat /*[M][3]*/, const float* W /*[C][3]*/, const float* b, void* out, int ld_out, int M, int C,
                      int dtype, void* stream);
int drn_pos_embed_bwd(const void* dout, int ld, const float* feat, int M, int C, float* dW, float* db, int accumulate,
                      float* ws /* >= 1024*C floats */, int dtype, void* stream);
/* The same two gradients taken THROUGH the convolution that reads the embedding as the last P of its Cin input channels
 * (model/backbone.py:31-32 cat -> forward_conv0), from the gradient dY (B*Lo rows x Cout) at that conv's output, so that the conv's
 * input-gradient product can leave those P columns out:  Q[tap][j][o] = sum_rows dY[s,to,o] * f[s*L + to*stride - pad + tap][j]
 * (f[.][3] = 1, rows outside [0, L) skipped), dW[c][j] = sum_{tap,o} Wd[c][tap*Cout + o] * Q[tap][j][o], db[c] = column j = 3.
 * Wd: row Cin-P of the (Cin, k, Cout) copy of the conv weight in `dtype` (ldw elements between rows); k = 1 or 3;
 * ws >= drn_conv_tail_bwd_ws_elems(B*Lo, k, Cout) floats. */
int64_t drn_conv_tail_bwd_ws_elems(int M, int k, int Cout);
int drn_conv_tail_bwd(const void* dY, int ld_dy, int B, int Lo, int Cout, const void* Wd, int64_t ldw, int k, int stride, int pad,
                      const float* feat /*[B*L][3]*/, int L, int P, float* dW /*[P][3]*/, float* db /*[P]*/, int accumulate,
                      float* ws, int dtype, void* stream);
/* backward of F.interpolate(nearest, x2) + add (model/FPN.py:63-68): dst[s,t] += src[s,2t] + src[s,2t+1] */
int drn_pairsum_add(void* dst, int ld_dst, const void* src, int ld_src, int Mdst, int C, int accumulate, int dtype, void* stream);
/* out-of-place form: dst[s,t] = base[s,t] + src[s,2t] + src[s,2t+1] (base = the level's own incoming gradient) */
int drn_pairsum_add_to(void* dst, int ld_dst, const void* base, int ld_base, const void* src, int ld_src, int Mdst, int C,
                       int dtype, void* stream);
/* both steps of a three-level pyramid in one launch: d1 = own1 + pairs(d0) (M1 rows), d2 = own2 + pairs(d1) (M1 / 2 rows); the bits
 * of two drn_pairsum_add_to launches */
int drn_pairsum_chain3(const void* d0, int ld0, const void* own1, int ldo1, void* d1, int ld1, const void* own2, int ldo2, void* d2,
                       int ld2, int M1, int C, int dtype, void* stream);
/* out[s,t,c] = z[s,t,c] * gate[s,c]: the level-0 query gate (model/backbone.py:28-30 on prop_fc's output) as its own pass, for
 * the schedule that runs the query encoder beside the prop_fc GEMM (otherwise drn_gemm_nt's epilogue applies the gate). */
int drn_gate_fwd(const void* z, int ld_z, const float* gate, int ldg, void* out, int ld_out, int nseq, int L, int C, int dtype,
                 void* stream);
/* backward of the query gating x = q[:, :, None] * x (model/backbone.py:28-30):
 * dC = (add ? add : 0) + dG * gate[seq] (skipped when dC is NULL); dgate[seq][c] = sum_t dG*act;
 * dsum (optional, [nseq][C]) = sum_t dG * gate[seq]: per-clip column sums of dC's gated term (bias-gradient partials) */
int drn_gate_bwd(const void* dG, int ld_dg, const void* act, int ld_act, const float* gate, int ldg, const void* add, int ld_add,
                 void* dC, int ld_dc, float* dgate, int ld_dgate, float* dsum, int nseq, int L, int C, int dtype, void* stream);
/* Input-stage variant: the gated gradient is written only TRANSPOSED, dCT[c][seq*L + t] (row stride ldt), as the K-major
 * operand of the prop_fc weight gradient; dgate / dsum as above.  L % 32 == 0. */
int drn_gate_bwd_t(const void* dG, int ld_dg, const void* act, int ld_act, const float* gate, int ldg, void* dCT, int64_t ldt,
                   float* dgate, int ld_dgate, float* dsum, int nseq, int L, int C, int dtype, void* stream);
/* out[c] (+)= sum_m X[m][c]  (bias gradients) */
int drn_colsum(const void* X, int ld, int M, int C, float* out, int accumulate, float* ws /* >= 64*C floats */, int dtype,
               void* stream);

/* ---- BatchNorm1d(+ReLU) (drn_amd/csrc/bn.hip; model/basic_blocks.py:23-26, model/fcos.py:34,38,60,66) ---- */
typedef struct DrnBnGroup {
  const float* stats; /* [tiles][2][C] from drn_gemm_nt */
  int32_t tiles, M;
  float* scale_shift; /* out [2][C] */
  float* save;        /* out [2][C]: mean, invstd (for backward) */
} DrnBnGroup;
/* Train mode: batch statistics -> scale/shift; running stats updated once per group IN ORDER (shared head modules
 * are applied once per pyramid level, model/fcos.py:93-102).  conv_bias (or NULL) only shifts running_mean. */
int drn_bn_finalize(const DrnBnGroup* groups /*host*/, int ngroups, int C, const float* gamma, const float* beta,
                    const float* conv_bias, float* running_mean, float* running_var, float momentum, float eps, void* stream);
int drn_bn_eval_scale_shift(int C, const float* gamma, const float* beta, const float* conv_bias, const float* running_mean,
                            const float* running_var, float eps, float* scale_shift, void* stream);
/* The same with per-group parameters, for groups that belong to different BatchNorm modules (the FPN levels) as well as
 * groups sharing one (pass the same pointers: they are updated in order). */
typedef struct DrnBnFinDesc {
  const float* stats;
  int32_t tiles, M;
  float* scale_shift;
  float* save;
  const float* gamma;
  const float* beta;
  const float* conv_bias;  /* or NULL */
  float* running_mean;     /* or NULL */
  float* running_var;      /* or NULL */
  float momentum, eps;
} DrnBnFinDesc;
int drn_bn_finalize_multi(const DrnBnFinDesc* descs /*host*/, int n, int C, void* stream);
/* out = [relu](raw*scale+shift) [+ up[seq, t/2]] ; gated = out*gate[seq]  (fused consumers' prologues) */
int drn_bn_apply(const void* raw, int ld_raw, const float* scale_shift, void* out, int ld_out, int M, int C, int L, const void* up,
                 int ld_up, const float* gate, int ldg, void* gated, int ld_gated, int relu, int dtype, void* stream);
/* ... for up to DRN_MAX_GROUPS pyramid levels of equal C in ONE launch (independent levels only). */
typedef struct DrnBnApplyDesc {
  const void* raw;
  const float* scale_shift;
  void* out;
  const void* up;     /* or NULL */
  const float* gate;  /* or NULL (with gated) */
  void* gated;
  int32_t ld_raw, ld_out, ld_up, ldg, ld_gated, M, L;
} DrnBnApplyDesc;
int drn_bn_apply_multi(const DrnBnApplyDesc* descs /*host*/, int n, int C, int relu, int dtype, void* stream);
/* Train-mode forward in ONE launch (C % 64 == 0): statistics merge + running-statistics update + apply.  Each workgroup of the
 * apply pass merges the slab statistics of its own 64 channels; scale_shift / save / the running statistics are written once
 * per channel, groups in order (groups may share a BatchNorm module: model/fcos.py:93-102).  Replaces drn_bn_finalize(_multi)
 * + drn_bn_apply_multi for nn.BatchNorm1d + ReLU in training (model/basic_blocks.py:23-26). */
typedef struct DrnBnTrainDesc {
  const float* stats; /* [tiles][2][C] from drn_gemm_nt, tiles = ceil(M/128) */
  float* scale_shift; /* out [2][C] */
  float* save;        /* out [2][C]: mean, invstd */
  const float* gamma;
  const float* beta;
  const float* conv_bias; /* or NULL */
  float* running_mean;    /* or NULL */
  float* running_var;     /* or NULL */
  const void* raw;
  void* out;
  const void* up;    /* or NULL */
  const float* gate; /* or NULL (with gated) */
  void* gated;
  float momentum, eps;
  int32_t tiles, ld_raw, ld_out, ld_up, ldg, ld_gated, M, L;
} DrnBnTrainDesc;
int drn_bn_train_apply(const DrnBnTrainDesc* descs /*host*/, int n, int C, int relu, int dtype, void* stream);
/* Conv1d -> BatchNorm1d (training) -> ReLU in ONE launch (model/basic_blocks.py:9-31; FPN laterals / output convs
 * model/FPN.py:54-69; head towers model/fcos.py:33-69 with statistics per level call, fcos.py:93-102): drn_gemm_nt +
 * drn_bn_train_apply without the second launch and without re-reading the raw conv output.  gemm[i] / bn[i] describe group i
 * (bn[i].raw == gemm[i].C; the raw output is still written: backward reads it; gemm[i].stats / bn[i].stats are ignored).
 * Every workgroup keeps its raw tile in registers, publishes its per-slab statistics as 64-bit {value, generation} pairs in
 * tagged_ws, merges the pairs of its tile column -- polling them until they carry this launch's generation -- exactly as
 * drn_bn_train_apply merges (same bits), and stores the normalised tile.  up_group (host, or NULL): up_group[i] = j > i makes
 * out_i += nearest_x2(out_j) (the FPN top-down chain), recomputed from group j's raw rows inside the launch; -1 = none.
 * tagged_ws: >= drn_conv_bn_train_ws_bytes() bytes, 8-byte aligned, ZERO when first used and afterwards written by this entry
 * point only; generation: one int32, zero at first, advanced by the kernel (one pair of buffers per stream).
 * Returns DRN_ERR_UNSUPPORTED -- nothing launched, call the two-launch path -- when the groups' N differ or are not a
 * multiple of the tile width, an epilogue option of drn_gemm_nt is requested, or the grid exceeds what the chip holds at once
 * (the wait needs every workgroup resident; the device must not be shared with another process that also waits).  */
int64_t drn_conv_bn_train_ws_bytes(const DrnGemmDesc* gemm /*host*/, int ngroups);
int drn_conv_bn_train(const DrnGemmDesc* gemm /*host*/, const DrnBnTrainDesc* bn /*host*/, int ngroups, int relu,
                      const int32_t* up_group /*host or NULL*/, void* tagged_ws, int64_t ws_bytes, int32_t* generation, int dtype,
                      void* stream);
/* Watchdog of that wait: workgroups that gave up after 2 s since the last reset (their launch's results are invalid).
 * Synchronises the device; -1 on error. */
int drn_conv_bn_train_timeouts(int reset);
/* dRaw, dgamma, dbeta from dOut; ReLU mask recomputed from raw; draw may alias dout. */
int drn_bn_bwd(const void* dout, int ld_dout, const void* raw, int ld_raw, const float* scale_shift, const float* save,
               const float* gamma, void* draw, int ld_draw, float* dgamma, float* dbeta, int accumulate, int M, int C, int relu,
               float* ws /* >= 515*C floats */, int dtype, void* stream);
/* ... for up to DRN_MAX_GROUPS levels of equal C in three launches; levels sharing one module pass the same dgamma/dbeta
 * with accumulate = 1 from the second level on (summed in level order).  ws >= n*515*C floats. */
typedef struct DrnBnBwdDesc {
  const void* dout;
  const void* raw;
  const float* scale_shift;
  const float* save;
  const float* gamma;
  void* draw;
  float* dgamma;
  float* dbeta;
  int32_t ld_dout, ld_raw, ld_draw, accumulate, M;
  /* drn_bn_bwd_one only.  gb_dg != NULL: the layer's output was also consumed GATED by the query (model/backbone.py:28-30,
   * gated = out * gate[clip]); gb_dg [M][gb_ld_dg] is the gradient of the gated output, dout (or NULL) the gradient of the plain one.
   * The launch does drn_gate_bwd's work on the rows it loads: dout_eff = dtype(dout + gb_dg * gb_gate[m / gb_L]) feeds the BatchNorm
   * backward (bit-identical to the two launches), gb_dgate[clip][c] ([M / gb_L][C] fp32) = sum_t gb_dg * out with `out` recomputed from
   * raw.  Needs gb_L a multiple of 32 rows (bf16; 16 in fp32) that divides the launch's row block: drn_bn_bwd_one_ws_bytes() says 0
   * otherwise, and the caller runs drn_gate_bwd itself. */
  const void* gb_dg;
  const float* gb_gate;
  float* gb_dgate;
  int32_t gb_ld_dg, gb_ldg, gb_L;
} DrnBnBwdDesc;
int drn_bn_bwd_multi(const DrnBnBwdDesc* descs /*host*/, int n, int C, int relu, float* ws, int dtype, void* stream);
/* The same in ONE launch: every workgroup keeps its rows of dout and raw in registers between the sums and the apply half and
 * meets the other row blocks of its channel tile through tagged pairs in `tagged_ws` (64-byte aligned, >=
 * drn_bn_bwd_one_ws_bytes() bytes, ZERO before its first use and then left to the library: it carries the launch generation;
 * one launch at a time per workspace).  Needs C % 64 == 0 and a grid the chip holds at once: DRN_ERR_UNSUPPORTED (nothing
 * launched) / 0 bytes otherwise -- the caller then takes drn_bn_bwd_multi.  Per-row-block partial sums are formed over
 * different row blocks than drn_bn_bwd_multi's, so the two agree to rounding, not bit for bit.
 * drn_bn_bwd_one_timeouts: workgroups that gave up waiting (2 s watchdog; 0 in a healthy run); synchronises the device. */
int64_t drn_bn_bwd_one_ws_bytes(const DrnBnBwdDesc* descs /*host*/, int n, int C, int dtype);
int drn_bn_bwd_one(const DrnBnBwdDesc* descs /*host*/, int n, int C, int relu, void* tagged_ws, int64_t ws_bytes, int dtype, void* stream);
int drn_bn_bwd_one_timeouts(int reset);

/* ---- 1-2 channel output heads (drn_amd/csrc/heads.hip; model/fcos.py:43-49,68,96-102) ------------------- */
typedef struct DrnHeadGroup {
  const void* X; /* level activations, channels-last (may be a column slice) */
  void* dX;      /* backward: gradient buffer, same geometry */
  int32_t ldx, M, L;
  const float* scale; /* exp mode: scales[l].scale (1 float) */
} DrnHeadGroup;
/* out[r][n] = bias[n] + conv(X, W)[r][n]; exp_mode: z = that, out = exp(scale_l*z).  `W` is the nn.Conv1d weight (N,C,taps)
 * RE-LAID as [N][taps][C] fp32 (what drn_pack_weight(perm 0,2,1) / drn_adam_tiled's kind-1 copy produce); dW comes back in the
 * parameter's own (N,C,taps) layout. */
int drn_head_out_fwd(const DrnHeadGroup* groups /*host*/, int ngroups, const float* W, const float* bias, int N, int C, int taps,
                     int exp_mode, float* out, float* z, int dtype, void* stream);
/* dout is the gradient w.r.t. `out`; exp_mode applies d out/d z = scale*out and accumulates dscale[l];
 * dX (+)= conv^T(dz), dW/dbias/dscale (+)= ... over all levels. */
int drn_head_out_bwd(const DrnHeadGroup* groups /*host*/, int ngroups, const float* W, const float* dout, const float* out,
                     const float* z, int N, int C, int taps, int exp_mode, int accumulate_dx, float* dW, float* dbias,
                     float* dscale, int accumulate_dw, float* ws /* >= drn_heads_ws_elems(sum of M, N, C, taps) floats */, int dtype, void* stream);

/* Up to 2 heads per launch (cls_logits + bbox_pred read the two halves of one tower output: side by side they fill the chip).
 * One DrnHeadCall = the arguments of drn_head_out_fwd / _bwd for one head; forward uses groups, W, bias, out, z; backward
 * groups (with dX), W, dout, out, z, dW, dbias, dscale (+ dscale_stride), ws (>= drn_heads_ws_elems() floats each), accumulate flags.
 * Backward is two launches: data gradient + weight-gradient partials side by side, then the reduction of the partials. */
typedef struct DrnHeadCall {
  const DrnHeadGroup* groups; /* host */
  int32_t ngroups, N, C, taps, exp_mode, accumulate_dx, accumulate_dw;
  int32_t dscale_stride; /* elements between dscale[l] and dscale[l+1] (0 = 1) */
  const float* W;
  const float* bias;
  float* out;
  float* z;
  const float* dout;
  float* dW;
  float* dbias;
  float* dscale;
  float* ws;
} DrnHeadCall;
int64_t drn_heads_ws_elems(int total_rows, int N, int C, int taps); /* fp32 workspace of one head's backward */
int drn_heads_fwd(const DrnHeadCall* calls /*host*/, int ncalls, int dtype, void* stream);
int drn_heads_bwd(const DrnHeadCall* calls /*host*/, int ncalls, int dtype, void* stream);

/* ---- losses (drn_amd/csrc/loss.hip; model/loss.py:40-239, model/layers/{iou_loss,sigmoid_focal_loss}.py) -- */
typedef struct DrnLossLevel {
  int32_t L;    /* locations per clip on this level */
  float stride; /* fpn_stride: location = t*stride + stride/2 (model/fcos.py:204-211) */
  float lo, hi; /* object_sizes_of_interest (model/loss.py:47-51) */
} DrnLossLevel;
/* logits [R], reg [R][2], iou [R] are fp32 over R = B*sum(L) rows ordered level-first, clip-major (the reference's
 * flatten order, model/loss.py:150-166).  gt [B][2] is fp32, or fp64 with gt_f64 = 1 (cast on load, main_model.py:74).
 * out6 = {loss_cls, loss_reg, loss_iou, n_pos, n_iou_pos, loss_cls + loss_reg + loss_iou (main.py:225)}.
 * bumps (host array, may be NULL): int64 device counters (BatchNorm num_batches_tracked) incremented by the same launch.
 * Replaces FCOSLossComputation.__call__ incl. fcos_core._C.sigmoid_focalloss_forward/backward + IOULoss +
 * segment_tiou/SmoothL1. */
#define DRN_LOSS_MAX_BUMPS 32
typedef struct DrnCounterBump {
  void* counter; /* int64 on the device */
  int32_t inc;
} DrnCounterBump;
int drn_fcos_loss_fwd(const DrnLossLevel* levels /*host*/, int nlevels, int B, const float* logits, const float* reg,
                      const float* iou, const void* gt /*[B][2]*/, int gt_f64, float gamma, float alpha, float target_scale,
                      int iou_stage, float* out6, float* labels /*[R] or NULL*/, float* ws /* >= 5*ceil(R/256) floats */,
                      int32_t* ticket /* one zeroed int32 (left zero), or NULL: the final summation as a second launch */,
                      const DrnCounterBump* bumps /*host*/, int nbumps, void* stream);
int drn_fcos_loss_bwd(const DrnLossLevel* levels /*host*/, int nlevels, int B, const float* logits, const float* reg,
                      const float* iou, const void* gt, int gt_f64, float gamma, float alpha, float target_scale, int iou_stage,
                      const float* fwd_out6, const float* g_cls, const float* g_reg, const float* g_iou /* 1 float each, NULL = 0 */,
                      float* dlogits, float* dreg, float* diou, void* stream);

/* The reference's ONE real FFI on this path, 1:1: fcos_core._C.sigmoid_focalloss_forward(logits, targets, num_classes, gamma,
 * alpha) -> losses and sigmoid_focalloss_backward(logits, targets, d_losses, num_classes, gamma, alpha) -> d_logits
 * (model/layers/sigmoid_focal_loss.py:18-20,31-33).  logits / losses / d_losses / d_logits are fp32 [N][num_classes],
 * targets int32 [N]: 0 = background, c in 1..num_classes = foreground class c, negative = ignored (loss and gradient 0).
 * Element-wise (the caller sums, sigmoid_focal_loss.py:68); any num_classes; finite for any finite logit. */
int drn_focal_fwd(const float* logits, const int32_t* targets, int64_t N, int num_classes, float gamma, float alpha, float* losses,
                  void* stream);
int drn_focal_bwd(const float* logits, const int32_t* targets, const float* d_losses, int64_t N, int num_classes, float gamma,
                  float alpha, float* d_logits, void* stream);

/* Stand-alone IOULoss (model/layers/iou_loss.py:5-24, exported by model/layers/__init__.py): pred / target fp32 [N][2] =
 * (left, right) distances, weight fp32 [N] or NULL.  out2[0] = sum(l_i w_i) / sum(w) when weight != NULL and sum(w) > 0, else
 * mean(l_i), l_i = -log((min(pr,tr) + min(pl,tl) + 1e-8) / (union + 1e-8)); out2[1] = the divisor, negative for the plain mean
 * (backward reads the mode there: no host sync).  N == 0 is an error (the reference asserts).  Backward: gout = upstream
 * gradient (1 float, NULL = 1); dpred / dtarget [N][2], either may be NULL; ties in min() split evenly. */
int drn_iou_loss_fwd(const float* pred, const float* target, const float* weight, int64_t N, float* out2, void* stream);
int drn_iou_loss_bwd(const float* pred, const float* target, const float* weight, int64_t N, const float* out2, const float* gout,
                     float* dpred, float* dtarget, void* stream);

/* ---- eval post-processor (drn_amd/csrc/postproc.hip; model/inference.py:51-120,166-199) ------------------------
 * Per clip and level: candidates sigmoid(logit) > thr, score = sigmoid(logit)[*sigmoid(iou)] (iou NULL in the first stage),
 * top_n per level, segments ((loc-reg0)/downsample, (loc+reg1)/downsample) clamped to [0,1], score = sqrt(.), loc/32.
 * Inputs are the head outputs in the loss layout (rows level-first / clip-major).  Outputs are padded per clip:
 * det [B][sum L][2], scores / locs [B][sum L], counts [B][nlevels] = kept candidates per level, written level after level. */
int drn_postprocess(const DrnLossLevel* levels /*host*/, int nlevels, int B, const float* logits, const float* reg, const float* iou,
                    float thr, int top_n, float downsample, float* det, float* scores, float* locs, int32_t* counts, void* stream);
/* Recall@k with temporal NMS on the device, straight from drn_postprocess's outputs (utils/evaluate_utils.py:131-215: stable
 * sort by score, greedy NMS at IoU threshold iou - 0.05 visiting score ties from the later prediction, hit = one of the first
 * k survivors overlaps gt by >= iou, un-clamped IoU; double arithmetic on the float32 detections, as the host path).
 * first_hit[b][q] = 0-based position among the NMS survivors of the first one that hits at ious[q], or max_topk when none of the
 * first max_topk does: recall@k counts first_hit < k.  gt: (B, 2) fp64 or fp32; ious: device array of n_iou doubles. */
int drn_eval_recall(const float* det, const float* scores, const int32_t* counts, int B, int nlevels, int rows_per_clip,
                    const void* gt, int gt_is_f64, const double* ious /*device*/, int n_iou, int max_topk, int32_t* first_hit,
                    void* stream);

/* ---- query-encoder glue (drn_amd/csrc/qenc.hip; model/language_module.py:17-63), all fp32 ----------------------
 * Word embedding lookup written time-major (L, B, E) and its dense gradient (row padding_idx stays zero). */
int drn_qe_embed_fwd(const int64_t* tokens /*[B][L]*/, const float* table /*[V][E]*/, float* out_tm, int B, int L, int E, void* stream);
int drn_qe_embed_bwd(const int64_t* tokens, const float* demb_tm, float* dtable /*[V][E], fully written*/, int B, int L, int E, int V,
                     int padding_idx, void* stream);
/* q_vector = [output[b][0] ; output[b][len_b-1]] (language_module.py:48-54); bwd ADDS into dout (B, L, C). */
int drn_qe_qvec_fwd(const float* out /*[B][L][C]*/, const int64_t* lengths, float* qvec /*[B][2C]*/, int B, int L, int C, void* stream);
int drn_qe_qvec_bwd(const float* dqvec, const int64_t* lengths, float* dout, int B, int L, int C, void* stream);
/* The three attention "commands" (language_module.py:17-36): logits = cmd_inter2logits(q_cmd[:,None,:] * output),
 * softmax over the words of each query (padding masked), cmd = att @ output.  qcmd (B,3,C), att (B,3,L), cmds (3,B,C).
 * bwd: dcmd0..2 (B,C) each or NULL; writes dqcmd (B,3,C), dout (B,L,C) and per-clip partials dw_part (B,C), dbias_part (B)
 * of the cmd_inter2logits weight / bias gradients (sum them over B). */
int drn_qe_attn_fwd(const float* out, const float* qcmd, const float* w /*[C]*/, const float* bias /*[1]*/, const int64_t* lengths,
                    float* att, float* cmds, int B, int L, int C, void* stream);
int drn_qe_attn_bwd(const float* dcmd0, const float* dcmd1, const float* dcmd2, const float* att, const float* out, const float* qcmd,
                    const float* w, const int64_t* lengths, float* dqcmd, float* dout, float* dw_part, float* dbias_part, int B, int L,
                    int C, void* stream);
/* dst_s[j] = sum_m X[m][col0_s + j] for up to DRN_COLSEG_MAX column ranges of one fp32 matrix (bias gradients). */
#define DRN_COLSEG_MAX 8
typedef struct {
  float* dst;
  int32_t col0, n;
} DrnColSeg;
int drn_colsum_segs(const float* X, int ld, int M, const DrnColSeg* segs /*host*/, int nsegs, void* stream);

/* ---- batch-sized dense layers of the query side (drn_amd/csrc/qdense.hip; model/language_module.py:13-23,38-63,
 * model/main_model.py:36-50), fp32, exact-fp32 MFMA, GROUPED: one launch serves up to DRN_QD_MAX problems.
 * drn_skinny_group:  Y[M][N] = X[M][K] * W[N][K]^T (+ bias[N]) (ReLU) (zeroed where mask[m][n] <= 0), M <= 64 rows, any N,
 *   K % 4 == 0: qInput* / the gate projections / the LSTM input projection forward, and -- through cached transposed
 *   weight copies -- their input gradients (mask = the ReLU of the layer in front).  Long K is split over workgroups; the
 *   last-arriving workgroup of a column tile adds the partial tiles in a fixed order (deterministic, no second launch):
 *   ws >= drn_skinny_group_ws_elems() floats, counters = DRN_QD_COUNTERS int32 zeros owned by the caller (the kernel
 *   leaves them zero).
 * drn_outer_wgrad:  dW[N][K] = sum_m dY[m][N] * X[m][K] (row stride ldw), db[N] = db2[N] = sum_m dY[m][N] (optional):
 *   the weight / bias gradients of the same layers (M = clips, or clips x words for the LSTM weights).  A problem with
 *   dW == NULL only produces the column sums.  dtype DRN_F32: exact-fp32 MFMA; DRN_BF16 (the bf16 model): the fp32 operands are
 *   rounded to bf16 as they enter the MFMA (fp32 accumulation, fp32 results; the column sums stay exact). */
#define DRN_QD_MAX 16
#define DRN_QD_COUNTERS 2048
typedef struct DrnSkinnyDesc {
  const void* X;     /* fp32 rows, or bf16 rows when x_dtype == DRN_BF16 (K % 8 == 0, ldx % 8 == 0 elements): the weights are then
                        rounded to bf16 on load and the product accumulates in fp32 on the bf16 MFMA */
  const float* W;
  const float* bias; /* or NULL */
  const float* mask; /* or NULL; [M][ldm] */
  float* Y;
  int32_t ldx, ldy, ldm, M, N, K, relu;
  int32_t x_dtype;   /* DRN_F32 (0) or DRN_BF16; the same for every problem of a launch */
} DrnSkinnyDesc;
int64_t drn_skinny_group_ws_elems(const DrnSkinnyDesc* descs /*host*/, int n);
int drn_skinny_group(const DrnSkinnyDesc* descs /*host*/, int n, float* ws, int32_t* counters, void* stream);
typedef struct DrnOuterDesc {
  const float* dY;
  const float* X;
  float* dW;
  float* db;  /* or NULL */
  float* db2; /* or NULL */
  int32_t ldy, ldx, ldw, M, N, K;
} DrnOuterDesc;
int drn_outer_wgrad(const DrnOuterDesc* descs /*host*/, int n, int dtype, void* stream);

/* ---- language-guided pooling (drn_amd/csrc/lgp.hip; model/LGP.py:29-51) ---------------------------------------
 * x (B, t, C) channels-last, qn (B, C) fp32 = BN(conv1x1(query)) prepared by the caller, out (B, t/2, C),
 * att (B, t/2, 2) fp32 (saved for backward).  C <= 2048 (bf16) / 1024 (f32). */
int drn_lgp_fwd(const void* x, int ldx, const float* qn, void* out, int ld_out, float* att, int B, int t, int C, int dtype,
                void* stream);
/* dx (B, t, C) and dqn (B, C) = gradient w.r.t. qn; ws >= B*ceil(t/8)*C floats. */
int drn_lgp_bwd(const void* x, int ldx, const float* qn, const float* att, const void* dout, int ld_dout, void* dx, int ld_dx,
                float* dqn, float* ws, int B, int t, int C, int dtype, void* stream);

/* ---- query-encoder BiLSTM recurrence (drn_amd/csrc/lstm.hip; model/language_module.py:13-15,38-45) ------------
 * One launch per time step for both directions; sequence lengths (int64) on the device (replaces pack_padded_sequence +
 * the cuDNN/MIOpen RNN).  fp32, gate order i,f,g,o.  xproj / gates / dgates are [L][B][2][4H] indexed by time and
 * direction (one (L*B) x 8H matrix); xproj = x_t W_ih^T without biases (b_ih + b_hh are added here); hseq/cseq
 * [2][L+1][B][H] by step (slot 0 is never read: zero initial state); hprev_t [L][B][2][H] = hidden state that entered
 * time t (operand of the W_hh gradient); out [B][L][2H].  Step s handles t = s (forward direction) and t = L-1-s
 * (reverse).  B <= 64, H % 64 == 0.  w_dtype: DRN_F32 (exact-fp32 MFMA, the parity mode) or DRN_BF16 (Whh_* / WhhT_* are bf16
 * copies in the same element order, H % 128 == 0: the hidden state / gate gradients are rounded to bf16 as they are loaded,
 * products accumulate in fp32; states, gates and every output stay fp32).  qvec (optional, [B][4H]): the sentence vector [out[b][0][:] ; out[b][len_b-1][:]]
 * (language_module.py:48-54), each half-row written by the step that produces it -- no gather launch after the recurrence.
 * hseq16 (optional, fp16 [2][L+1][B][H], with fp32 weights and H % 128 == 0): every step also writes its state there and reads the
 * previous one from there; the recurrent product then runs on the fp16 MFMA (W_hh rounded in registers, fp32 accumulation). */
int drn_lstm_step_fwd(const float* xproj, const void* Whh_f, const void* Whh_r, int w_dtype, const float* b_ih_f, const float* b_hh_f,
                      const float* b_ih_r, const float* b_hh_r, float* hseq, float* cseq, float* gates, float* out, float* hprev_t,
                      float* qvec, void* hseq16, const int64_t* lengths, int B, int L, int H, int s, void* stream);
/* The hseq16 forward with ALL steps in ONE launch (the bits of L drn_lstm_step_fwd launches): workgroups stay for the whole sequence and
 * hand the hidden state over through xch_ws -- 64-byte aligned, >= drn_lstm_seq_fwd_ws_bytes(B, L, H) bytes, ZERO before its first use and
 * then left to the library (one buffer per (B, L, H); launches on it are stream-ordered).  H % 128 == 0, fp32 W_hh.  DRN_ERR_UNSUPPORTED
 * (nothing launched) when the grid does not fit the chip at once.  drn_lstm_seq_fwd_timeouts: workgroups that gave up waiting (2 s
 * watchdog; 0 in a healthy run); synchronises the device. */
int64_t drn_lstm_seq_fwd_ws_bytes(int B, int L, int H);
int drn_lstm_seq_fwd(const float* xproj, const void* Whh_f, const void* Whh_r, const float* b_ih_f, const float* b_hh_f, const float* b_ih_r,
                     const float* b_hh_r, float* hseq, float* cseq, float* gates, float* out, float* hprev_t, float* qvec, void* xch_ws,
                     int64_t ws_bytes, const int64_t* lengths, int B, int L, int H, void* stream);
int drn_lstm_seq_fwd_timeouts(int reset);
/* Backward: drn_lstm_bwd_first does the cell backward of the last step (s = L-1) from dout [B][L][2H] alone; then
 * drn_lstm_step_bwd for s = L-1 .. 1 propagates through W_hh of step s (dgates[t(s)] x Whh) and applies the cell backward
 * of step s-1 in its epilogue (the recurrent dL/dh never goes to memory).  dgates is the operand of the weight-gradient
 * GEMMs; dc / dh_pass are [2][B][H] running state.  WhhT_* = Whh^T as [H][4H].  dqvec (optional, [B][4H]): gradient of the
 * sentence vector, added to dout rows 0 and len_b-1 as they are read (dout itself is not modified).  dgates16 (optional, with
 * w_dtype DRN_BF16): a bf16 buffer shaped like dgates; the cell backward writes a rounded copy of every gate gradient there and the
 * recurrent product reads that copy (half the bytes) -- dgates itself stays fp32 for the weight / input-gradient products. */
int drn_lstm_bwd_first(const float* dout, const float* gates, const float* cseq, float* dgates, float* dc, float* dh_pass,
                       const float* dqvec, void* dgates16, const int64_t* lengths, int B, int L, int H, void* stream);
int drn_lstm_step_bwd(const float* dout, const float* gates, const float* cseq, const void* WhhT_f, const void* WhhT_r, int w_dtype,
                      float* dgates, float* dc, float* dh_pass, const float* dqvec, void* dgates16, const int64_t* lengths, int B, int L,
                      int H, int s, void* stream);

/* ---- fused clip_grad_norm_ + Adam over flat gradient buckets (drn_amd/csrc/optim.hip; main.py:140,238-243) ---- */
int64_t drn_opt_nblocks(int64_t n); /* partial sums produced by drn_sumsq_partials for n elements */
/* partials[b] = sum of g^2 over block b; step_counter (device int, or NULL) is incremented once per call. */
int drn_sumsq_partials(const float* g, int64_t n, float* partials, int* step_counter, void* stream);
/* total_sumsq[0] = grad_scale^2 * sum of ALL buckets' partials, one workgroup, fixed order: the squared global norm of the
 * gradients the Adam kernels see, grad_scale * g (grad_scale = 1/world when the buckets hold the all-reduced SUM; 1 otherwise). */
int drn_sumsq_finalize(const float* partials, int npartials, float* total_sumsq, float grad_scale, void* stream);
/* The norm pass WITHOUT the gradients whose squared sums the producing kernels left behind (one process only: after an all-reduce
 * the producers' sums are stale).  drn_sumsq_partials_skip: elements in the nskip <= DRN_SUMSQ_MAX_SKIP ranges [skip_lo[i],
 * skip_hi[i]) (host arrays, element offsets into g) are not read.  drn_sumsq_finalize2 adds next <= DRN_SUMSQ_MAX_EXT external
 * partial arrays (host arrays of device pointers / lengths: DrnGemmDesc::sumsq of the prop_fc weight gradient,
 * drn_wgrad_reduce_pending's sumsq) to the pass's own partials, everything in a fixed order. */
#define DRN_SUMSQ_MAX_SKIP 16
#define DRN_SUMSQ_MAX_EXT 8
int drn_sumsq_partials_skip(const float* g, int64_t n, float* partials, int* step_counter, const int64_t* skip_lo /*host*/,
                            const int64_t* skip_hi /*host*/, int nskip, const unsigned char* block_classes /*device or NULL*/, void* stream);
/* Host helper: block_classes[b] for the drn_opt_nblocks(n) blocks of that pass (0 = clear of every range, 1 = inside one, 2 = on a
 * boundary); uploaded once per range set, it spares every block the walk over the ranges. */
int drn_sumsq_block_classes(int64_t n, const int64_t* skip_lo /*host*/, const int64_t* skip_hi /*host*/, int nskip, unsigned char* classes /*host*/);
int drn_sumsq_finalize2(const float* partials, int npartials, const float* const* ext /*host*/, const int32_t* ext_n /*host*/, int next,
                        float* total_sumsq, float grad_scale, void* stream);
/* One bucket: g/m/v flat [n]; tensor i covers [seg_start[i], seg_start[i+1]) and lives at p_ptr[i] (both tables on the
 * device).  blk_seg (device, drn_opt_nblocks(n) ints, or NULL): index of the tensor holding element 4096*b, so a block
 * does not search the table.  mirror_dev (device table of nseg pointers, or NULL; entries may be NULL): a bf16 copy of tensor
 * i in the same element order (the GEMM operand of a Linear / 1x1 conv), rewritten from the updated value in the same pass.
 * Every gradient is read as grad_scale * g; clip coef = min(1, max_norm/(sqrt(total_sumsq)+1e-6)) (max_norm<=0: off). */
int drn_adam_bucket(const float* g, float* m, float* v, int64_t n, const int64_t* seg_start_dev, float* const* p_ptr_dev, int nseg,
                    const int32_t* blk_seg, void* const* mirror_dev, const float* total_sumsq, const int* step_counter, float lr,
                    float beta1, float beta2, float eps, float max_norm, float grad_scale, void* stream);

/* The same update for tensors whose GEMM operands are RE-LAID copies (conv weights (Cout, Cin, k) as [Cout][k][Cin] and
 * [Cin][k][Cout]; Linear weights transposed): the tensor is walked in 64 x 64-channel tiles instead of linearly, each tile's
 * updated values pass through LDS and leave as 16-byte pieces of BOTH copies -- the per-step drn_pack_weights launches over
 * these tensors (76 us per step) disappear.  items: DEVICE array; blk_item / blk_tile: device, one int per workgroup (which
 * item, which tile of it), nblocks of them.  The caller leaves these tensors out of drn_adam_bucket (NULL in p_ptr_dev). */
typedef struct DrnAdamTiledItem {
  float* p;        /* parameter, R x (C*k) row-major fp32 */
  int64_t off;     /* its offset inside the flat g / m / v buffers */
  void* m1;        /* copy [r][tap][c]: element (r, c, tap) at m1[(r*k + tap)*ld1 + c], or NULL */
  void* m2;        /* copy [c][tap][r]: element (r, c, tap) at m2[(c*k + tap)*ld2 + r], or NULL */
  int64_t ld1, ld2;
  int32_t R, C, k;
  int32_t code1, code2; /* dtype of m1 / m2: DRN_F32 or DRN_BF16 */
  int32_t tiles_c;      /* ceil(C / 64) */
} DrnAdamTiledItem;
int drn_adam_tiled(const float* g, float* m, float* v, const DrnAdamTiledItem* items_dev, const int32_t* blk_item_dev,
                   const int32_t* blk_tile_dev, int nblocks, const float* total_sumsq, const int* step_counter, float lr, float beta1,
                   float beta2, float eps, float max_norm, float grad_scale, void* stream);

/* MEASUREMENT infrastructure (bench.py; nothing on the product path calls it): what this chip sustains on bf16 MFMA with operands that
 * toggle like data.  MI355X clocks to its power budget: a register-only v_mfma_f32_32x32x16_bf16 loop at the issue floor (32 cycles per
 * MFMA and SIMD, 256 workgroups x 4 waves, no LDS or memory traffic) runs at ~2.3 GHz on zeros and ~1.7-1.8 GHz on random bf16 operands.
 * Launches on `stream`, waits for it, returns TFLOP/s, the effective clock (shader cycles / wall time) and cycles per MFMA.
 * ws: drn_diag_mfma_ws_bytes() bytes of device memory.  iters: 8 MFMAs per wave each (20000 ~ 3 ms). */
long drn_diag_mfma_ws_bytes(void);
int drn_diag_mfma_sustained(void* ws, int iters, int zero_operands, double* tflops, double* clock_ghz, double* cycles_per_mfma, void* stream);

#ifdef __cplusplus
}
#endif
#endif
