// Bidirectional single-layer LSTM recurrence of the DRN query encoder (model/language_module.py:13-15,38-45:
// nn.LSTM(300, 512, bidirectional, batch_first) on packed sequences), one launch per time step for BOTH directions,
// with sequence lengths on the device (no packing, no host sync -> hipGraph-capturable, unlike the MIOpen RNN path).
//
// The input projection x_t W_ih^T + b_ih + b_hh for all t is a plain GEMM done by the caller; these kernels do the
// recurrent part, which is latency-bound: per step and direction a (B=32..64) x 2048 x 512 product plus the cell
// update.  It runs on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) straight from L2-resident operands: a workgroup
// owns 16 hidden units (all four gates), its 4 waves split K, partial tiles meet in LDS and the cell update is fused
// into the epilogue.  Padded positions (t >= len[b]) keep the state and emit zeros, which reproduces
// pack_padded_sequence / pad_packed_sequence semantics for both directions.
//
// Layouts (fp32): xproj / gates / dgates [L][B][2][4H] indexed by TIME t and direction (gate order i,f,g,o as in
// PyTorch) -- one (L*B) x 8H matrix, so the input projection and its weight / input gradients are single GEMMs over
// both directions; Whh [2] pointers to [4H][H]; hseq / cseq [2][L+1][B][H] by STEP (slot s+1 = state after step s; slot 0
// is never read, the initial state is zero); hprev_t [L][B][2][H] = hidden state that entered time t (operand of the
// W_hh gradient); out [B][L][2H].  Step s handles t = s for the forward direction and t = L-1-s for the reverse one.
// xproj holds x_t W_ih^T only: both bias vectors are added here.
#include "common.h"
#include "../../include/drn_hip.h"

#define LSTM_THREADS 256
#define MAX_BT 4   // batch tiles of 16 -> B <= 64

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

struct LstmFwdArgs {
  const float* xproj;
  const float* Whh[2];
  float* hseq;
  float* cseq;
  float* gates;
  float* out;
  float* hprev_t;
  const float* b_ih[2];
  const float* b_hh[2];
  const long long* lengths;   // [B] int64 (the dtype the data layer hands over)
  int B, L, H, s;
};

// grid (H/16, 2 dirs); block 256 = 4 waves, wave w reduces k in [w*H/4, (w+1)*H/4)
template <int NBT>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_step_fwd_kernel(const LstmFwdArgs A) {
  __shared__ float red[4][4 * NBT][64][4];   // [wave][gate*NBT + bt][lane][reg]
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int dir = blockIdx.y, j0 = blockIdx.x * 16;
  const int B = A.B, L = A.L, H = A.H, s = A.s;
  const int t = dir == 0 ? s : L - 1 - s;
  const float* hprev = A.hseq + ((long)(dir * (L + 1) + s) * B) * H;
  const float* W = A.Whh[dir];
  f32x4 acc[4][NBT];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) acc[g][bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int kq = H / 4;
  const int row = l & 15, kc = (l >> 4) * 4;
  for (int k0 = w * kq; k0 < (w + 1) * kq; k0 += 16) {
    f32x4 a[NBT], b[4];
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) {
      const int bb = bt * 16 + row;
      a[bt] = (bb < B && s > 0) ? *(const f32x4*)(hprev + (long)bb * H + k0 + kc) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) b[g] = *(const f32x4*)(W + (long)(g * H + j0 + row) * H + k0 + kc);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt)
          acc[g][bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[bt][e], b[g][e], acc[g][bt], 0, 0, 0);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][g * NBT + bt][l][r] = acc[g][bt][r];
  __syncthreads();
  // epilogue: wave w finishes batch tiles bt = w, w+4, ... ; D layout: b = bt*16 + (l>>4)*4 + r, j = j0 + (l&15)
  for (int bt = w; bt < NBT; bt += 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bb = bt * 16 + (l >> 4) * 4 + r;
      if (bb >= B) continue;
      const int j = j0 + (l & 15);
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = red[0][g * NBT + bt][l][r] + red[1][g * NBT + bt][l][r] + red[2][g * NBT + bt][l][r] + red[3][g * NBT + bt][l][r];
        pre[g] = v + A.xproj[(((long)t * B + bb) * 2 + dir) * 4 * H + g * H + j] + A.b_ih[dir][g * H + j] + A.b_hh[dir][g * H + j];
      }
      const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
      const long st_prev = ((long)(dir * (L + 1) + s) * B + bb) * H + j;
      const long st_new = ((long)(dir * (L + 1) + s + 1) * B + bb) * H + j;
      const float cp = s > 0 ? A.cseq[st_prev] : 0.f, hp = s > 0 ? A.hseq[st_prev] : 0.f;
      const float cn = fg * cp + ig * gg;
      const float hn = og * tanhf(cn);
      const bool valid = t < A.lengths[bb];
      A.cseq[st_new] = valid ? cn : cp;
      A.hseq[st_new] = valid ? hn : hp;
      float* gs = A.gates + (((long)t * B + bb) * 2 + dir) * 4 * H + j;
      gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
      A.hprev_t[(((long)t * B + bb) * 2 + dir) * H + j] = hp;
      A.out[((long)bb * L + t) * 2 * H + dir * H + j] = valid ? hn : 0.f;
    }
  }
}

extern "C" int drn_lstm_step_fwd(const float* xproj, const float* Whh_f, const float* Whh_r, const float* b_ih_f, const float* b_hh_f,
                                 const float* b_ih_r, const float* b_hh_r, float* hseq, float* cseq, float* gates, float* out,
                                 float* hprev_t, const int64_t* lengths, int B, int L, int H, int s, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(xproj && Whh_f && Whh_r && b_ih_f && b_hh_f && b_ih_r && b_hh_r && hseq && cseq && gates && out && hprev_t && lengths,
                "drn_lstm_step_fwd: null pointer");
  DRN_CHECK_ARG(B > 0 && B <= 16 * MAX_BT && L > 0 && H % 64 == 0 && s >= 0 && s < L, "drn_lstm_step_fwd: need B<=64, H%%64==0");
  LstmFwdArgs A;
  A.xproj = xproj; A.Whh[0] = Whh_f; A.Whh[1] = Whh_r; A.hseq = hseq; A.cseq = cseq; A.gates = gates; A.out = out;
  A.hprev_t = hprev_t; A.b_ih[0] = b_ih_f; A.b_hh[0] = b_hh_f; A.b_ih[1] = b_ih_r; A.b_hh[1] = b_hh_r;
  A.lengths = (const long long*)lengths; A.B = B; A.L = L; A.H = H; A.s = s;
  dim3 grid(H / 16, 2);
  const int nbt = cdiv(B, 16);
  if (nbt == 1) lstm_step_fwd_kernel<1><<<grid, LSTM_THREADS, 0, (hipStream_t)stream>>>(A);
  else if (nbt == 2) lstm_step_fwd_kernel<2><<<grid, LSTM_THREADS, 0, (hipStream_t)stream>>>(A);
  else lstm_step_fwd_kernel<4><<<grid, LSTM_THREADS, 0, (hipStream_t)stream>>>(A);
  return drn_launch_status("drn_lstm_step_fwd");
}

// ------------------------------------------------------------------------------------------------ backward
struct LstmBwdArgs {
  const float* dout;     // [B][L][2H]
  const float* gates;    // activated i,f,g,o
  const float* cseq;
  const float* WhhT[2];  // [H][4H] = Whh^T (contiguous along the gate row index)
  float* dgates;         // [L][B][2][4H] by time
  float* dh;             // [2][B][H]  recurrent dL/dh entering step s (in/out)
  float* dc;             // [2][B][H]
  float* dh_pass;        // [2][B][H]  scratch
  const long long* lengths;
  int B, L, H, s;
};

// pointwise: dgates for step s, dc for step s-1, and the part of dh that bypasses the cell at padded positions
__global__ void lstm_step_bwd_pointwise_kernel(const LstmBwdArgs A) {
  const int B = A.B, L = A.L, H = A.H, s = A.s;
  const int total = 2 * B * H;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx % H, bb = (idx / H) % B, dir = idx / (H * B);
    const int t = dir == 0 ? s : L - 1 - s;
    const bool valid = t < A.lengths[bb];
    const long sidx = ((long)dir * B + bb) * H + j;
    float* dg = A.dgates + (((long)t * B + bb) * 2 + dir) * 4 * H + j;
    const bool first = s == L - 1;                       // nothing flows in from beyond the last step
    const float dh = (first ? 0.f : A.dh[sidx]) + (valid ? A.dout[((long)bb * L + t) * 2 * H + dir * H + j] : 0.f);
    const float dcn = first ? 0.f : A.dc[sidx];
    if (valid) {
      const float* gs = A.gates + (((long)t * B + bb) * 2 + dir) * 4 * H + j;
      const float ig = gs[0], fg = gs[H], gg = gs[2 * H], og = gs[3 * H];
      const float cn = A.cseq[((long)(dir * (L + 1) + s + 1) * B + bb) * H + j];
      const float cp = s > 0 ? A.cseq[((long)(dir * (L + 1) + s) * B + bb) * H + j] : 0.f;
      const float tc = tanhf(cn);
      const float dcv = dcn + dh * og * (1.f - tc * tc);
      dg[0] = dcv * gg * ig * (1.f - ig);
      dg[H] = dcv * cp * fg * (1.f - fg);
      dg[2 * H] = dcv * ig * (1.f - gg * gg);
      dg[3 * H] = dh * tc * og * (1.f - og);
      A.dc[sidx] = dcv * fg;
      A.dh_pass[sidx] = 0.f;
    } else {
      dg[0] = 0.f; dg[H] = 0.f; dg[2 * H] = 0.f; dg[3 * H] = 0.f;
      A.dc[sidx] = dcn;
      A.dh_pass[sidx] = dh;
    }
  }
}

// dh[dir][b][k] = dh_pass + sum_r dgates[dir][s][b][r] * Whh[dir][r][k];  grid (H/16, 2), 4 waves split r (K = 4H)
template <int NBT>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_step_bwd_gemm_kernel(const LstmBwdArgs A) {
  __shared__ float red[4][NBT][64][4];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int dir = blockIdx.y, k0 = blockIdx.x * 16;
  const int B = A.B, L = A.L, H = A.H, s = A.s;
  const int K = 4 * H, kq = K / 4;
  const int t = dir == 0 ? s : L - 1 - s;
  const float* dg = A.dgates + ((long)t * B * 2 + dir) * K;      // row bb at dg + bb * 2K
  const float* WT = A.WhhT[dir];
  f32x4 acc[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) acc[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int row = l & 15, rc = (l >> 4) * 4;
  for (int r0 = w * kq; r0 < (w + 1) * kq; r0 += 16) {
    f32x4 a[NBT];
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) {
      const int bb = bt * 16 + row;
      a[bt] = bb < B ? *(const f32x4*)(dg + (long)bb * 2 * K + r0 + rc) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 b = *(const f32x4*)(WT + (long)(k0 + row) * K + r0 + rc);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) acc[bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[bt][e], b[e], acc[bt], 0, 0, 0);
  }
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][bt][l][r] = acc[bt][r];
  __syncthreads();
  for (int bt = w; bt < NBT; bt += 4)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bb = bt * 16 + (l >> 4) * 4 + r;
      if (bb >= B) continue;
      const long sidx = ((long)dir * B + bb) * H + k0 + (l & 15);
      A.dh[sidx] = A.dh_pass[sidx] + red[0][bt][l][r] + red[1][bt][l][r] + red[2][bt][l][r] + red[3][bt][l][r];
    }
}

extern "C" int drn_lstm_step_bwd(const float* dout, const float* gates, const float* cseq, const float* WhhT_f, const float* WhhT_r,
                                 float* dgates, float* dh, float* dc, float* dh_pass, const int64_t* lengths, int B, int L, int H, int s,
                                 void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(dout && gates && cseq && WhhT_f && WhhT_r && dgates && dh && dc && dh_pass && lengths, "drn_lstm_step_bwd: null pointer");
  DRN_CHECK_ARG(B > 0 && B <= 16 * MAX_BT && L > 0 && H % 64 == 0 && s >= 0 && s < L, "drn_lstm_step_bwd: need B<=64, H%%64==0");
  LstmBwdArgs A;
  A.dout = dout; A.gates = gates; A.cseq = cseq; A.WhhT[0] = WhhT_f; A.WhhT[1] = WhhT_r; A.dgates = dgates; A.dh = dh; A.dc = dc;
  A.dh_pass = dh_pass; A.lengths = (const long long*)lengths; A.B = B; A.L = L; A.H = H; A.s = s;
  lstm_step_bwd_pointwise_kernel<<<cdiv(2 * B * H, 256), 256, 0, stream>>>(A);
  dim3 grid(H / 16, 2);
  const int nbt = cdiv(B, 16);
  if (nbt == 1) lstm_step_bwd_gemm_kernel<1><<<grid, LSTM_THREADS, 0, stream>>>(A);
  else if (nbt == 2) lstm_step_bwd_gemm_kernel<2><<<grid, LSTM_THREADS, 0, stream>>>(A);
  else lstm_step_bwd_gemm_kernel<4><<<grid, LSTM_THREADS, 0, stream>>>(A);
  return drn_launch_status("drn_lstm_step_bwd");
}
