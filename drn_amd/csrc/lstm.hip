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

// grid (H/16, 2 dirs, batch tiles of 16); block 256 = 4 waves, wave w reduces k in [w*H/4, (w+1)*H/4).  The step is
// latency-bound, so every operand of a wave's K range is requested before the first MFMA (KU iterations of 16 at a time).
template <int KU>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_step_fwd_kernel(const LstmFwdArgs A) {
  __shared__ float red[4][4][64][4];   // [wave][gate][lane][reg]
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int dir = blockIdx.y, j0 = blockIdx.x * 16, bt = blockIdx.z;
  const int B = A.B, L = A.L, H = A.H, s = A.s;
  const int t = dir == 0 ? s : L - 1 - s;
  const float* hprev = A.hseq + ((long)(dir * (L + 1) + s) * B) * H;
  const float* W = A.Whh[dir];
  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int kq = H / 4;
  const int row = l & 15, kc = (l >> 4) * 4;
  const int bb_a = bt * 16 + row;
  // The cell update's own operands (input projection, biases, previous state, length) are requested BEFORE the recurrent
  // product: issued in the epilogue they were a second, serial memory round trip of ~1 us in an 8 us kernel.
  const int e_bb = bt * 16 + (l >> 4) * 4 + w, e_j = j0 + (l & 15);
  const bool e_own = e_bb < B;
  float e_x[4], e_bi[4], e_bh[4], e_cp = 0.f, e_hp = 0.f;
  long long e_len = 0;
  {
    const int bq = e_own ? e_bb : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      e_x[g] = A.xproj[(((long)t * B + bq) * 2 + dir) * 4 * H + g * H + e_j];
      e_bi[g] = A.b_ih[dir][g * H + e_j];
      e_bh[g] = A.b_hh[dir][g * H + e_j];
    }
    const long sp = ((long)(dir * (L + 1) + (s > 0 ? s : 1)) * B + bq) * H + e_j;     // (slot 0 is never written: not read at s = 0)
    e_cp = A.cseq[sp];
    e_hp = A.hseq[sp];
    e_len = A.lengths[bq];
    if (s == 0) { e_cp = 0.f; e_hp = 0.f; }
  }
  if (s > 0)                            // zero initial state: the first step has no recurrent term
    for (int k0 = w * kq; k0 < (w + 1) * kq; k0 += 16 * KU) {
      f32x4 a[KU], b[KU][4];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int k = k0 + u * 16 + kc;
        a[u] = bb_a < B ? *(const f32x4*)(hprev + (long)bb_a * H + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) b[u][g] = *(const f32x4*)(W + (long)(g * H + j0 + row) * H + k);
      }
#pragma unroll
      for (int u = 0; u < KU; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][g][e], acc[g], 0, 0, 0);
    }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][g][l][r] = acc[g][r];
  __syncthreads();
  // epilogue: 256 threads = 64 lanes x 4 regs of the D tile; D layout: b = bt*16 + (l>>4)*4 + r, j = j0 + (l&15)
  {
    const int r = w;                    // wave w finishes register r of every lane
    const int bb = e_bb;
    if (!e_own) return;
    const int j = e_j;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v = red[0][g][l][r] + red[1][g][l][r] + red[2][g][l][r] + red[3][g][l][r];
      pre[g] = v + e_x[g] + e_bi[g] + e_bh[g];
    }
    const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
    const long st_new = ((long)(dir * (L + 1) + s + 1) * B + bb) * H + j;
    const float cp = e_cp, hp = e_hp;
    const float cn = fg * cp + ig * gg;
    const float hn = og * tanhf(cn);
    const bool valid = t < e_len;
    A.cseq[st_new] = valid ? cn : cp;
    A.hseq[st_new] = valid ? hn : hp;
    float* gs = A.gates + (((long)t * B + bb) * 2 + dir) * 4 * H + j;
    gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
    A.hprev_t[(((long)t * B + bb) * 2 + dir) * H + j] = hp;
    A.out[((long)bb * L + t) * 2 * H + dir * H + j] = valid ? hn : 0.f;
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
  dim3 grid(H / 16, 2, cdiv(B, 16));
  const int kq16 = H / 64;               // 16-wide K iterations per wave
  if (kq16 % 8 == 0) lstm_step_fwd_kernel<8><<<grid, LSTM_THREADS, 0, (hipStream_t)stream>>>(A);
  else if (kq16 % 4 == 0) lstm_step_fwd_kernel<4><<<grid, LSTM_THREADS, 0, (hipStream_t)stream>>>(A);
  else if (kq16 % 2 == 0) lstm_step_fwd_kernel<2><<<grid, LSTM_THREADS, 0, (hipStream_t)stream>>>(A);
  else lstm_step_fwd_kernel<1><<<grid, LSTM_THREADS, 0, (hipStream_t)stream>>>(A);
  return drn_launch_status("drn_lstm_step_fwd");
}

// ------------------------------------------------------------------------------------------------ backward
struct LstmBwdArgs {
  const float* dout;     // [B][L][2H]
  const float* gates;    // activated i,f,g,o
  const float* cseq;
  const float* WhhT[2];  // [H][4H] = Whh^T (contiguous along the gate row index)
  float* dgates;         // [L][B][2][4H] by time
  float* dc;             // [2][B][H]
  float* dh_pass;        // [2][B][H]  dL/dh that bypasses the cell at padded positions (in/out)
  const long long* lengths;
  int B, L, H, s;
};

// Cell backward for one (direction, clip, hidden unit) at step s given dL/dh entering that step: writes dgates for time
// t(s), the running dL/dc for step s-1, and the part of dh that bypasses the cell at padded positions.  In two halves so that
// the step kernel can request the cell's operands BEFORE its recurrent product (they were a second, serial memory round trip).
struct LstmCellIn {
  float dout, dcn, ig, fg, gg, og, cn, cp;
  bool valid;
};
__device__ __forceinline__ LstmCellIn lstm_cell_bwd_load(const LstmBwdArgs& A, int dir, int bb, int j, int s) {
  const int B = A.B, L = A.L, H = A.H;
  const int t = dir == 0 ? s : L - 1 - s;
  LstmCellIn c;
  c.valid = t < A.lengths[bb];
  const long sidx = ((long)dir * B + bb) * H + j;
  c.dout = A.dout[((long)bb * L + t) * 2 * H + dir * H + j];
  c.dcn = s == L - 1 ? 0.f : A.dc[sidx];       // nothing flows in from beyond the last step
  const float* gs = A.gates + (((long)t * B + bb) * 2 + dir) * 4 * H + j;
  c.ig = gs[0]; c.fg = gs[H]; c.gg = gs[2 * H]; c.og = gs[3 * H];
  c.cn = A.cseq[((long)(dir * (L + 1) + s + 1) * B + bb) * H + j];
  c.cp = A.cseq[((long)(dir * (L + 1) + (s > 0 ? s : 1)) * B + bb) * H + j];
  if (s == 0) c.cp = 0.f;
  return c;
}
__device__ __forceinline__ void lstm_cell_bwd_apply(const LstmBwdArgs& A, const LstmCellIn& c, int dir, int bb, int j, int s, float dh_in) {
  const int B = A.B, L = A.L, H = A.H;
  const int t = dir == 0 ? s : L - 1 - s;
  const long sidx = ((long)dir * B + bb) * H + j;
  float* dg = A.dgates + (((long)t * B + bb) * 2 + dir) * 4 * H + j;
  const float dh = dh_in + (c.valid ? c.dout : 0.f);
  const float dcn = c.dcn;
  if (c.valid) {
    const float ig = c.ig, fg = c.fg, gg = c.gg, og = c.og, cn = c.cn, cp = c.cp;
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
__device__ __forceinline__ void lstm_cell_bwd(const LstmBwdArgs& A, int dir, int bb, int j, int s, float dh_in) {
  const LstmCellIn c = lstm_cell_bwd_load(A, dir, bb, j, s);
  lstm_cell_bwd_apply(A, c, dir, bb, j, s, dh_in);
}

// first backward step (s = L-1): no recurrent gradient yet
__global__ void lstm_bwd_first_kernel(const LstmBwdArgs A) {
  const int B = A.B, H = A.H;
  const int total = 2 * B * H;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x)
    lstm_cell_bwd(A, idx / (H * B), (idx / H) % B, idx % H, A.L - 1, 0.f);
}

// dh[dir][b][k] = dh_pass + sum_r dgates[t(s)][b][dir][r] * Whh[dir][r][k], immediately consumed by the cell backward of
// step s-1 for the same (b, k) -- dh itself never goes to memory.  grid (H/16, 2, batch tiles), 4 waves split r (K = 4H),
// KU iterations of 16 prefetched at a time.
template <int KU>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_step_bwd_kernel(const LstmBwdArgs A) {
  __shared__ float red[4][64][4];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int dir = blockIdx.y, k0 = blockIdx.x * 16, bt = blockIdx.z;
  const int B = A.B, L = A.L, H = A.H, s = A.s;
  const int K = 4 * H, kq = K / 4;
  const int t = dir == 0 ? s : L - 1 - s;
  const float* dg = A.dgates + ((long)t * B * 2 + dir) * K;      // row bb at dg + bb * 2K
  const float* WT = A.WhhT[dir];
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int row = l & 15, rc = (l >> 4) * 4;
  const int bb_a = bt * 16 + row;
  // operands of the cell backward this thread will run in the epilogue: requested up front
  const int e_bb = bt * 16 + (l >> 4) * 4 + w, e_k = k0 + (l & 15);
  const bool e_own = e_bb < B;
  const LstmCellIn e_c = lstm_cell_bwd_load(A, dir, e_own ? e_bb : 0, e_k, s - 1);
  const float e_dhp = A.dh_pass[((long)dir * B + (e_own ? e_bb : 0)) * H + e_k];
  for (int r0 = w * kq; r0 < (w + 1) * kq; r0 += 16 * KU) {
    f32x4 a[KU], b[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int r = r0 + u * 16 + rc;
      a[u] = bb_a < B ? *(const f32x4*)(dg + (long)bb_a * 2 * K + r) : (f32x4){0.f, 0.f, 0.f, 0.f};
      b[u] = *(const f32x4*)(WT + (long)(k0 + row) * K + r);
    }
#pragma unroll
    for (int u = 0; u < KU; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][e], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][l][r] = acc[r];
  __syncthreads();
  const int r = w;
  if (!e_own) return;
  const float dh = e_dhp + red[0][l][r] + red[1][l][r] + red[2][l][r] + red[3][l][r];
  lstm_cell_bwd_apply(A, e_c, dir, e_bb, e_k, s - 1, dh);
}

extern "C" int drn_lstm_bwd_first(const float* dout, const float* gates, const float* cseq, float* dgates, float* dc, float* dh_pass,
                                  const int64_t* lengths, int B, int L, int H, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(dout && gates && cseq && dgates && dc && dh_pass && lengths, "drn_lstm_bwd_first: null pointer");
  DRN_CHECK_ARG(B > 0 && B <= 16 * MAX_BT && L > 0 && H % 64 == 0, "drn_lstm_bwd_first: need B<=64, H%%64==0");
  LstmBwdArgs A;
  memset(&A, 0, sizeof(A));
  A.dout = dout; A.gates = gates; A.cseq = cseq; A.dgates = dgates; A.dc = dc; A.dh_pass = dh_pass;
  A.lengths = (const long long*)lengths; A.B = B; A.L = L; A.H = H; A.s = L - 1;
  lstm_bwd_first_kernel<<<cdiv(2 * B * H, 256), 256, 0, (hipStream_t)stream>>>(A);
  return drn_launch_status("drn_lstm_bwd_first");
}

extern "C" int drn_lstm_step_bwd(const float* dout, const float* gates, const float* cseq, const float* WhhT_f, const float* WhhT_r,
                                 float* dgates, float* dc, float* dh_pass, const int64_t* lengths, int B, int L, int H, int s,
                                 void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(dout && gates && cseq && WhhT_f && WhhT_r && dgates && dc && dh_pass && lengths, "drn_lstm_step_bwd: null pointer");
  DRN_CHECK_ARG(B > 0 && B <= 16 * MAX_BT && L > 0 && H % 64 == 0 && s >= 1 && s < L, "drn_lstm_step_bwd: need B<=64, H%%64==0, 1<=s<L");
  LstmBwdArgs A;
  memset(&A, 0, sizeof(A));
  A.dout = dout; A.gates = gates; A.cseq = cseq; A.WhhT[0] = WhhT_f; A.WhhT[1] = WhhT_r; A.dgates = dgates; A.dc = dc;
  A.dh_pass = dh_pass; A.lengths = (const long long*)lengths; A.B = B; A.L = L; A.H = H; A.s = s;
  dim3 grid(H / 16, 2, cdiv(B, 16));
  const int kq16 = H / 16;               // 16-wide K iterations per wave (K = 4H over 4 waves)
  if (kq16 % 8 == 0) lstm_step_bwd_kernel<8><<<grid, LSTM_THREADS, 0, stream>>>(A);
  else lstm_step_bwd_kernel<4><<<grid, LSTM_THREADS, 0, stream>>>(A);
  return drn_launch_status("drn_lstm_step_bwd");
}
