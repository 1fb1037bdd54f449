// Query-encoder glue of the DRN path (model/language_module.py:17-63): everything around the LSTM recurrence
// (lstm.hip) and the few dense products (library GEMMs issued by the caller) that used to be ~60 tiny framework
// kernels per step: embedding gather / scatter, the [first ; last] sentence vector, and the three attention
// "commands" (logits -> masked softmax -> weighted sum) with their backward.  All fp32, batch-sized work, one
// workgroup per clip where a reduction over words or channels is needed; no atomics (bitwise reproducible).
#include "common.h"
#include "../../include/drn_hip.h"

#define QE_MAX_L 64      // words per query (Charades-STA / ANet / TACoS queries are shorter after the data layer truncates them)
#define QE_NCMD 3

// ---------------------------------------------------------------- embedding
// out_tm[t][b][:] = table[tokens[b][t]][:]   (time-major, the layout the input projection GEMM and the LSTM read)
__global__ void qe_embed_fwd_kernel(const long long* __restrict__ tokens, const float* __restrict__ table, float* __restrict__ out_tm,
                                    int B, int L, int E) {
  const long total = (long)L * B * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i % E);
    const int tb = (int)(i / E);
    const int b = tb % B, t = tb / B;
    out_tm[i] = table[tokens[(long)b * L + t] * E + e];
  }
}
extern "C" int drn_qe_embed_fwd(const int64_t* tokens, const float* table, float* out_tm, int B, int L, int E, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(tokens && table && out_tm && B > 0 && L > 0 && E > 0, "drn_qe_embed_fwd: bad args");
  const long total = (long)L * B * E;
  qe_embed_fwd_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>((const long long*)tokens, table, out_tm, B, L, E);
  return drn_launch_status("drn_qe_embed_fwd");
}

// dtable[v][:] = sum over (b,t) with tokens[b][t] == v of demb_tm[t][b][:], row `padding_idx` = 0 (nn.Embedding(padding_idx));
// one workgroup per vocabulary row scans the B*L tokens in order, so the sum order is fixed and the dense
// gradient needs no zero-fill pass.
__global__ __launch_bounds__(128) void qe_embed_bwd_kernel(const long long* __restrict__ tokens, const float* __restrict__ demb_tm,
                                                           float* __restrict__ dtable, int B, int L, int E, int padding_idx) {
  __shared__ unsigned long long mask[QE_MAX_L];          // one 64-token ballot per entry, B*L <= 64*QE_MAX_L
  const int v = blockIdx.x, n = B * L, nch = (n + 63) >> 6;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int ch = wv; ch < nch; ch += 2) {
    const int i = ch * 64 + lane;
    const unsigned long long m = __ballot(i < n && v != padding_idx && tokens[i] == v);
    if (lane == 0) mask[ch] = m;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float acc = 0.f;
    for (int ch = 0; ch < nch; ++ch) {
      unsigned long long m = mask[ch];
      while (m) {                                          // set bits in ascending token order: fixed summation order
        const int i = ch * 64 + __ffsll((long long)m) - 1;
        m &= m - 1;
        acc += demb_tm[((long)(i % L) * B + i / L) * E + e];   // (b,t) -> time-major row
      }
    }
    dtable[(long)v * E + e] = acc;
  }
}
extern "C" int drn_qe_embed_bwd(const int64_t* tokens, const float* demb_tm, float* dtable, int B, int L, int E, int V,
                                int padding_idx, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(tokens && demb_tm && dtable && B > 0 && L > 0 && E > 0 && V > 0, "drn_qe_embed_bwd: bad args");
  DRN_CHECK_ARG(B * L <= QE_MAX_L * 64, "drn_qe_embed_bwd: B*L > %d", QE_MAX_L * 64);
  qe_embed_bwd_kernel<<<V, 128, 0, (hipStream_t)stream>>>((const long long*)tokens, demb_tm, dtable, B, L, E, padding_idx);
  return drn_launch_status("drn_qe_embed_bwd");
}

// ---------------------------------------------------------------- [first ; last] sentence vector (language_module.py:48-54)
// qvec[b] = [out[b][0][:], out[b][len_b - 1][:]]
__global__ void qe_qvec_fwd_kernel(const float* __restrict__ out, const long long* __restrict__ lengths, float* __restrict__ qvec,
                                   int B, int L, int C) {
  const int total = B * 2 * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % C, half = (i / C) & 1, b = i / (2 * C);
    const int t = half ? (int)lengths[b] - 1 : 0;
    qvec[i] = out[((long)b * L + t) * C + c];
  }
}
// dout[b][0] += dqvec[b][:C]; dout[b][len_b-1] += dqvec[b][C:]  (one thread does both, so len_b == 1 is race-free)
__global__ void qe_qvec_bwd_kernel(const float* __restrict__ dqvec, const long long* __restrict__ lengths, float* __restrict__ dout,
                                   int B, int L, int C) {
  const int total = B * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % C, b = i / C;
    const int t = (int)lengths[b] - 1;
    dout[((long)b * L) * C + c] += dqvec[(long)b * 2 * C + c];
    dout[((long)b * L + t) * C + c] += dqvec[(long)b * 2 * C + C + c];
  }
}
extern "C" int drn_qe_qvec_fwd(const float* out, const int64_t* lengths, float* qvec, int B, int L, int C, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(out && lengths && qvec && B > 0 && L > 0 && C > 0, "drn_qe_qvec_fwd: bad args");
  qe_qvec_fwd_kernel<<<cdiv(B * 2 * C, 256), 256, 0, (hipStream_t)stream>>>(out, (const long long*)lengths, qvec, B, L, C);
  return drn_launch_status("drn_qe_qvec_fwd");
}
extern "C" int drn_qe_qvec_bwd(const float* dqvec, const int64_t* lengths, float* dout, int B, int L, int C, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(dqvec && lengths && dout && B > 0 && L > 0 && C > 0, "drn_qe_qvec_bwd: bad args");
  qe_qvec_bwd_kernel<<<cdiv(B * C, 256), 256, 0, (hipStream_t)stream>>>(dqvec, (const long long*)lengths, dout, B, L, C);
  return drn_launch_status("drn_qe_qvec_bwd");
}

// ---------------------------------------------------------------- attention commands (language_module.py:17-36)
//   logit[b][t][l] = bias + sum_c q_cmd[b][t][c] * w[c] * out[b][l][c]      (cmd_inter2logits on q_cmd[:,None,:] * out)
//   att[b][t][:]   = softmax over l < len_b (padded positions masked out)
//   cmd[t][b][:]   = sum_l att[b][t][l] * out[b][l][:]
// one workgroup (256 threads) per clip.
// All 3*len dot products of a clip are accumulated together: a thread owns channels c = tid, tid+256, ... and keeps one
// partial per (command, word) of the current 8-word chunk, so `out` is read once with independent loads; wave shuffles
// + a 4-wave LDS sum finish the chunk.
template <bool BWD>
__device__ __forceinline__ void qe_dots(const float* __restrict__ ob, const float* __restrict__ v0, const float* __restrict__ v1,
                                        const float* __restrict__ v2, const float* __restrict__ w, int len, int C,
                                        float (*res)[QE_MAX_L], float (*part)[QE_NCMD][8]) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int l0 = 0; l0 < len; l0 += 8) {
    float acc[QE_NCMD][8];
#pragma unroll
    for (int t = 0; t < QE_NCMD; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
      const float wc = BWD ? 1.f : w[c];
      const float q0 = v0 ? v0[c] * wc : 0.f, q1 = v1 ? v1[c] * wc : 0.f, q2 = v2 ? v2[c] * wc : 0.f;
      // unconditional loads (row index clamped, the value masked): a load under a run-time condition gets its own branch and
      // its own vmcnt(0), i.e. eight serial memory round trips per channel here
      float ov[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ov[j] = ob[(long)max(min(l0 + j, len - 1), 0) * C + c];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float o = l0 + j < len ? ov[j] : 0.f;
        acc[0][j] = fmaf(q0, o, acc[0][j]);
        acc[1][j] = fmaf(q1, o, acc[1][j]);
        acc[2][j] = fmaf(q2, o, acc[2][j]);
      }
    }
#pragma unroll
    for (int t = 0; t < QE_NCMD; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float r = wave_sum(acc[t][j]);
        if (lane == 0) part[wv][t][j] = r;
      }
    __syncthreads();
    if (threadIdx.x < QE_NCMD * 8) {
      const int t = threadIdx.x >> 3, j = threadIdx.x & 7;
      if (l0 + j < len) res[t][l0 + j] = part[0][t][j] + part[1][t][j] + part[2][t][j] + part[3][t][j];
    }
    __syncthreads();
  }
}

// ---- fast path of the two attention kernels: L <= 8 words, C <= 1024 channels (C % 4 == 0) -- the DRN query encoder.
// A thread owns ONE 16-byte channel quad and keeps the clip's rows of `out` in registers for both passes, so the kernel is one
// memory round trip (all loads requested up front), one 32-value wave reduction (value index -> lane >> 1: at offsets 32 .. 2 a
// lane hands the half of its values the partner will own to it; 32 shuffles instead of 6 per value), a softmax done by 24
// lanes with 8-lane shuffles, and the weighted sums out of registers.  The general kernels below walk channels and words in
// loops (one round trip per 256 channels and 8 words: 16 us for a 32-clip batch where this path takes ~7).
#define QE_FAST_L 8
__device__ __forceinline__ float qe_dot4(const f32x4& a, const f32x4& b) {
  return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])));
}
// red[0..31] summed over the 64 lanes; every lane returns the total of value (lane >> 1)
__device__ __forceinline__ float qe_reduce32(float (&red)[32], int lane) {
#pragma unroll
  for (int stage = 0; stage < 5; ++stage) {
    const int off = 32 >> stage, half = 16 >> stage;
    const bool upper = (lane & off) != 0;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < half) {
        // pin both candidates in registers first: left alone, the compiler rewrites `upper ? red[k] : red[k + half]` as
        // red[k + (upper ? 0 : half)], a run-time index into a register array = a 32-way compare/select chain per access
        // (3000 instructions, 10 us of this kernel)
        float lo = red[k], hi = red[k + half];
        asm volatile("" : "+v"(lo), "+v"(hi));
        const float send = upper ? lo : hi;
        const float keep = upper ? hi : lo;
        red[k] = keep + __shfl_xor(send, off, 64);
      }
  }
  return red[0] + __shfl_xor(red[0], 1, 64);
}
__device__ __forceinline__ float qe_group8_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}
__device__ __forceinline__ float qe_group8_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64));
  return v;
}

__global__ __launch_bounds__(256) void qe_attn_fwd_small_kernel(const float* __restrict__ out, const float* __restrict__ qcmd,
                                                                const float* __restrict__ w, const float* __restrict__ bias,
                                                                const long long* __restrict__ lengths, float* __restrict__ att,
                                                                float* __restrict__ cmds, int B, int L, int C) {
  __shared__ float part[4][32];
  __shared__ float lg[QE_NCMD][QE_FAST_L];
  const int b = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int len = min((int)lengths[b], L);
  const int c = threadIdx.x * 4;
  const bool live = c < C;
  const int cc = live ? c : 0;
  const float* ob = out + (long)b * L * C + cc;
  const float* qb = qcmd + (long)b * QE_NCMD * C + cc;
  const f32x4 w4 = *(const f32x4*)(w + cc);
  f32x4 q[QE_NCMD], ov[QE_FAST_L];
#pragma unroll
  for (int t = 0; t < QE_NCMD; ++t) q[t] = *(const f32x4*)(qb + (long)t * C);
#pragma unroll
  for (int j = 0; j < QE_FAST_L; ++j) ov[j] = *(const f32x4*)(ob + (long)max(min(j, len - 1), 0) * C);   // clamped rows are masked below
  const float bs = bias[0];
  float red[32];
#pragma unroll
  for (int t = 0; t < QE_NCMD; ++t) {
    const f32x4 qw = q[t] * w4;
#pragma unroll
    for (int j = 0; j < QE_FAST_L; ++j) red[t * 8 + j] = (live && j < len) ? qe_dot4(qw, ov[j]) : 0.f;
  }
#pragma unroll
  for (int k = 24; k < 32; ++k) red[k] = 0.f;
  const float tot = qe_reduce32(red, lane);
  if ((lane & 1) == 0) part[wv][lane >> 1] = tot;
  __syncthreads();
  if (threadIdx.x < QE_NCMD * 8) {                    // 24 lanes of wave 0: (command t, word j); softmax over the 8 lanes of a command
    const int t = threadIdx.x >> 3, j = threadIdx.x & 7;
    const float logit = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]) + bs;
    const float mx = qe_group8_max(j < len ? logit : -INFINITY);
    const float e = j < len ? expf(logit - mx) : 0.f;
    const float a = e / qe_group8_sum(e);
    lg[t][j] = j < len ? a : 0.f;
    if (j < L) att[((long)b * QE_NCMD + t) * L + j] = j < len ? a : 0.f;
  }
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int t = 0; t < QE_NCMD; ++t) {
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < QE_FAST_L; ++j) a += lg[t][j] * ov[j];            // lg is zero beyond len
    *(f32x4*)(cmds + ((long)t * B + b) * C + c) = a;
  }
}

__global__ __launch_bounds__(256) void qe_attn_bwd_small_kernel(const float* __restrict__ dcmd0, const float* __restrict__ dcmd1,
                                                                const float* __restrict__ dcmd2, const float* __restrict__ att,
                                                                const float* __restrict__ out, const float* __restrict__ qcmd,
                                                                const float* __restrict__ w, const long long* __restrict__ lengths,
                                                                float* __restrict__ dqcmd, float* __restrict__ dout,
                                                                float* __restrict__ dw_part, float* __restrict__ dbias_part, int B,
                                                                int L, int C) {
  __shared__ float part[4][32];
  __shared__ float at[QE_NCMD][QE_FAST_L], dl[QE_NCMD][QE_FAST_L];
  __shared__ float gsum[QE_NCMD];
  const int b = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int len = min((int)lengths[b], L);
  const int c = threadIdx.x * 4;
  const bool live = c < C;
  const int cc = live ? c : 0;
  const float* ob = out + (long)b * L * C + cc;
  const float* qb = qcmd + (long)b * QE_NCMD * C + cc;
  const float* dc[QE_NCMD] = {dcmd0, dcmd1, dcmd2};
  const f32x4 w4 = *(const f32x4*)(w + cc);
  f32x4 q[QE_NCMD], d[QE_NCMD], ov[QE_FAST_L];
#pragma unroll
  for (int t = 0; t < QE_NCMD; ++t) {
    q[t] = *(const f32x4*)(qb + (long)t * C);
    d[t] = dc[t] ? *(const f32x4*)(dc[t] + (long)b * C + cc) : (f32x4){0.f, 0.f, 0.f, 0.f};      // (uniform branch)
  }
#pragma unroll
  for (int j = 0; j < QE_FAST_L; ++j) ov[j] = *(const f32x4*)(ob + (long)max(min(j, len - 1), 0) * C);
  float a_own = 0.f;                                   // att[t][j] of the (t, j) this thread finishes below
  if (threadIdx.x < QE_NCMD * 8) {
    const int t = threadIdx.x >> 3, j = threadIdx.x & 7;
    a_own = j < len ? att[((long)b * QE_NCMD + t) * L + min(j, L - 1)] : 0.f;
  }
  float red[32];
#pragma unroll
  for (int t = 0; t < QE_NCMD; ++t)
#pragma unroll
    for (int j = 0; j < QE_FAST_L; ++j) red[t * 8 + j] = (live && j < len) ? qe_dot4(d[t], ov[j]) : 0.f;     // datt[t][j]
#pragma unroll
  for (int k = 24; k < 32; ++k) red[k] = 0.f;
  const float tot = qe_reduce32(red, lane);
  if ((lane & 1) == 0) part[wv][lane >> 1] = tot;
  __syncthreads();
  if (threadIdx.x < QE_NCMD * 8) {
    const int t = threadIdx.x >> 3, j = threadIdx.x & 7;
    const float datt = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    const float dot = qe_group8_sum(a_own * datt);
    const float dlog = a_own * (datt - dot);           // zero beyond len (a_own is)
    at[t][j] = a_own;
    dl[t][j] = dlog;
    const float s8 = qe_group8_sum(dlog);
    if (j == 0) gsum[t] = s8;
  }
  __syncthreads();
  if (threadIdx.x == 0) dbias_part[b] = (gsum[0] + gsum[1]) + gsum[2];
  if (!live) return;
  f32x4 S[QE_NCMD];
#pragma unroll
  for (int t = 0; t < QE_NCMD; ++t) S[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < QE_FAST_L; ++j) {
    if (j >= L) break;
    f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < QE_NCMD; ++t) {                // at / dl are zero beyond len: padded rows get a zero gradient
      S[t] += dl[t][j] * ov[j];
      g += at[t][j] * d[t];
      g += (dl[t][j] * q[t]) * w4;
    }
    *(f32x4*)(dout + ((long)b * L + j) * C + c) = g;
  }
  f32x4 dw = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < QE_NCMD; ++t) {
    *(f32x4*)(dqcmd + ((long)b * QE_NCMD + t) * C + c) = w4 * S[t];
    dw += q[t] * S[t];
  }
  *(f32x4*)(dw_part + (long)b * C + c) = dw;
}
static bool qe_small_ok(int L, int C, const void* const* ptrs, int n) {
  if (L > QE_FAST_L || C > 1024 || (C & 3)) return false;
  for (int i = 0; i < n; ++i)
    if (ptrs[i] && (((uintptr_t)ptrs[i]) & 15)) return false;
  return true;
}

__global__ __launch_bounds__(256) void qe_attn_fwd_kernel(const float* __restrict__ out, const float* __restrict__ qcmd,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          const long long* __restrict__ lengths, float* __restrict__ att,
                                                          float* __restrict__ cmds, int B, int L, int C) {
  __shared__ float lg[QE_NCMD][QE_MAX_L];
  __shared__ float part[4][QE_NCMD][8];
  const int b = blockIdx.x;
  const int len = min((int)lengths[b], L);
  const float* ob = out + (long)b * L * C;
  const float* qb = qcmd + (long)b * QE_NCMD * C;
  qe_dots<false>(ob, qb, qb + C, qb + 2 * C, w, len, C, lg, part);
  if (threadIdx.x < QE_NCMD) {
    const int t = threadIdx.x;
    const float bs = bias[0];
    float mx = -INFINITY;
    for (int l = 0; l < len; ++l) mx = fmaxf(mx, lg[t][l] + bs);
    float sum = 0.f;
    for (int l = 0; l < len; ++l) {
      const float e = expf(lg[t][l] + bs - mx);
      lg[t][l] = e;
      sum += e;
    }
    for (int l = 0; l < L; ++l) {
      const float a = l < len ? lg[t][l] / sum : 0.f;
      if (l < len) lg[t][l] = a;
      att[((long)b * QE_NCMD + t) * L + l] = a;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int l0 = 0; l0 < len; l0 += 8) {          // eight independent row loads per trip (clamped: lg is zero beyond len)
      float ov[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ov[j] = ob[(long)min(l0 + j, len - 1) * C + c];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int l = l0 + j;
        if (l < len) {                               // (register-only: lg beyond len is not initialised)
          a0 = fmaf(lg[0][l], ov[j], a0);
          a1 = fmaf(lg[1][l], ov[j], a1);
          a2 = fmaf(lg[2][l], ov[j], a2);
        }
      }
    }
    cmds[((long)0 * B + b) * C + c] = a0;
    cmds[((long)1 * B + b) * C + c] = a1;
    cmds[((long)2 * B + b) * C + c] = a2;
  }
}
extern "C" int drn_qe_attn_fwd(const float* out, const float* qcmd, const float* w, const float* bias, const int64_t* lengths,
                               float* att, float* cmds, int B, int L, int C, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(out && qcmd && w && bias && lengths && att && cmds && B > 0 && C > 0, "drn_qe_attn_fwd: bad args");
  DRN_CHECK_ARG(L > 0 && L <= QE_MAX_L, "drn_qe_attn_fwd: at most %d words per query", QE_MAX_L);
  const void* al[] = {out, qcmd, w, cmds};
  if (qe_small_ok(L, C, al, 4))
    qe_attn_fwd_small_kernel<<<B, 256, 0, (hipStream_t)stream>>>(out, qcmd, w, bias, (const long long*)lengths, att, cmds, B, L, C);
  else
    qe_attn_fwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(out, qcmd, w, bias, (const long long*)lengths, att, cmds, B, L, C);
  return drn_launch_status("drn_qe_attn_fwd");
}

// backward of the block above for one clip per workgroup.  dcmd[t] may be NULL (= zero gradient).
//   datt[t][l] = <dcmd[t][b], out[b][l]>;  dlog = att * (datt - sum_l att*datt)
//   S[t][c] = sum_l dlog[t][l] * out[b][l][c]
//   dqcmd[b][t][c] = w[c] * S[t][c];  dw_part[b][c] = sum_t qcmd[b][t][c] * S[t][c];  dbias_part[b] = sum_{t,l} dlog
//   dout[b][l][c] = sum_t att[t][l] * dcmd[t][b][c] + dlog[t][l] * qcmd[b][t][c] * w[c]   (0 for l >= len_b)
__global__ __launch_bounds__(256) void qe_attn_bwd_kernel(const float* __restrict__ dcmd0, const float* __restrict__ dcmd1,
                                                          const float* __restrict__ dcmd2, const float* __restrict__ att,
                                                          const float* __restrict__ out, const float* __restrict__ qcmd,
                                                          const float* __restrict__ w, const long long* __restrict__ lengths,
                                                          float* __restrict__ dqcmd, float* __restrict__ dout,
                                                          float* __restrict__ dw_part, float* __restrict__ dbias_part, int B, int L,
                                                          int C) {
  __shared__ float at[QE_NCMD][QE_MAX_L], dl[QE_NCMD][QE_MAX_L];
  __shared__ float part[4][QE_NCMD][8];
  const int b = blockIdx.x;
  const int len = min((int)lengths[b], L);
  const float* ob = out + (long)b * L * C;
  const float* qb = qcmd + (long)b * QE_NCMD * C;
  const float* dc[QE_NCMD] = {dcmd0 ? dcmd0 + (long)b * C : nullptr, dcmd1 ? dcmd1 + (long)b * C : nullptr,
                              dcmd2 ? dcmd2 + (long)b * C : nullptr};
  for (int i = threadIdx.x; i < QE_NCMD * QE_MAX_L; i += 256) {
    const int t = i / QE_MAX_L, l = i % QE_MAX_L;
    at[t][l] = l < len ? att[((long)b * QE_NCMD + t) * L + l] : 0.f;
    dl[t][l] = 0.f;
  }
  __syncthreads();
  qe_dots<true>(ob, dc[0], dc[1], dc[2], w, len, C, dl, part);          // dl = datt for now
  if (threadIdx.x < QE_NCMD) {
    const int t = threadIdx.x;
    float dot = 0.f;
    for (int l = 0; l < len; ++l) dot = fmaf(at[t][l], dl[t][l], dot);
    for (int l = 0; l < len; ++l) dl[t][l] = at[t][l] * (dl[t][l] - dot);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int t = 0; t < QE_NCMD; ++t)
      for (int l = 0; l < len; ++l) s += dl[t][l];
    dbias_part[b] = s;
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    const float wc = w[c];
    float S[QE_NCMD] = {0.f, 0.f, 0.f}, q[QE_NCMD], d[QE_NCMD];
#pragma unroll
    for (int t = 0; t < QE_NCMD; ++t) {
      q[t] = qb[t * C + c];
      d[t] = dc[t] ? dc[t][c] : 0.f;
    }
    for (int l0 = 0; l0 < L; l0 += 8) {              // eight independent row loads per trip (row index clamped)
      float ov[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ov[j] = ob[(long)max(min(l0 + j, len - 1), 0) * C + c];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int l = l0 + j;
        if (l >= L) break;
        float g = 0.f;
        if (l < len) {
          const float o = ov[j];
#pragma unroll
          for (int t = 0; t < QE_NCMD; ++t) {
            S[t] = fmaf(dl[t][l], o, S[t]);
            g = fmaf(at[t][l], d[t], g);
            g = fmaf(dl[t][l] * q[t], wc, g);
          }
        }
        dout[((long)b * L + l) * C + c] = g;
      }
    }
    float dw = 0.f;
#pragma unroll
    for (int t = 0; t < QE_NCMD; ++t) {
      dqcmd[((long)b * QE_NCMD + t) * C + c] = wc * S[t];
      dw = fmaf(q[t], S[t], dw);
    }
    dw_part[(long)b * C + c] = dw;
  }
}
extern "C" int drn_qe_attn_bwd(const float* dcmd0, const float* dcmd1, const float* dcmd2, const float* att, const float* out,
                               const float* qcmd, const float* w, const int64_t* lengths, float* dqcmd, float* dout, float* dw_part,
                               float* dbias_part, int B, int L, int C, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(att && out && qcmd && w && lengths && dqcmd && dout && dw_part && dbias_part && B > 0 && C > 0,
                "drn_qe_attn_bwd: bad args");
  DRN_CHECK_ARG(L > 0 && L <= QE_MAX_L, "drn_qe_attn_bwd: at most %d words per query", QE_MAX_L);
  const void* al[] = {dcmd0, dcmd1, dcmd2, out, qcmd, w, dqcmd, dout, dw_part};
  if (qe_small_ok(L, C, al, 9))
    qe_attn_bwd_small_kernel<<<B, 256, 0, (hipStream_t)stream>>>(dcmd0, dcmd1, dcmd2, att, out, qcmd, w, (const long long*)lengths, dqcmd,
                                                                  dout, dw_part, dbias_part, B, L, C);
  else
    qe_attn_bwd_kernel<<<B, 256, 0, (hipStream_t)stream>>>(dcmd0, dcmd1, dcmd2, att, out, qcmd, w, (const long long*)lengths, dqcmd, dout,
                                                            dw_part, dbias_part, B, L, C);
  return drn_launch_status("drn_qe_attn_bwd");
}

// ---------------------------------------------------------------- column sums of an fp32 matrix into several destinations
// seg s: dst[s][j] = sum_m X[m][col0[s] + j], j < n[s]  (bias gradients that live in different parameters; a column range
// may appear twice, e.g. b_ih and b_hh of an LSTM receive the same gradient).  block 256 = 32 columns x 8 row lanes.
struct ColSegParams {
  float* dst[DRN_COLSEG_MAX];
  int col0[DRN_COLSEG_MAX], n[DRN_COLSEG_MAX], blk0[DRN_COLSEG_MAX];
  int nseg;
};
__global__ __launch_bounds__(256) void colsum_segs_kernel(const float* __restrict__ X, int ld, int M, ColSegParams P) {
  __shared__ float red[8][33];
  int s = 0;
  for (int i = 1; i < P.nseg; ++i)
    if ((int)blockIdx.x >= P.blk0[i]) s = i;
  const int jl = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int j = (blockIdx.x - P.blk0[s]) * 32 + jl;
  float acc = 0.f;
  if (j < P.n[s])
#pragma unroll 4
    for (int m = ry; m < M; m += 8) acc += X[(long)m * ld + P.col0[s] + j];
  red[ry][jl] = acc;
  __syncthreads();
  if (ry == 0 && j < P.n[s]) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][jl];
    P.dst[s][j] = t;
  }
}
extern "C" int drn_colsum_segs(const float* X, int ld, int M, const DrnColSeg* segs, int nsegs, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(X && segs && M > 0 && nsegs > 0 && nsegs <= DRN_COLSEG_MAX, "drn_colsum_segs: bad args (at most %d segments)",
                DRN_COLSEG_MAX);
  ColSegParams P;
  int blocks = 0;
  for (int i = 0; i < nsegs; ++i) {
    DRN_CHECK_ARG(segs[i].dst && segs[i].n > 0 && segs[i].col0 >= 0 && segs[i].col0 + segs[i].n <= ld, "drn_colsum_segs: bad segment %d", i);
    P.dst[i] = segs[i].dst; P.col0[i] = segs[i].col0; P.n[i] = segs[i].n; P.blk0[i] = blocks;
    blocks += cdiv(segs[i].n, 32);
  }
  P.nseg = nsegs;
  colsum_segs_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(X, ld, M, P);
  return drn_launch_status("drn_colsum_segs");
}
