// Bidirectional single-layer LSTM recurrence of the DRN query encoder (model/language_module.py:13-15,38-45:
// nn.LSTM(300, 512, bidirectional, batch_first) on packed sequences), one launch per time step for BOTH directions,
// with sequence lengths on the device (no packing, no host sync -> hipGraph-capturable, unlike the MIOpen RNN path).
//
// The input projection x_t W_ih^T + b_ih + b_hh for all t is a plain GEMM done by the caller; these kernels do the
// recurrent part, which is latency-bound: per step and direction a (B=32..64) x 2048 x 512 product plus the cell
// update.  It runs on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) -- or, with the optimizer-maintained bf16 copies of
// W_hh / W_hh^T (the bf16 model), on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- straight from L2-resident operands:
// the waves of a workgroup split K, partial tiles meet in LDS and the cell update is fused into the epilogue.  Padded positions (t >= len[b]) keep the state and emit zeros, which reproduces
// pack_padded_sequence / pad_packed_sequence semantics for both directions.
//
// Layouts (fp32): xproj / gates / dgates [L][B][2][4H] indexed by TIME t and direction (gate order i,f,g,o as in
// PyTorch) -- one (L*B) x 8H matrix, so the input projection and its weight / input gradients are single GEMMs over
// both directions; Whh [2] pointers to [4H][H]; hseq / cseq [2][L+1][B][H] by STEP (slot s+1 = state after step s; slot 0
// is never read, the initial state is zero); hprev_t [L][B][2][H] = hidden state that entered time t (operand of the
// W_hh gradient); out [B][L][2H].  Step s handles t = s for the forward direction and t = L-1-s for the reverse one.
// xproj holds x_t W_ih^T only: both bias vectors are added here.
#include <type_traits>
#include "common.h"
#include "../../include/drn_hip.h"

#define LSTM_THREADS 256
#define MAX_BT 4   // batch tiles of 16 -> B <= 64

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

struct LstmFwdArgs {
  const float* xproj;
  const void* Whh[2];         // [4H][H], fp32 or (w_bf16) the optimizer-maintained bf16 copy in the same element order
  float* hseq;
  _Float16* hseq16;           // optional fp16 copy of hseq (written by every step): with it the recurrent product reads 32 KB of h per
                              // workgroup instead of 64 and runs on v_mfma_f32_16x16x32_f16 (W_hh rounded to fp16 in registers)
  float* cseq;
  float* gates;
  float* out;
  float* hprev_t;
  float* qvec;                // optional [B][4H]: [out[b][0][:], out[b][len_b-1][:]] (language_module.py:48-54), written by the steps that produce those rows
  const float* b_ih[2];
  const float* b_hh[2];
  const long long* lengths;   // [B] int64 (the dtype the data layer hands over)
  int B, L, H, s;
};

// grid (H/4, 2 dirs); block 256 = 4 waves.  A workgroup owns FOUR hidden units -- their i,f,g,o gates are the 16 columns of one MFMA
// tile, column c = unit * 4 + gate -- for ALL batch rows (MT tiles of 16), its waves split K; the partial tiles meet in LDS and
// thread (clip, unit) runs the cell update.  256 workgroups of H = 512 fill the chip and each pulls 32 KB of W_hh (16 KB in
// bf16) + the 64 KB of h through its L1 per step; round 3's first layout (16 units x 16 clips per workgroup, 128 workgroups,
// 160 KB and 128 fp32 MFMAs per wave) took 7.9 us per step hot and alone (scripts/bench_nodes.py), ~6 of them in the kernel.
// Every operand of a wave's K range -- and the cell update's own operands -- is requested before the first MFMA.
//   WT = float : v_mfma_f32_16x16x4_f32 (exact fp32, the parity mode), KU iterations of 16 k at a time
//   (a bf16 variant -- bf16 W_hh copy, h rounded to bf16 on load -- was measured and dropped: it moved a head output of configs[3]
//   past its 6e-2 gate)
//   WT = _Float16: the bf16 MODEL's forward: h from its fp16 copy (|h| < 1, unit roundoff 2^-11 -- four times finer than bf16), W_hh
//                fp32 from memory and rounded to fp16 in registers, v_mfma_f32_16x16x32_f16, fp32 accumulation: 64 KB per workgroup
//                instead of 96, and a step is as long as its operand bytes (section 5 of DESIGN.md)
template <typename WT, int MT, int KU>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_step_fwd_kernel(const LstmFwdArgs A) {
  __shared__ float red[4][MT][64][4];   // [wave][batch tile][lane][reg]
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int dir = blockIdx.y, j0 = blockIdx.x * 4;
  const int B = A.B, L = A.L, H = A.H, s = A.s;
  const int t = dir == 0 ? s : L - 1 - s;
  const float* hprev = A.hseq + ((long)(dir * (L + 1) + s) * B) * H;
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // The cell update's own operands (input projection, biases, previous state, length) are requested BEFORE the recurrent
  // product: issued in the epilogue they were a second, serial memory round trip of ~1 us.
  const int e_bb = threadIdx.x >> 2, e_j = j0 + (threadIdx.x & 3);
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
  if (s > 0) {                          // zero initial state: the first step has no recurrent term
    const int kq = H / 4;
    const int col = l & 15;
    const long wrow = (long)((col & 3) * H + j0 + (col >> 2)) * H;     // W_hh row of MFMA column c: gate c & 3 of unit c >> 2
    if constexpr (std::is_same<WT, float>::value) {
      const float* W = (const float*)A.Whh[dir];
      const int kc = (l >> 4) * 4;
      for (int k0 = w * kq; k0 < (w + 1) * kq; k0 += 16 * KU) {
        f32x4 a[KU][MT], b[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int k = k0 + u * 16 + kc;
          b[u] = *(const f32x4*)(W + wrow + k);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int bb = mt * 16 + col;
            a[u][mt] = *(const f32x4*)(hprev + (long)(bb < B ? bb : 0) * H + k);
            if (bb >= B) a[u][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][mt][e], b[u][e], acc[mt], 0, 0, 0);
      }
    } else if constexpr (std::is_same<WT, _Float16>::value) {
      typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
      const float* W = (const float*)A.Whh[dir];
      const _Float16* h16 = A.hseq16 + ((long)(dir * (L + 1) + s) * B) * H;
      const int kc = (l >> 4) * 8;
      for (int k0 = w * kq; k0 < (w + 1) * kq; k0 += 32 * KU) {
        f16x8 a[KU][MT];
        f32x4 blo[KU], bhi[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int k = k0 + u * 32 + kc;
          blo[u] = *(const f32x4*)(W + wrow + k);
          bhi[u] = *(const f32x4*)(W + wrow + k + 4);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int bb = mt * 16 + col;
            a[u][mt] = *(const f16x8*)(h16 + (long)(bb < B ? bb : 0) * H + k);
            if (bb >= B)
#pragma unroll
              for (int e = 0; e < 8; ++e) a[u][mt][e] = (_Float16)0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          f16x8 bv;
#pragma unroll
          for (int e = 0; e < 4; ++e) { bv[e] = (_Float16)blo[u][e]; bv[4 + e] = (_Float16)bhi[u][e]; }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u][mt], bv, acc[mt], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[w][mt][l][r] = acc[mt][r];
  __syncthreads();
  // epilogue: thread (clip bb, unit u).  D layout of tile mt: clip = mt*16 + (lane>>4)*4 + reg, column = lane & 15.
  if (!e_own) return;
  {
    const int bb = e_bb, j = e_j;
    const int mt = bb >> 4, r16 = bb & 15, ln = (r16 >> 2) * 16 + (threadIdx.x & 3) * 4, rg = r16 & 3;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v = red[0][mt][ln + g][rg] + red[1][mt][ln + g][rg] + red[2][mt][ln + g][rg] + red[3][mt][ln + g][rg];
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
    if (A.hseq16) A.hseq16[st_new] = (_Float16)(valid ? hn : hp);
    float* gs = A.gates + (((long)t * B + bb) * 2 + dir) * 4 * H + j;
    gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
    A.hprev_t[(((long)t * B + bb) * 2 + dir) * H + j] = hp;
    A.out[((long)bb * L + t) * 2 * H + dir * H + j] = valid ? hn : 0.f;
    if (A.qvec) {                       // the [first ; last] sentence vector: each half-row is produced by exactly one step
      float* qv = A.qvec + (long)bb * 4 * H + dir * H + j;
      if (t == 0) qv[0] = valid ? hn : 0.f;
      if (t == e_len - 1) qv[2 * H] = hn;
    }
  }
}

template <typename WT, int MT>
static void lstm_fwd_launch(const LstmFwdArgs& A, hipStream_t stream) {
  dim3 grid(A.H / 4, 2);
  const int per = std::is_same<WT, float>::value ? 16 : 32;       // k per iteration
  const int it = A.H / 4 / per;                                   // iterations per wave
  if (it % 8 == 0 && std::is_same<WT, float>::value) lstm_step_fwd_kernel<WT, MT, 8><<<grid, LSTM_THREADS, 0, stream>>>(A);
  else if (it % 4 == 0) lstm_step_fwd_kernel<WT, MT, 4><<<grid, LSTM_THREADS, 0, stream>>>(A);
  else if (it % 2 == 0) lstm_step_fwd_kernel<WT, MT, 2><<<grid, LSTM_THREADS, 0, stream>>>(A);
  else lstm_step_fwd_kernel<WT, MT, 1><<<grid, LSTM_THREADS, 0, stream>>>(A);
}

extern "C" int drn_lstm_step_fwd(const float* xproj, const void* Whh_f, const void* Whh_r, int w_dtype, const float* b_ih_f,
                                 const float* b_hh_f, const float* b_ih_r, const float* b_hh_r, float* hseq, float* cseq, float* gates,
                                 float* out, float* hprev_t, float* qvec, void* hseq16, const int64_t* lengths, int B, int L, int H, int s,
                                 void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(xproj && Whh_f && Whh_r && b_ih_f && b_hh_f && b_ih_r && b_hh_r && hseq && cseq && gates && out && hprev_t && lengths,
                "drn_lstm_step_fwd: null pointer");
  DRN_CHECK_ARG(B > 0 && B <= 16 * MAX_BT && L > 0 && H % 64 == 0 && s >= 0 && s < L, "drn_lstm_step_fwd: need B<=64, H%%64==0");
  DRN_CHECK_ARG(w_dtype == DRN_F32, "drn_lstm_step_fwd: W_hh is fp32 (low precision enters through hseq16)");
  LstmFwdArgs A;
  DRN_CHECK_ARG(!hseq16 || (w_dtype == DRN_F32 && H % 128 == 0), "drn_lstm_step_fwd: the fp16 state copy goes with fp32 weights and H %% 128 == 0");
  A.hseq16 = (_Float16*)hseq16;
  A.xproj = xproj; A.Whh[0] = Whh_f; A.Whh[1] = Whh_r; A.hseq = hseq; A.cseq = cseq; A.gates = gates; A.out = out;
  A.hprev_t = hprev_t; A.qvec = qvec; A.b_ih[0] = b_ih_f; A.b_hh[0] = b_hh_f; A.b_ih[1] = b_ih_r; A.b_hh[1] = b_hh_r;
  A.lengths = (const long long*)lengths; A.B = B; A.L = L; A.H = H; A.s = s;
  const int mt = cdiv(B, 16);
  if (hseq16) {
    if (mt == 1) lstm_fwd_launch<_Float16, 1>(A, stream); else if (mt == 2) lstm_fwd_launch<_Float16, 2>(A, stream);
    else if (mt == 3) lstm_fwd_launch<_Float16, 3>(A, stream); else lstm_fwd_launch<_Float16, 4>(A, stream);
  } else {
    if (mt == 1) lstm_fwd_launch<float, 1>(A, stream); else if (mt == 2) lstm_fwd_launch<float, 2>(A, stream);
    else if (mt == 3) lstm_fwd_launch<float, 3>(A, stream); else lstm_fwd_launch<float, 4>(A, stream);
  }
  return drn_launch_status("drn_lstm_step_fwd");
}

// ------------------------------------------------------------------------------------------------ forward, ALL steps in one launch
// The fp16-state forward (the bf16 model) as ONE launch: the same (4 hidden units x all clips) workgroups, the same arithmetic in the same
// order as lstm_step_fwd_kernel<_Float16> -- so the same bits -- but a workgroup stays for the whole sequence, keeps its own c / h in
// registers, and receives the other workgroups' h through an exchange buffer whose elements carry their own freshness mark: |h| <= 1,
// so bit 14 of an fp16 h (the top exponent bit) is always 0, and is set to the launch's PARITY instead.  A consumer polls the data
// itself (one memory round trip per hand-off; the round-2 experiment's store -> drain -> ticket -> poll -> load took four and was no faster
// than the launches) with write-through 8-byte stores (a clip's four units) and L2-bypassing loads.  Every slot of the buffer is rewritten
// by every launch (one buffer per (B, L, H)), so stale data always has the other parity; the launch number comes from a counter every
// workgroup increments once (launch = count / workgroups).  Needs the grid resident at once (256 workgroups of H = 512: checked).
struct LstmSeqArgs {
  LstmFwdArgs f;
  unsigned long long* xch;   // [2][L][B][H / 4] words of four tagged fp16: h after step s of direction d at ((d * L + s) * B + b) * (H / 4) + j / 4
  unsigned* counter;
};
static __device__ int g_lstm_seq_timeouts;

template <int MT, int KU>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_seq_fwd_kernel(const LstmSeqArgs S) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  __shared__ float red[4][MT][64][4];
  __shared__ unsigned s_launch;
  const LstmFwdArgs& A = S.f;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int dir = blockIdx.y, j0 = blockIdx.x * 4;
  const int B = A.B, L = A.L, H = A.H;
  // (launch numbers start at 1: a zero-initialised buffer reads as parity 0)
  if (threadIdx.x == 0) s_launch = __hip_atomic_fetch_add(S.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (gridDim.x * gridDim.y) + 1u;
  const int e_bb = threadIdx.x >> 2, e_j = j0 + (threadIdx.x & 3);
  const bool e_own = e_bb < B;
  const int bq = e_own ? e_bb : 0;
  float e_bi[4], e_bh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    e_bi[g] = A.b_ih[dir][g * H + e_j];
    e_bh[g] = A.b_hh[dir][g * H + e_j];
  }
  const long long e_len = A.lengths[bq];
  float e_cp = 0.f, e_hp = 0.f;                        // this thread's (clip, unit) state: stays in registers
  const int kq = H / 4, col = l & 15, kc = (l >> 4) * 8;
  const long wrow = (long)((col & 3) * H + j0 + (col >> 2)) * H;
  const float* W = (const float*)A.Whh[dir];
  __syncthreads();
  const unsigned long long tagm = (s_launch & 1u) ? 0x4000400040004000ull : 0ull;
  const long long t0 = wall_clock64();
  // W_hh does not change between the steps: when a wave's K quarter is ONE pass of the loop below (H = 128 * KU: the benchmarked H = 512
  // with KU = 4) its fragments are converted once and stay in 4 * KU registers -- 32 KB of L2 reads and KU * 8 conversions per step and
  // workgroup less (round 6; same values, same order)
  const bool w_resident = kq == 32 * KU;
  f16x8 wv[KU];
  if (w_resident) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int k = w * kq + u * 32 + kc;
      const f32x4 lo = *(const f32x4*)(W + wrow + k), hi = *(const f32x4*)(W + wrow + k + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { wv[u][e] = (_Float16)lo[e]; wv[u][4 + e] = (_Float16)hi[e]; }
    }
  }
  for (int s = 0; s < L; ++s) {
    const int t = dir == 0 ? s : L - 1 - s;
    float e_x[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) e_x[g] = A.xproj[(((long)t * B + bq) * 2 + dir) * 4 * H + g * H + e_j];
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
      const unsigned long long* hx = S.xch + ((long)(dir * L + s - 1) * B) * (H / 4);
      // quiet wait: wave 0 looks at ONE word per producing workgroup (clip 0's four units of each of the H / 4 workgroups of this
      // direction) until all carry this launch's parity; the other waves park at the barrier.  (Every thread polling the 8 + 8
      // words it needs put 16 MB of L2-bypassing loads per round on the fabric: 7.8 us per step, slower than the launches.)
      if (w == 0) {
        bool ready = false;
        for (;;) {
          if (!ready) {
            ready = true;
            for (int p = l; p < H / 4; p += 64)
              ready = ready && ((__hip_atomic_load(hx + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x4000400040004000ull) == tagm);
          }
          if (__builtin_amdgcn_ballot_w64(!ready) == 0) break;
          __builtin_amdgcn_s_sleep(4);
          if (wall_clock64() - t0 > 200000000LL) break;        // (the loads below count the timeout)
        }
      }
      __syncthreads();
      for (int k0 = w * kq; k0 < (w + 1) * kq; k0 += 32 * KU) {
        unsigned long long alo[KU][MT], ahi[KU][MT];
        f32x4 blo[KU], bhi[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int k = k0 + u * 32 + kc;
          if (!w_resident) {
            blo[u] = *(const f32x4*)(W + wrow + k);
            bhi[u] = *(const f32x4*)(W + wrow + k + 4);
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int bb = mt * 16 + col;
            const unsigned long long* q = hx + (long)(bb < B ? bb : 0) * (H / 4) + (k >> 2);
            alo[u][mt] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ahi[u][mt] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          f16x8 bv;
          if (w_resident) bv = wv[u];
          else
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = (_Float16)blo[u][e]; bv[4 + e] = (_Float16)bhi[u][e]; }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int bb = mt * 16 + col;
            const unsigned long long* q = S.xch + ((long)(dir * L + s - 1) * B + (bb < B ? bb : 0)) * (H / 4) + ((k0 + u * 32 + kc) >> 2);
            while ((alo[u][mt] & 0x4000400040004000ull) != tagm || (ahi[u][mt] & 0x4000400040004000ull) != tagm) {     // not this launch's yet
              __builtin_amdgcn_s_sleep(2);
              if (wall_clock64() - t0 > 200000000LL) {          // 2 s of the 100 MHz wall clock: give up (results invalid, counted)
                __hip_atomic_fetch_add(&g_lstm_seq_timeouts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
              alo[u][mt] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ahi[u][mt] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned long long lo = alo[u][mt] & ~0x4000400040004000ull, hi = ahi[u][mt] & ~0x4000400040004000ull;
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            f16x8 av = __builtin_bit_cast(f16x8, (u64x2){lo, hi});
            if (bb >= B)
#pragma unroll
              for (int e = 0; e < 8; ++e) av[e] = (_Float16)0.f;
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc[mt], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[w][mt][l][r] = acc[mt][r];
    __syncthreads();
    float h_new = 0.f;
    if (e_own) {                                        // the statements of lstm_step_fwd_kernel's epilogue
      const int bb = e_bb, j = e_j;
      const int mt = bb >> 4, r16 = bb & 15, ln = (r16 >> 2) * 16 + (threadIdx.x & 3) * 4, rg = r16 & 3;
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v = red[0][mt][ln + g][rg] + red[1][mt][ln + g][rg] + red[2][mt][ln + g][rg] + red[3][mt][ln + g][rg];
        pre[g] = v + e_x[g] + e_bi[g] + e_bh[g];
      }
      const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
      const long st_new = ((long)(dir * (L + 1) + s + 1) * B + bb) * H + j;
      const float cp = e_cp, hp = e_hp;
      const float cn = fg * cp + ig * gg;
      const float hn = og * tanhf(cn);
      const bool valid = t < e_len;
      e_cp = valid ? cn : cp;
      e_hp = valid ? hn : hp;
      h_new = e_hp;
      // hand-off FIRST (round 6: it used to queue behind the bookkeeping stores below): the four units of a clip as ONE tagged 8-byte
      // write-through store (lanes 4q .. 4q + 3 hold them: a clip's four lanes take this branch together)
      if (s + 1 < L) {
        const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)h_new);
        const unsigned h1 = (unsigned)__shfl_down((int)hb, 1, 64), h2 = (unsigned)__shfl_down((int)hb, 2, 64), h3 = (unsigned)__shfl_down((int)hb, 3, 64);
        if ((threadIdx.x & 3) == 0) {
          const unsigned long long word = ((unsigned long long)hb | ((unsigned long long)h1 << 16) | ((unsigned long long)h2 << 32) | ((unsigned long long)h3 << 48)) | tagm;
          __hip_atomic_store(S.xch + ((long)(dir * L + s) * B + e_bb) * (H / 4) + (j0 >> 2), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      A.cseq[st_new] = e_cp;
      A.hseq[st_new] = e_hp;
      float* gs = A.gates + (((long)t * B + bb) * 2 + dir) * 4 * H + j;
      gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
      A.hprev_t[(((long)t * B + bb) * 2 + dir) * H + j] = hp;
      A.out[((long)bb * L + t) * 2 * H + dir * H + j] = valid ? hn : 0.f;
      if (A.qvec) {
        float* qv = A.qvec + (long)bb * 4 * H + dir * H + j;
        if (t == 0) qv[0] = valid ? hn : 0.f;
        if (t == e_len - 1) qv[2 * H] = hn;
      }
    }
    __syncthreads();                                    // red[] is free for the next step
  }
}

static int lstm_seq_capacity(const void* kernel) {
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, LSTM_THREADS, 0) != hipSuccess) return 0;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, kernel) != hipSuccess || fa.localSizeBytes > 0) return 0;
  return per_cu * cus;
}

extern "C" int64_t drn_lstm_seq_fwd_ws_bytes(int B, int L, int H) { return 64 + (int64_t)2 * L * B * (H / 4) * 8; }

extern "C" int drn_lstm_seq_fwd(const float* xproj, const void* Whh_f, const void* Whh_r, const float* b_ih_f, const float* b_hh_f,
                                const float* b_ih_r, const float* b_hh_r, float* hseq, float* cseq, float* gates, float* out, float* hprev_t,
                                float* qvec, void* xch_ws, int64_t ws_bytes, const int64_t* lengths, int B, int L, int H, void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(xproj && Whh_f && Whh_r && b_ih_f && b_hh_f && b_ih_r && b_hh_r && hseq && cseq && gates && out && hprev_t && lengths && xch_ws,
                "drn_lstm_seq_fwd: null pointer");
  DRN_CHECK_ARG(B > 0 && B <= 16 * MAX_BT && L > 0 && H % 128 == 0, "drn_lstm_seq_fwd: need B <= 64, H %% 128 == 0");
  DRN_CHECK_ARG(((uintptr_t)xch_ws & 63) == 0 && ws_bytes >= drn_lstm_seq_fwd_ws_bytes(B, L, H), "drn_lstm_seq_fwd: workspace too small or misaligned");
  LstmSeqArgs S;
  LstmFwdArgs& A = S.f;
  A.hseq16 = nullptr;
  A.xproj = xproj; A.Whh[0] = Whh_f; A.Whh[1] = Whh_r; A.hseq = hseq; A.cseq = cseq; A.gates = gates; A.out = out;
  A.hprev_t = hprev_t; A.qvec = qvec; A.b_ih[0] = b_ih_f; A.b_hh[0] = b_hh_f; A.b_ih[1] = b_ih_r; A.b_hh[1] = b_hh_r;
  A.lengths = (const long long*)lengths; A.B = B; A.L = L; A.H = H; A.s = 0;
  S.counter = (unsigned*)xch_ws;
  S.xch = (unsigned long long*)((char*)xch_ws + 64);
  const int mt = cdiv(B, 16), it = H / 4 / 32;          // iterations per wave
  const dim3 grid(H / 4, 2);
  const int total = (H / 4) * 2;
#define SEQ_CASE(MTV, KUV) do { \
    static int cap = 0; \
    if (!cap) cap = lstm_seq_capacity((const void*)lstm_seq_fwd_kernel<MTV, KUV>); \
    if (total > cap) { drn_set_error("drn_lstm_seq_fwd: %d workgroups exceed the %d the chip holds at once", total, cap); return DRN_ERR_UNSUPPORTED; } \
    lstm_seq_fwd_kernel<MTV, KUV><<<grid, LSTM_THREADS, 0, stream>>>(S); } while (0)
#define SEQ_MT(KUV) do { if (mt == 1) SEQ_CASE(1, KUV); else if (mt == 2) SEQ_CASE(2, KUV); else if (mt == 3) SEQ_CASE(3, KUV); else SEQ_CASE(4, KUV); } while (0)
  if (it % 4 == 0) SEQ_MT(4); else if (it % 2 == 0) SEQ_MT(2); else SEQ_MT(1);
#undef SEQ_MT
#undef SEQ_CASE
  return drn_launch_status("drn_lstm_seq_fwd");
}

extern "C" int drn_lstm_seq_fwd_timeouts(int reset) {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_lstm_seq_timeouts), sizeof(int)) != hipSuccess) return -1;
  if (reset && v) {
    const int z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_seq_timeouts), &z, sizeof(int));
  }
  return v;
}

// ------------------------------------------------------------------------------------------------ backward
struct LstmBwdArgs {
  const float* dout;     // [B][L][2H]
  const float* gates;    // activated i,f,g,o
  const float* cseq;
  const void* WhhT[2];   // [H][4H] = Whh^T (contiguous along the gate row index), fp32 or bf16
  float* dgates;         // [L][B][2][4H] by time
  float* dc;             // [2][B][H]
  float* dh_pass;        // [2][B][H]  dL/dh that bypasses the cell at padded positions (in/out)
  bf16_t* dgates16;      // optional: a bf16 copy of dgates, written by the cell backward and read by the bf16 step product (half the bytes)
  const float* dqvec;    // optional [B][4H]: gradient of the [first ; last] sentence vector, added to dout rows 0 and len_b-1 on load
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
  const long long len = A.lengths[bb];
  c.valid = t < len;
  const long sidx = ((long)dir * B + bb) * H + j;
  c.dout = A.dout[((long)bb * L + t) * 2 * H + dir * H + j];
  if (A.dqvec) {                        // (same order as the former in-place pass: row 0's half first, then row len-1's)
    const float* dq = A.dqvec + (long)bb * 4 * H + dir * H + j;
    const float d0 = dq[0], d1 = dq[2 * H];
    if (t == 0) c.dout += d0;
    if (t == len - 1) c.dout += d1;
  }
  c.dcn = s == L - 1 ? 0.f : A.dc[sidx];       // nothing flows in from beyond the last step
  const float* gs = A.gates + (((long)t * B + bb) * 2 + dir) * 4 * H + j;
  c.ig = gs[0]; c.fg = gs[H]; c.gg = gs[2 * H]; c.og = gs[3 * H];
  c.cn = A.cseq[((long)(dir * (L + 1) + s + 1) * B + bb) * H + j];
  c.cp = A.cseq[((long)(dir * (L + 1) + (s > 0 ? s : 1)) * B + bb) * H + j];
  if (s == 0) c.cp = 0.f;
  return c;
}
__device__ __forceinline__ void lstm_cell_bwd_apply(const LstmBwdArgs& A, const LstmCellIn& c, int dir, int bb, int j, int s, float dh_in) {
  // No FMA contraction in here: this function is inlined into TWO kernels (the first backward step and the epilogue of the step
  // kernel), and which of them handles a given time step depends on how far the batch was padded (a padded query adds all-invalid
  // steps in front).  With contraction left to the compiler the two copies once came out with different fused pairs -- gate
  // gradients one ulp apart between the padded (hipGraph) and the unpadded (eager) trainer.
#pragma clang fp contract(off)
  const int B = A.B, L = A.L, H = A.H;
  const int t = dir == 0 ? s : L - 1 - s;
  const long sidx = ((long)dir * B + bb) * H + j;
  const long gidx = (((long)t * B + bb) * 2 + dir) * 4 * H + j;
  float* dg = A.dgates + gidx;
  const float dh = dh_in + (c.valid ? c.dout : 0.f);
  const float dcn = c.dcn;
  float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
  if (c.valid) {
    const float ig = c.ig, fg = c.fg, gg = c.gg, og = c.og, cn = c.cn, cp = c.cp;
    const float tc = tanhf(cn);
    const float dcv = dcn + dh * og * (1.f - tc * tc);
    d0 = dcv * gg * ig * (1.f - ig);
    d1 = dcv * cp * fg * (1.f - fg);
    d2 = dcv * ig * (1.f - gg * gg);
    d3 = dh * tc * og * (1.f - og);
    A.dc[sidx] = dcv * fg;
    A.dh_pass[sidx] = 0.f;
  } else {
    A.dc[sidx] = dcn;
    A.dh_pass[sidx] = dh;
  }
  dg[0] = d0; dg[H] = d1; dg[2 * H] = d2; dg[3 * H] = d3;
  if (A.dgates16) {
    bf16_t* dg16 = A.dgates16 + gidx;
    dg16[0] = (bf16_t)d0; dg16[H] = (bf16_t)d1; dg16[2 * H] = (bf16_t)d2; dg16[3 * H] = (bf16_t)d3;
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
// KU iterations prefetched at a time.  WT = float: exact-fp32 MFMA, 16 r per iteration; WT = bf16_t: the bf16 copy of Whh^T,
// dgates rounded to bf16 as they are loaded, v_mfma_f32_16x16x32_bf16 with fp32 accumulation, 32 r per iteration -- half the
// weight bytes through the L1 and 1/16 of the MFMA cycles (the fp32 step is 9.8 us hot and alone, scripts/bench_nodes.py).
template <typename WT, int KU>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_step_bwd_kernel(const LstmBwdArgs A) {
  __shared__ float red[4][64][4];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int dir = blockIdx.y, k0 = blockIdx.x * 16, bt = blockIdx.z;
  const int B = A.B, L = A.L, H = A.H, s = A.s;
  const int K = 4 * H, kq = K / 4;
  const int t = dir == 0 ? s : L - 1 - s;
  const float* dg = A.dgates + ((long)t * B * 2 + dir) * K;      // row bb at dg + bb * 2K
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int row = l & 15;
  const int bb_a = bt * 16 + row;
  // operands of the cell backward this thread will run in the epilogue: requested up front
  const int e_bb = bt * 16 + (l >> 4) * 4 + w, e_k = k0 + (l & 15);
  const bool e_own = e_bb < B;
  const LstmCellIn e_c = lstm_cell_bwd_load(A, dir, e_own ? e_bb : 0, e_k, s - 1);
  const float e_dhp = A.dh_pass[((long)dir * B + (e_own ? e_bb : 0)) * H + e_k];
  const float* arow = dg + (long)(bb_a < B ? bb_a : 0) * 2 * K;
  if constexpr (std::is_same<WT, float>::value) {
    const float* WT_ = (const float*)A.WhhT[dir];
    const int rc = (l >> 4) * 4;
    for (int r0 = w * kq; r0 < (w + 1) * kq; r0 += 16 * KU) {
      f32x4 a[KU], b[KU];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int r = r0 + u * 16 + rc;
        a[u] = *(const f32x4*)(arow + r);
        if (bb_a >= B) a[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        b[u] = *(const f32x4*)(WT_ + (long)(k0 + row) * K + r);
      }
#pragma unroll
      for (int u = 0; u < KU; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][e], acc, 0, 0, 0);
    }
  } else {
    const bf16_t* WT_ = (const bf16_t*)A.WhhT[dir];
    const int rc = (l >> 4) * 8;
    if (A.dgates16) {                   // (uniform) gate gradients from their bf16 copy: 16 bytes per lane and K step instead of 32
      const bf16_t* arow16 = A.dgates16 + ((long)t * B * 2 + dir) * K + (long)(bb_a < B ? bb_a : 0) * 2 * K;
      for (int r0 = w * kq; r0 < (w + 1) * kq; r0 += 32 * KU) {
        bf16x8 a[KU], b[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int r = r0 + u * 32 + rc;
          a[u] = *(const bf16x8*)(arow16 + r);
          if (bb_a >= B)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[u][e] = (bf16_t)0.f;
          b[u] = *(const bf16x8*)(WT_ + (long)(k0 + row) * K + r);
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[u], acc, 0, 0, 0);
      }
    } else
    for (int r0 = w * kq; r0 < (w + 1) * kq; r0 += 32 * KU) {
      f32x4 alo[KU], ahi[KU];
      bf16x8 b[KU];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int r = r0 + u * 32 + rc;
        alo[u] = *(const f32x4*)(arow + r);
        ahi[u] = *(const f32x4*)(arow + r + 4);
        if (bb_a >= B) { alo[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; ahi[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        b[u] = *(const bf16x8*)(WT_ + (long)(k0 + row) * K + r);
      }
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        bf16x8 av;
#pragma unroll
        for (int e = 0; e < 4; ++e) { av[e] = (bf16_t)alo[u][e]; av[4 + e] = (bf16_t)ahi[u][e]; }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u], acc, 0, 0, 0);
      }
    }
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
                                  const float* dqvec, void* dgates16, const int64_t* lengths, int B, int L, int H, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(dout && gates && cseq && dgates && dc && dh_pass && lengths, "drn_lstm_bwd_first: null pointer");
  DRN_CHECK_ARG(B > 0 && B <= 16 * MAX_BT && L > 0 && H % 64 == 0, "drn_lstm_bwd_first: need B<=64, H%%64==0");
  LstmBwdArgs A;
  memset(&A, 0, sizeof(A));
  A.dout = dout; A.gates = gates; A.cseq = cseq; A.dgates = dgates; A.dc = dc; A.dh_pass = dh_pass; A.dqvec = dqvec;
  A.dgates16 = (bf16_t*)dgates16;
  A.lengths = (const long long*)lengths; A.B = B; A.L = L; A.H = H; A.s = L - 1;
  lstm_bwd_first_kernel<<<cdiv(2 * B * H, 256), 256, 0, (hipStream_t)stream>>>(A);
  return drn_launch_status("drn_lstm_bwd_first");
}

extern "C" int drn_lstm_step_bwd(const float* dout, const float* gates, const float* cseq, const void* WhhT_f, const void* WhhT_r,
                                 int w_dtype, float* dgates, float* dc, float* dh_pass, const float* dqvec, void* dgates16, const int64_t* lengths,
                                 int B, int L, int H, int s, void* stream_) {
  drn_clear_status();
  hipStream_t stream = (hipStream_t)stream_;
  DRN_CHECK_ARG(dout && gates && cseq && WhhT_f && WhhT_r && dgates && dc && dh_pass && lengths, "drn_lstm_step_bwd: null pointer");
  DRN_CHECK_ARG(B > 0 && B <= 16 * MAX_BT && L > 0 && H % 64 == 0 && s >= 1 && s < L, "drn_lstm_step_bwd: need B<=64, H%%64==0, 1<=s<L");
  LstmBwdArgs A;
  memset(&A, 0, sizeof(A));
  A.dout = dout; A.gates = gates; A.cseq = cseq; A.WhhT[0] = WhhT_f; A.WhhT[1] = WhhT_r; A.dgates = dgates; A.dc = dc;
  A.dh_pass = dh_pass; A.dqvec = dqvec; A.dgates16 = (bf16_t*)dgates16;
  A.lengths = (const long long*)lengths; A.B = B; A.L = L; A.H = H; A.s = s;
  DRN_CHECK_ARG(w_dtype == DRN_F32 || (w_dtype == DRN_BF16 && H % 128 == 0), "drn_lstm_step_bwd: bf16 weights need H %% 128 == 0");
  dim3 grid(H / 16, 2, cdiv(B, 16));
  if (w_dtype == DRN_BF16) {
    const int it = H / 32;               // 32-wide K iterations per wave (K = 4H over 4 waves)
    if (it % 8 == 0) lstm_step_bwd_kernel<bf16_t, 8><<<grid, LSTM_THREADS, 0, stream>>>(A);
    else lstm_step_bwd_kernel<bf16_t, 4><<<grid, LSTM_THREADS, 0, stream>>>(A);
  } else {
    const int kq16 = H / 16;             // 16-wide K iterations per wave
    if (kq16 % 8 == 0) lstm_step_bwd_kernel<float, 8><<<grid, LSTM_THREADS, 0, stream>>>(A);
    else lstm_step_bwd_kernel<float, 4><<<grid, LSTM_THREADS, 0, stream>>>(A);
  }
  return drn_launch_status("drn_lstm_step_bwd");
}
