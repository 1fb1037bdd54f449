// FCOS-style target assignment and the three DRN losses, forward and backward, fused into two
// single-workgroup kernels (<= ~30k locations; everything is a few hundred KB).
//
// Reference: FCOSLossComputation (model/loss.py:40-239): compute_targets_for_locations (90-127),
// SigmoidFocalLoss (model/layers/sigmoid_focal_loss.py:40-52; the CUDA kernel of fcos_core._C is the same
// function in its numerically stable form, which is what is computed here), IOULoss
// (model/layers/iou_loss.py:6-24), segment_tiou + SmoothL1 on predictions with tIoU > 0.9 (loss.py:168-197).
// Quirks reproduced: divisor n_pos + B; the clamp at loss.py:180-181 hits LOCATION
// index 0 (level 0, t = 0) of every clip; the SmoothL1 target carries gradient back into the box
// regression; ties in min/max split their gradient evenly (torch's elementwise min/max rule).
// No host synchronisation: positive counts stay on the device (out[3], out[4]) and the backward
// kernel reads them there.
#include "common.h"
#include "../../include/drn_hip.h"

#define LOSS_THREADS 1024

// Five block-wide sums at once (blockDim.x <= 1024, a multiple of 64): the same wave trees and the same cross-wave tree as five
// block_sum() calls -- bit-identical results -- behind 2 barriers instead of 15.  sh: >= 5 * 17 floats.
__device__ __forceinline__ void block_sum5(float (&v)[5], float* sh) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; ++k) v[k] = wave_sum(v[k]);
  __syncthreads();
  if (l == 0)
#pragma unroll
    for (int k = 0; k < 5; ++k) sh[k * 17 + w] = v[k];
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float t = l < nw ? sh[k * 17 + l] : 0.f;
      t = wave_sum(t);
      if (l == 0) sh[k * 17 + 16] = t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 5; ++k) v[k] = sh[k * 17 + 16];
}

struct LossParams {
  int nlevels, B, total_rows;
  int row_start[DRN_MAX_GROUPS], L[DRN_MAX_GROUPS];
  float stride[DRN_MAX_GROUPS], lo[DRN_MAX_GROUPS], hi[DRN_MAX_GROUPS];
  float gamma, alpha, target_scale;  // 2.0, 0.25, 32
  int iou_stage;                     // 0: first stage (no IoU-score loss)
  int gt_f64;                        // ground truth arrives as float64 (the reference casts with .float(), main_model.py:74)
};
// counters (BatchNorm num_batches_tracked, int64) the forward pass owes an increment: applied by the loss's final kernel
// instead of a launch of their own
struct LossBumps {
  long long* ptr[DRN_LOSS_MAX_BUMPS];
  int inc[DRN_LOSS_MAX_BUMPS];
  int n;
};
__device__ __forceinline__ float load_gt(const LossParams& P, const void* gt, int i) {
  return P.gt_f64 ? (float)((const double*)gt)[i] : ((const float*)gt)[i];
}

struct Loc {
  int level, b, t;
  float loc, lo, hi;      // centre of the location; size bounds of its level (loss.py:103-110)
};
// The level's fields are picked with selects over STATIC indices: indexing the kernel-argument arrays with the per-lane level made
// the compiler walk the distinct levels of a wave one by one, a scalar load and a wait each time (most of this kernel's 8 us).
__device__ __forceinline__ Loc locate(const LossParams& P, int r) {
  int g = 0, rs = P.row_start[0], L = P.L[0];
  float st = P.stride[0], lo = P.lo[0], hi = P.hi[0];
#pragma unroll
  for (int i = 1; i < DRN_MAX_GROUPS; ++i) {
    const bool in = i < P.nlevels && r >= P.row_start[i];
    g = in ? i : g;
    rs = in ? P.row_start[i] : rs;
    L = in ? P.L[i] : L;
    st = in ? P.stride[i] : st;
    lo = in ? P.lo[i] : lo;
    hi = in ? P.hi[i] : hi;
  }
  Loc o;
  o.level = g;
  const int m = r - rs;
  o.b = m / L;
  o.t = m - o.b * L;
  o.loc = (float)o.t * st + st * 0.5f;  // model/fcos.py:204-211
  o.lo = lo;
  o.hi = hi;
  return o;
}

__device__ __forceinline__ float softplus(float x) { return x > 0.f ? x + log1pf(expf(-x)) : log1pf(expf(x)); }

struct IouTerm {
  bool masked;
  float tiou, p, d;
  float dt_ds, dt_de;  // d tIoU / d pred_start, d pred_end (already including the clamp mask)
};
__device__ __forceinline__ IouTerm iou_term(const LossParams& P, const Loc& q, float r0, float r1, float gs, float ge, float iou_logit) {
  IouTerm o;
  float ps = (q.loc - r0) / P.target_scale, pe = (q.loc + r1) / P.target_scale;
  float ms = 1.f, me = 1.f;
  if (q.level == 0 && q.t == 0) {  // loss.py:180-181
    ms = (ps >= 0.f && ps <= 1.f) ? 1.f : 0.f;
    me = (pe >= 0.f && pe <= 1.f) ? 1.f : 0.f;
    ps = fminf(fmaxf(ps, 0.f), 1.f);
    pe = fminf(fmaxf(pe, 0.f), 1.f);
  }
  const float imax = fminf(pe, ge), imin = fmaxf(ps, gs);
  const float umax = fmaxf(pe, ge), umin = fminf(ps, gs);
  const float inter = fmaxf(imax - imin, 0.f), uni = fmaxf(umax - umin, 0.f);
  o.tiou = inter / (uni + 1e-6f);
  o.masked = o.tiou > 0.9f;
  o.p = 1.f / (1.f + expf(-iou_logit));
  o.d = o.p - o.tiou;
  // gradients of tIoU (only used where masked: inter > 0 and union > 0 there)
  const float di_de = (imax - imin >= 0.f) ? (pe < ge ? 1.f : (pe == ge ? 0.5f : 0.f)) : 0.f;
  const float di_ds = (imax - imin >= 0.f) ? -(ps > gs ? 1.f : (ps == gs ? 0.5f : 0.f)) : 0.f;
  const float du_de = (umax - umin >= 0.f) ? (pe > ge ? 1.f : (pe == ge ? 0.5f : 0.f)) : 0.f;
  const float du_ds = (umax - umin >= 0.f) ? -(ps < gs ? 1.f : (ps == gs ? 0.5f : 0.f)) : 0.f;
  const float den = uni + 1e-6f;
  o.dt_de = me * (di_de * den - inter * du_de) / (den * den);
  o.dt_ds = ms * (di_ds * den - inter * du_ds) / (den * den);
  return o;
}

__device__ __forceinline__ bool assign_label(const LossParams& P, const Loc& q, float gs, float ge, float& tl, float& tr) {
  tl = q.loc - gs * P.target_scale;   // loss.py:98-101
  tr = ge * P.target_scale - q.loc;
  const float mn = fminf(tl, tr), mx = fmaxf(tl, tr);
  return mn > 0.f && mx >= q.lo && mx <= q.hi;
}

// The final summation (one workgroup of 256 threads, fixed order): out[0..2] = loss_cls, loss_reg, loss_iou ; out[3] = n_pos ;
// out[4] = n_iou_pos ; out[5] = their sum (main.py:225); applies the counter bumps.  coherent: the partials were published
// write-through by other workgroups of the SAME launch -- read them past the caches.
__device__ __forceinline__ void loss_finalize(const float* __restrict__ partial, int nblk, int B, float* __restrict__ out, const LossBumps& U,
                                              float* sh, bool coherent) {
  if ((int)threadIdx.x < U.n) *U.ptr[threadIdx.x] += U.inc[threadIdx.x];
  float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < nblk; b += blockDim.x)
#pragma unroll
    for (int k = 0; k < 5; ++k)
      v[k] += coherent ? __hip_atomic_load(partial + (long)b * 5 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : partial[(long)b * 5 + k];
  block_sum5(v, sh);
  if (threadIdx.x == 0) {
    out[0] = v[0] / (v[3] + (float)B);               // loss.py:213
    out[1] = v[3] > 0.f ? v[1] / v[3] : 0.f;          // loss.py:219-231
    out[2] = v[4] > 0.f ? v[2] / v[4] : 0.f;          // loss.py:194-197
    out[3] = v[3];
    out[4] = v[4];
    out[5] = out[0] + out[1] + out[2];
  }
}

// Phase 1: one location per thread, per-block partial sums partial[blk][5] = {focal, iou-loss, smooth-l1, n_pos, n_iou}.
__global__ __launch_bounds__(256) void fcos_loss_fwd_partial_kernel(const LossParams P, const float* __restrict__ logits,
                                                                    const float* __restrict__ reg, const float* __restrict__ iou,
                                                                    const void* __restrict__ gt, float* __restrict__ partial,
                                                                    float* __restrict__ labels, int* __restrict__ ticket,
                                                                    float* __restrict__ out, const LossBumps U) {
  __shared__ float sh[5 * 17];
  __shared__ int s_last;
  float s_focal = 0.f, s_ioul = 0.f, s_sl1 = 0.f, n_pos = 0.f, n_iou = 0.f;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < P.total_rows) {
    // every operand is requested HERE, before the transcendental math: left where they are first used, the loads of reg / iou
    // came after ~1000 instructions of powf / log1pf each behind its own wait -- three serial memory trips in an 8 us kernel
    const float x = logits[r];
    const float pl = reg[r * 2], prr = reg[r * 2 + 1];
    const float iou_x = P.iou_stage ? iou[r] : 0.f;
    const Loc q = locate(P, r);
    const float gs = load_gt(P, gt, q.b * 2), ge = load_gt(P, gt, q.b * 2 + 1);
    float tl, tr;
    const bool pos = assign_label(P, q, gs, ge, tl, tr);
    if (labels) labels[r] = pos ? 1.f : 0.f;
    const float p = 1.f / (1.f + expf(-x));
    // -log(p) = softplus(-x), -log(1-p) = softplus(x)
    if (pos) s_focal = P.alpha * powf(1.f - p, P.gamma) * softplus(-x);
    else s_focal = (1.f - P.alpha) * powf(p, P.gamma) * softplus(x);
    if (pos) {
      n_pos = 1.f;
      const float inter = fminf(prr, tr) + fminf(pl, tl);
      const float uni = (tl + tr) + (pl + prr) - inter;
      s_ioul = -logf((inter + 1e-8f) / (uni + 1e-8f));
    }
    if (P.iou_stage) {
      const IouTerm it = iou_term(P, q, pl, prr, gs, ge, iou_x);
      if (it.masked) {
        n_iou = 1.f;
        const float ad = fabsf(it.d);
        s_sl1 = ad < 1.f ? 0.5f * it.d * it.d : ad - 0.5f;
      }
    }
  }
  float v5s[5] = {s_focal, s_ioul, s_sl1, n_pos, n_iou};
  block_sum5(v5s, sh);
  s_focal = v5s[0]; s_ioul = v5s[1]; s_sl1 = v5s[2]; n_pos = v5s[3]; n_iou = v5s[4];
  if (threadIdx.x == 0) {
    float* p = partial + (long)blockIdx.x * 5;
    if (!ticket) {
      p[0] = s_focal; p[1] = s_ioul; p[2] = s_sl1; p[3] = n_pos; p[4] = n_iou;
    } else {
      // one launch: the partial sums are published write-through, the workgroup that arrives LAST adds them all in the order the
      // separate final kernel uses (same values whoever is last) and writes the losses
      const float v5[5] = {s_focal, s_ioul, s_sl1, n_pos, n_iou};
#pragma unroll
      for (int k = 0; k < 5; ++k) __hip_atomic_store(p + k, v5[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // confirmed by a returning read-modify-write per address before the ticket (skinny_group_kernel, qdense.hip, says why)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        unsigned back;
        asm volatile("global_atomic_or %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(back) : "v"(p + k), "v"(0u) : "memory");
      }
      const int prev = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = prev == (int)gridDim.x - 1;
      if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // re-armed
    }
  }
  if (!ticket) return;
  __syncthreads();
  if (!s_last) return;
  loss_finalize(partial, gridDim.x, P.B, out, U, sh, true);
}

// Phase 2 (one workgroup, fixed order): out[0..2] = loss_cls, loss_reg, loss_iou ; out[3] = n_pos ; out[4] = n_iou_pos ;
// out[5] = their sum (main.py:225)
__global__ __launch_bounds__(256) void fcos_loss_fwd_final_kernel(const float* __restrict__ partial, int nblk, int B, float* __restrict__ out,
                                                                  const LossBumps U) {
  __shared__ float sh[5 * 17];
  loss_finalize(partial, nblk, B, out, U, sh, false);
}

// g_cls / g_reg / g_iou: upstream gradients (one float each, NULL = 0) of (loss_cls, loss_reg, loss_iou).  Outputs: dlogits[r], dreg[r][2], diou[r].
__global__ __launch_bounds__(256) void fcos_loss_bwd_kernel(const LossParams P, const float* __restrict__ logits,
                                                                     const float* __restrict__ reg, const float* __restrict__ iou,
                                                                     const void* __restrict__ gt, const float* __restrict__ fwd_out,
                                                                     const float* __restrict__ g_cls, const float* __restrict__ g_reg,
                                                                     const float* __restrict__ g_iou, float* __restrict__ dlogits,
                                                                     float* __restrict__ dreg, float* __restrict__ diou) {
  const float n_pos = fwd_out[3], n_iou = fwd_out[4];
  const float k_cls = (g_cls ? g_cls[0] : 0.f) / (n_pos + (float)P.B);
  const float k_reg = n_pos > 0.f ? (g_reg ? g_reg[0] : 0.f) / n_pos : 0.f;
  const float k_iou = (P.iou_stage && n_iou > 0.f) ? (g_iou ? g_iou[0] : 0.f) / n_iou : 0.f;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < P.total_rows; r += gridDim.x * blockDim.x) {
    const float x = logits[r];                         // (all loads first: see the forward kernel)
    const float pl = reg[r * 2], prr = reg[r * 2 + 1];
    const float iou_x = P.iou_stage ? iou[r] : 0.f;
    const Loc q = locate(P, r);
    const float gs = load_gt(P, gt, q.b * 2), ge = load_gt(P, gt, q.b * 2 + 1);
    float tl, tr;
    const bool pos = assign_label(P, q, gs, ge, tl, tr);
    const float p = 1.f / (1.f + expf(-x));
    float dx;
    if (pos) {
      // d/dx [ alpha (1-p)^g softplus(-x) ] = alpha (1-p)^g [ -g p softplus(-x) - (1-p) ]   
      const float om = 1.f - p;
      dx = P.alpha * powf(om, P.gamma) * (-P.gamma * p * softplus(-x) - om);
    } else {
      dx = (1.f - P.alpha) * powf(p, P.gamma) * (P.gamma * (1.f - p) * softplus(x) + p);
    }
    dlogits[r] = dx * k_cls;
    float d0 = 0.f, d1 = 0.f;  // gradient w.r.t. reg[r][0..1]
    if (pos && k_reg != 0.f) {
      const float inter = fminf(prr, tr) + fminf(pl, tl);
      const float uni = (tl + tr) + (pl + prr) - inter;
      const float dil = pl < tl ? 1.f : (pl == tl ? 0.5f : 0.f);
      const float dir = prr < tr ? 1.f : (prr == tr ? 0.5f : 0.f);
      d0 = k_reg * (-dil / (inter + 1e-8f) + (1.f - dil) / (uni + 1e-8f));
      d1 = k_reg * (-dir / (inter + 1e-8f) + (1.f - dir) / (uni + 1e-8f));
    }
    if (P.iou_stage) {
      float di = 0.f;
      const IouTerm it = iou_term(P, q, pl, prr, gs, ge, iou_x);
      if (it.masked && k_iou != 0.f) {
        const float sl = fabsf(it.d) < 1.f ? it.d : (it.d > 0.f ? 1.f : -1.f);
        di = k_iou * sl * it.p * (1.f - it.p);
        const float dt = -k_iou * sl;                 // the target carries gradient (A.3 #2)
        d0 += dt * it.dt_ds * (-1.f / P.target_scale);
        d1 += dt * it.dt_de * (1.f / P.target_scale);
      }
      diou[r] = di;
    }
    dreg[r * 2] = d0;
    dreg[r * 2 + 1] = d1;
  }
}

static int fill_loss_params(LossParams& P, const DrnLossLevel* levels, int nlevels, int B, float gamma, float alpha, float target_scale,
                            int iou_stage, int gt_f64, const char* who) {
  DRN_CHECK_ARG(levels && nlevels >= 1 && nlevels <= DRN_MAX_GROUPS && B > 0, "%s: bad level table", who);
  memset(&P, 0, sizeof(P));
  P.nlevels = nlevels; P.B = B;
  int rows = 0;
  for (int l = 0; l < nlevels; ++l) {
    DRN_CHECK_ARG(levels[l].L > 0, "%s: level %d has no locations", who, l);
    P.row_start[l] = rows; P.L[l] = levels[l].L; P.stride[l] = levels[l].stride; P.lo[l] = levels[l].lo; P.hi[l] = levels[l].hi;
    rows += B * levels[l].L;
  }
  P.total_rows = rows;
  P.gamma = gamma; P.alpha = alpha; P.target_scale = target_scale; P.iou_stage = iou_stage; P.gt_f64 = gt_f64;
  return DRN_OK;
}

extern "C" int drn_fcos_loss_fwd(const DrnLossLevel* levels, int nlevels, int B, const float* logits, const float* reg, const float* iou,
                                 const void* gt, int gt_f64, float gamma, float alpha, float target_scale, int iou_stage, float* out6,
                                 float* labels, float* ws, int32_t* ticket, const DrnCounterBump* bumps, int nbumps, void* stream) {
  drn_clear_status();
  LossParams P;
  int rc = fill_loss_params(P, levels, nlevels, B, gamma, alpha, target_scale, iou_stage, gt_f64, "drn_fcos_loss_fwd");
  if (rc) return rc;
  DRN_CHECK_ARG(logits && reg && gt && out6 && (!iou_stage || iou), "drn_fcos_loss_fwd: null pointer");
  DRN_CHECK_ARG(nbumps >= 0 && nbumps <= DRN_LOSS_MAX_BUMPS && (nbumps == 0 || bumps), "drn_fcos_loss_fwd: at most %d counter bumps",
                DRN_LOSS_MAX_BUMPS);
  const int nblk = cdiv(P.total_rows, 256);
  DRN_CHECK_ARG(ws, "drn_fcos_loss_fwd: workspace (5*ceil(R/256) floats) required");
  LossBumps U;
  memset(&U, 0, sizeof(U));
  U.n = nbumps;
  for (int i = 0; i < nbumps; ++i) { U.ptr[i] = (long long*)bumps[i].counter; U.inc[i] = bumps[i].inc; }
  fcos_loss_fwd_partial_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(P, logits, reg, iou, gt, ws, labels, ticket, out6, U);
  if (!ticket) fcos_loss_fwd_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(ws, nblk, B, out6, U);
  return drn_launch_status("drn_fcos_loss_fwd");
}

extern "C" int drn_fcos_loss_bwd(const DrnLossLevel* levels, int nlevels, int B, const float* logits, const float* reg,
                                 const float* iou, const void* gt, int gt_f64, float gamma, float alpha, float target_scale, int iou_stage,
                                 const float* fwd_out5, const float* g_cls, const float* g_reg, const float* g_iou, float* dlogits,
                                 float* dreg, float* diou, void* stream) {
  drn_clear_status();
  LossParams P;
  int rc = fill_loss_params(P, levels, nlevels, B, gamma, alpha, target_scale, iou_stage, gt_f64, "drn_fcos_loss_bwd");
  if (rc) return rc;
  DRN_CHECK_ARG(logits && reg && gt && fwd_out5 && dlogits && dreg && (!iou_stage || (iou && diou)),
                "drn_fcos_loss_bwd: null pointer");
  fcos_loss_bwd_kernel<<<cdiv(P.total_rows, 256), 256, 0, (hipStream_t)stream>>>(P, logits, reg, iou, gt, fwd_out5, g_cls, g_reg, g_iou, dlogits,
                                                                                 dreg, diou);
  return drn_launch_status("drn_fcos_loss_bwd");
}

// ---------------------------------------------------------------------------------------------------------------------
// Class-general sigmoid focal loss with the call shape of the reference's only FFI, fcos_core._C.sigmoid_focalloss_forward /
// _backward (model/layers/sigmoid_focal_loss.py:18-20,31-33): per-element losses over logits (N, C) with int32 class
// targets (0 = background, c in 1..C = foreground class c, negative = ignored).  Same function as the in-repo formula
// (sigmoid_focal_loss.py:40-52), written through softplus so that |logit| > 88 stays finite: -log p = softplus(-x),
// -log(1-p) = softplus(x).
__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ logits, const int* __restrict__ targets, long total,
                                                        int C, float gamma, float alpha, float* __restrict__ losses) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / C;
    const int cls = (int)(i - n * C) + 1;
    const int t = targets[n];
    const float x = logits[i];
    const float p = 1.f / (1.f + expf(-x));
    float v = 0.f;
    if (t == cls) v = alpha * powf(1.f - p, gamma) * softplus(-x);
    else if (t >= 0) v = (1.f - alpha) * powf(p, gamma) * softplus(x);
    losses[i] = v;
  }
}

__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                                        const float* __restrict__ d_losses, long total, int C, float gamma, float alpha,
                                                        float* __restrict__ d_logits) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / C;
    const int cls = (int)(i - n * C) + 1;
    const int t = targets[n];
    const float x = logits[i];
    const float p = 1.f / (1.f + expf(-x));
    float dx = 0.f;
    if (t == cls) {
      const float om = 1.f - p;
      dx = alpha * powf(om, gamma) * (-gamma * p * softplus(-x) - om);
    } else if (t >= 0) {
      dx = (1.f - alpha) * powf(p, gamma) * (gamma * (1.f - p) * softplus(x) + p);
    }
    d_logits[i] = dx * d_losses[i];
  }
}

extern "C" int drn_focal_fwd(const float* logits, const int32_t* targets, int64_t N, int num_classes, float gamma, float alpha,
                             float* losses, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(N >= 0 && num_classes >= 1, "drn_focal_fwd: bad shape N=%lld C=%d", (long long)N, num_classes);
  if (N == 0) return DRN_OK;
  DRN_CHECK_ARG(logits && targets && losses, "drn_focal_fwd: null pointer");
  const long total = (long)N * num_classes;
  const int nblk = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  focal_fwd_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(logits, targets, total, num_classes, gamma, alpha, losses);
  return drn_launch_status("drn_focal_fwd");
}

extern "C" int drn_focal_bwd(const float* logits, const int32_t* targets, const float* d_losses, int64_t N, int num_classes, float gamma,
                             float alpha, float* d_logits, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(N >= 0 && num_classes >= 1, "drn_focal_bwd: bad shape N=%lld C=%d", (long long)N, num_classes);
  if (N == 0) return DRN_OK;
  DRN_CHECK_ARG(logits && targets && d_losses && d_logits, "drn_focal_bwd: null pointer");
  const long total = (long)N * num_classes;
  const int nblk = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  focal_bwd_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(logits, targets, d_losses, total, num_classes, gamma, alpha, d_logits);
  return drn_launch_status("drn_focal_bwd");
}

// ---------------------------------------------------------------------------------------------------------------------
// Stand-alone IOULoss (model/layers/iou_loss.py:5-24; the module the reference exports from model.layers next to the focal loss):
// per row i of pred / target (N, 2) = (left, right) distances,
//   inter = min(pr, tr) + min(pl, tl), union = (tl + tr) + (pl + pr) - inter, l_i = -log((inter + 1e-8) / (union + 1e-8));
//   weight given and sum(weight) > 0: sum(l_i w_i) / sum(w), else mean(l_i).
// One workgroup, fixed summation order (deterministic); the decision "weighted or not" stays on the device: out2 = {loss, the
// divisor used WITH its sign as the mode flag (> 0: sum of weights, < 0: -(N) = plain mean)}; backward reads it there.
// Gradients flow to pred and target (ties in min split evenly: torch's elementwise rule); the weight is treated as a constant.
__global__ __launch_bounds__(LOSS_THREADS) void iou_loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                                    const float* __restrict__ weight, long N, float* __restrict__ out2) {
  __shared__ float sh[5 * 17];
  float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};       // sum l, sum l*w, sum w
  for (long i = threadIdx.x; i < N; i += LOSS_THREADS) {
    const float pl = pred[2 * i], pr = pred[2 * i + 1], tl = target[2 * i], tr = target[2 * i + 1];
    const float inter = fminf(pr, tr) + fminf(pl, tl);
    const float uni = (tl + tr) + (pl + pr) - inter;
    const float l = -logf((inter + 1e-8f) / (uni + 1e-8f));
    const float w = weight ? weight[i] : 0.f;
    v[0] += l;
    v[1] += l * w;
    v[2] += w;
  }
  block_sum5(v, sh);
  if (threadIdx.x == 0) {
    const bool weighted = weight != nullptr && v[2] > 0.f;
    out2[0] = weighted ? v[1] / v[2] : v[0] / (float)N;
    out2[1] = weighted ? v[2] : -(float)N;
  }
}

__global__ __launch_bounds__(256) void iou_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           const float* __restrict__ weight, long N, const float* __restrict__ out2,
                                                           const float* __restrict__ gout, float* __restrict__ dpred,
                                                           float* __restrict__ dtarget) {
  const float div = out2[1], g = gout ? gout[0] : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const float pl = pred[2 * i], pr = pred[2 * i + 1], tl = target[2 * i], tr = target[2 * i + 1];
    const float inter = fminf(pr, tr) + fminf(pl, tl);
    const float uni = (tl + tr) + (pl + pr) - inter;
    const float coef = g * (div > 0.f ? weight[i] / div : 1.f / -div);
    // l = -log(inter + eps) + log(union + eps)
    const float di = coef * (-1.f / (inter + 1e-8f) - 1.f / (uni + 1e-8f)), du = coef / (uni + 1e-8f);
    const float ml_p = pl < tl ? 1.f : (pl == tl ? 0.5f : 0.f), mr_p = pr < tr ? 1.f : (pr == tr ? 0.5f : 0.f);
    if (dpred) {
      dpred[2 * i] = di * ml_p + du;
      dpred[2 * i + 1] = di * mr_p + du;
    }
    if (dtarget) {
      dtarget[2 * i] = di * (1.f - ml_p) + du;
      dtarget[2 * i + 1] = di * (1.f - mr_p) + du;
    }
  }
}

extern "C" int drn_iou_loss_fwd(const float* pred, const float* target, const float* weight, int64_t N, float* out2, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(N > 0, "drn_iou_loss_fwd: no rows (the reference asserts losses.numel() != 0, iou_loss.py:23)");
  DRN_CHECK_ARG(pred && target && out2, "drn_iou_loss_fwd: null pointer");
  iou_loss_fwd_kernel<<<1, LOSS_THREADS, 0, (hipStream_t)stream>>>(pred, target, weight, (long)N, out2);
  return drn_launch_status("drn_iou_loss_fwd");
}

extern "C" int drn_iou_loss_bwd(const float* pred, const float* target, const float* weight, int64_t N, const float* out2,
                                const float* gout, float* dpred, float* dtarget, void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(N > 0 && pred && target && out2 && (dpred || dtarget), "drn_iou_loss_bwd: bad args");
  const int nblk = (int)((N + 255) / 256 < 1024 ? (N + 255) / 256 : 1024);
  iou_loss_bwd_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(pred, target, weight, (long)N, out2, gout, dpred, dtarget);
  return drn_launch_status("drn_iou_loss_bwd");
}
