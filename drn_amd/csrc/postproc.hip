// Eval-time post-processor of the dense heads (model/inference.py:51-120,166-199), one workgroup per clip:
//   candidates  sigmoid(logit) > thr        (tested BEFORE the IoU-score product, inference.py:71-79)
//   score       sigmoid(logit) [* sigmoid(iou)]   (second / third stage)
//   per level   keep the top_n scores (torch.topk(sorted=False): any order -- kept here in location order)
//   decode      ((loc - reg0)/32, (loc + reg1)/32) clamped to [0,1], score = sqrt(score), location = loc/32
// Kept candidates of a clip are written level after level (the order select_over_all_levels concatenates), counts per
// (clip, level) tell the host how to slice them: ONE device->host copy per batch instead of the reference's
// nonzero / tolist round trips per clip and level.  reg = exp(.) > 0, so the reference's min_size = 0 filter never fires.
#include "common.h"
#include "../../include/drn_hip.h"

#define PP_THREADS 256
#define PP_MAX_L 2048

struct PostParams {
  int nlevels, B, rows_per_clip;
  int row_start[DRN_MAX_GROUPS], L[DRN_MAX_GROUPS];
  float stride[DRN_MAX_GROUPS];
  float thr, downsample;
  int top_n, use_iou;
};

__device__ __forceinline__ float sigmoid_pp(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(PP_THREADS) void postprocess_kernel(const PostParams P, const float* __restrict__ logits,
                                                                 const float* __restrict__ reg, const float* __restrict__ iou,
                                                                 float* __restrict__ det, float* __restrict__ scores,
                                                                 float* __restrict__ locs, int* __restrict__ counts) {
  __shared__ float sc[PP_MAX_L];            // score of candidates, -1 for the rest
  __shared__ unsigned char keep[PP_MAX_L];
  __shared__ int wsum[PP_THREADS / 64], s_total;
  const int b = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  int out_base = 0;                         // kept candidates of the previous levels of this clip
  for (int l = 0; l < P.nlevels; ++l) {
    const int L = P.L[l];
    const long r0 = P.row_start[l] + (long)b * L;
    for (int t = tid; t < L; t += PP_THREADS) {
      const float c = sigmoid_pp(logits[r0 + t]);
      float s = -1.f;
      if (c > P.thr) s = P.use_iou ? c * sigmoid_pp(iou[r0 + t]) : c;
      sc[t] = s;
    }
    __syncthreads();
    // how many candidates?  (block count through ballots)
    int n_c = 0;
    for (int t0 = 0; t0 < L; t0 += PP_THREADS) {
      const int t = t0 + tid;
      const unsigned long long m = __ballot(t < L && sc[t] >= 0.f);
      if (lane == 0) wsum[wv] = __popcll(m);
      __syncthreads();
      n_c += wsum[0] + wsum[1] + wsum[2] + wsum[3];
      __syncthreads();
    }
    // top_n by rank (ties: earlier location first) when there are more candidates than that
    for (int t = tid; t < L; t += PP_THREADS) {
      const float s = sc[t];
      bool k = s >= 0.f;
      if (k && n_c > P.top_n) {
        int rank = 0;
        for (int j = 0; j < L; ++j) {
          const float o = sc[j];
          rank += (o > s) || (o == s && j < t);
        }
        k = rank < P.top_n;
      }
      keep[t] = k;
    }
    __syncthreads();
    // ordered compaction
    int level_kept = 0;
    for (int t0 = 0; t0 < L; t0 += PP_THREADS) {
      const int t = t0 + tid;
      const bool k = t < L && keep[t];
      const unsigned long long m = __ballot(k);
      if (lane == 0) wsum[wv] = __popcll(m);
      __syncthreads();
      int before = __popcll(m & ((1ull << lane) - 1ull));
      for (int q = 0; q < wv; ++q) before += wsum[q];
      const int chunk = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      if (k) {
        const long o = (long)b * P.rows_per_clip + out_base + level_kept + before;
        const float loc = (float)t * P.stride[l] + P.stride[l] * 0.5f;      // model/fcos.py:204-211
        const float d0 = (loc - reg[(r0 + t) * 2 + 0]) / P.downsample, d1 = (loc + reg[(r0 + t) * 2 + 1]) / P.downsample;
        det[o * 2 + 0] = fminf(fmaxf(d0, 0.f), 1.f);
        det[o * 2 + 1] = fminf(fmaxf(d1, 0.f), 1.f);
        scores[o] = sqrtf(sc[t]);
        locs[o] = loc / 32.f;
      }
      level_kept += chunk;
      __syncthreads();
    }
    if (tid == 0) counts[b * P.nlevels + l] = level_kept;
    out_base += level_kept;
    __syncthreads();
  }
}

extern "C" int drn_postprocess(const DrnLossLevel* levels, int nlevels, int B, const float* logits, const float* reg, const float* iou,
                               float thr, int top_n, float downsample, float* det, float* scores, float* locs, int* counts,
                               void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(levels && nlevels >= 1 && nlevels <= DRN_MAX_GROUPS && B > 0, "drn_postprocess: bad level table");
  DRN_CHECK_ARG(logits && reg && det && scores && locs && counts && top_n > 0 && downsample > 0.f, "drn_postprocess: bad args");
  PostParams P;
  memset(&P, 0, sizeof(P));
  P.nlevels = nlevels; P.B = B; P.thr = thr; P.top_n = top_n; P.downsample = downsample; P.use_iou = iou != nullptr;
  int rows = 0, per_clip = 0;
  for (int l = 0; l < nlevels; ++l) {
    DRN_CHECK_ARG(levels[l].L > 0 && levels[l].L <= PP_MAX_L, "drn_postprocess: level %d has %d locations (max %d)", l, levels[l].L, PP_MAX_L);
    P.row_start[l] = rows; P.L[l] = levels[l].L; P.stride[l] = levels[l].stride;
    rows += B * levels[l].L;
    per_clip += levels[l].L;
  }
  P.rows_per_clip = per_clip;
  postprocess_kernel<<<B, PP_THREADS, 0, (hipStream_t)stream>>>(P, logits, reg, iou, det, scores, locs, counts);
  return drn_launch_status("drn_postprocess");
}

// ---------------------------------------------------------------------------------------------------------------------------
// Recall@k at temporal-IoU thresholds with temporal NMS, on the device (utils/evaluate_utils.py:131-215 as driven by
// main.py:324-364): one wavefront per (clip, IoU threshold).  The host evaluator sorts a clip's predictions by score (stable),
// runs a greedy NMS at threshold iou - 0.05 from the highest score down -- score ties go to the LATER prediction -- and counts
// the clip when one of the first k survivors overlaps the ground truth by >= iou (un-clamped IoU).  Only the first
// K = max(k) survivors matter, so no sort is needed: K times, the best candidate still alive is found with a wave reduction
// over (score, index), tested against the ground truth, and everything it suppresses is struck out.  All arithmetic in
// double on the float32 detections, exactly the numbers the host path sees after .tolist().
// out[b][q] = position of the first survivor that hits (0-based), or K when none of the first K does.
// every level table drn_postprocess accepts fits: DRN_MAX_GROUPS levels of at most PP_MAX_L locations
#define ER_MAX_CAND (DRN_MAX_GROUPS * PP_MAX_L)
__global__ __launch_bounds__(64) void eval_recall_kernel(const float* __restrict__ det, const float* __restrict__ scores,
                                                         const int* __restrict__ counts, int nlevels, int rows_per_clip,
                                                         const void* __restrict__ gt, int gt_f64, const double* __restrict__ ious,
                                                         int K, int* __restrict__ out, int n_iou) {
  __shared__ unsigned char alive[ER_MAX_CAND];
  const int b = blockIdx.x, q = blockIdx.y, lane = threadIdx.x;
  int n = 0;
  for (int l = 0; l < nlevels; ++l) n += counts[b * nlevels + l];
  const float* __restrict__ d = det + (long)b * rows_per_clip * 2;
  const float* __restrict__ s = scores + (long)b * rows_per_clip;
  const double iou_thr = ious[q], overlap = ious[q] - 0.05;
  const double g0 = gt_f64 ? ((const double*)gt)[b * 2] : (double)((const float*)gt)[b * 2];
  const double g1 = gt_f64 ? ((const double*)gt)[b * 2 + 1] : (double)((const float*)gt)[b * 2 + 1];
  const bool empty = n == 0;                 // model/inference.py:192-197: one detection (0, 1) with score 1
  if (empty) n = 1;
  for (int j = lane; j < n; j += 64) alive[j] = 1;
  __syncthreads();
  int first_hit = K;
  for (int p = 0; p < K; ++p) {
    float bs = -1.f;
    int bj = -1;
    for (int j = lane; j < n; j += 64)
      if (alive[j]) {
        const float sj = empty ? 1.f : s[j];
        if (bj < 0 || sj > bs || (sj == bs && j > bj)) { bs = sj; bj = j; }
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64);
      const int oj = __shfl_xor(bj, o, 64);
      if (oj >= 0 && (bj < 0 || os > bs || (os == bs && oj > bj))) { bs = os; bj = oj; }
    }
    if (bj < 0) break;                        // nothing left
    const double x1 = empty ? 0.0 : (double)d[bj * 2], x2 = empty ? 1.0 : (double)d[bj * 2 + 1];
    if (first_hit == K) {
      const double iou = (fmin(g1, x2) - fmax(g0, x1)) / (fmax(g1, x2) - fmin(g0, x1));      // evaluate_utils.py:228-232
      if (iou >= iou_thr) first_hit = p;
    }
    const double len = x2 - x1;
    for (int j = lane; j < n; j += 64)
      if (alive[j]) {
        const double y1 = (double)d[j * 2], y2 = (double)d[j * 2 + 1];
        const double inter = fmax(0.0, fmin(x2, y2) - fmax(x1, y1));
        const double o = inter / (len + (y2 - y1) - inter);
        if (!(o <= overlap) || j == bj) alive[j] = 0;     // (0/0 = NaN is not <= overlap: struck out, as in the reference)
      }
    __syncthreads();
  }
  if (lane == 0) out[b * n_iou + q] = first_hit;
}

extern "C" int drn_eval_recall(const float* det, const float* scores, const int32_t* counts, int B, int nlevels, int rows_per_clip,
                               const void* gt, int gt_is_f64, const double* ious, int n_iou, int max_topk, int32_t* first_hit,
                               void* stream) {
  drn_clear_status();
  DRN_CHECK_ARG(det && scores && counts && gt && ious && first_hit && B > 0 && nlevels >= 1 && n_iou >= 1 && max_topk >= 1,
                "drn_eval_recall: bad args");
  DRN_CHECK_ARG(rows_per_clip > 0 && rows_per_clip <= ER_MAX_CAND, "drn_eval_recall: %d candidate slots per clip (max %d)", rows_per_clip, ER_MAX_CAND);
  eval_recall_kernel<<<dim3(B, n_iou), 64, 0, (hipStream_t)stream>>>(det, scores, counts, nlevels, rows_per_clip, gt, gt_is_f64, ious,
                                                                     max_topk, first_hit, n_iou);
  return drn_launch_status("drn_eval_recall");
}
