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
