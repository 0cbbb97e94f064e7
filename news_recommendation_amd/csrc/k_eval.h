// Per-impression ranking metrics on the device: AUC, MRR, nDCG@5, nDCG@10 of src/evaluate.py:24-42,160-168
// (roc_auc_score + mrr_score + ndcg_score per impression), one wave per impression, for the batched evaluation driver that
// replaces the per-impression Python loop + multiprocessing pool of src/evaluate.py:245-268.
//
//   rank_i (0-based position in np.argsort(score)[::-1]) = #{j : s_j > s_i} + #{j : s_j == s_i, j > i}
//     (ties: the reversed ascending sort puts the higher index first; exact for numpy's stable small-array path,
//      the reference's tie order is otherwise implementation-defined)
//   AUC   = sum_{pos i} ( #{neg j : s_j < s_i} + 0.5 #{neg j : s_j == s_i} ) / (P * N)          (Mann-Whitney = trapezoidal ROC area)
//   MRR   = sum_{pos i} 1 / (rank_i + 1) / P
//   nDCG@k = sum_{pos i, rank_i < k} 1 / log2(rank_i + 2)  /  sum_{r < min(P, k)} 1 / log2(r + 2)
//   Single-class impressions follow what src/evaluate.py:160-168 does with the scikit-learn it runs on today (1.7.2, observed by running
//   the reference here): roc_auc_score returns NaN with a warning instead of raising, so an ALL-POSITIVE impression keeps its MRR and nDCG
//   (AUC = NaN only) and an all-negative one is 0 / 0 = NaN in all four; the caller's nanmean skips NaNs column by column.  (Under
//   scikit-learn < 1.? roc_auc_score raised ValueError and the reference dropped all four values of both kinds.)
#pragma once
#include "nr_common.h"
#include "k_misc.h"

namespace nr {

__global__ __launch_bounds__(256) void impression_metrics_kernel(const float* __restrict__ scores, const int32_t* __restrict__ labels,
                                                                 const int64_t* __restrict__ ptr, float* __restrict__ out, int64_t n_impr) {
  const int64_t imp = (int64_t)blockIdx.x * 4 + wave_id();
  if (imp >= n_impr) return;
  const int l = lane_id();
  const int64_t b = ptr[imp], e = ptr[imp + 1];
  const int C = (int)(e - b);
  const float* s = scores + b;
  const int32_t* y = labels + b;
  float npos = 0.f;
  for (int i = l; i < C; i += 64) npos += y[i] != 0 ? 1.f : 0.f;
  npos = wave_sum(npos);
  const float nneg = (float)C - npos;
  float auc = 0.f, mrr = 0.f, d5 = 0.f, d10 = 0.f;
  for (int i = l; i < C; i += 64) {
    if (y[i] == 0) continue;
    const float si = s[i];
    int rank = 0;
    float below = 0.f;
    for (int j = 0; j < C; ++j) {
      const float sj = s[j];
      rank += (sj > si || (sj == si && j > i)) ? 1 : 0;
      if (y[j] == 0) below += sj < si ? 1.f : (sj == si ? 0.5f : 0.f);
    }
    auc += below;
    mrr += 1.0f / (float)(rank + 1);
    const float g = 1.0f / log2f((float)(rank + 2));
    if (rank < 5) d5 += g;
    if (rank < 10) d10 += g;
  }
  auc = wave_sum(auc); mrr = wave_sum(mrr); d5 = wave_sum(d5); d10 = wave_sum(d10);
  if (l == 0) {
    float* o = out + imp * 4;
    const float nan = __builtin_nanf("");
    if (npos == 0.f) {
      o[0] = o[1] = o[2] = o[3] = nan;
    } else {
      float b5 = 0.f, b10 = 0.f;
      for (int r = 0; r < 10 && r < (int)npos; ++r) {
        const float g = 1.0f / log2f((float)(r + 2));
        if (r < 5) b5 += g;
        b10 += g;
      }
      o[0] = nneg == 0.f ? nan : auc / (npos * nneg); o[1] = mrr / npos; o[2] = d5 / b5; o[3] = d10 / b10;
    }
  }
}

}  // namespace nr
