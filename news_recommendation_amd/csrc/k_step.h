// Step-level kernels (round 6): what a training step used to leave to framework element-wise kernels between the engine's own launches --
//   * score_ce_*: DotProductClickPredictor + CrossEntropyLoss of the training loop in one pass (dot_product.py:8-19, train.py:205-206 /
//     LSTUR train.py:186-187: `loss = criterion(y_pred, y)` with y = 0), forward and the logit gradient together;
//   * accum_many_kernel: every small weight gradient of a backward pass added into its persistent gradient buffer by ONE launch
//     (what autograd's AccumulateGrad does with one add per parameter: 21 launches per NAML step).
// HBM-bound element work: float4 loads, one wave per impression / a grid-stride loop per item.
#pragma once
#include "nr_common.h"

namespace nr {

// ---- scorer + cross entropy ----------------------------------------------------------------------------------
// One wave per impression b: logits[b][c] = cand[b][c][:] . user[b][:] (same order of operations as score_dot_kernel: the logits are bit-identical
// to nr_score_dot's), loss_rows[b] = logsumexp(logits[b]) - logits[b][target[b]] (maximum subtracted, as log_softmax does), and
// dl[b][c] = (softmax(logits[b])[c] - [c == target[b]]) * inv_B -- the gradient of the MEAN loss with respect to the logits.
// C <= 64 (lane c keeps logit c).
__global__ __launch_bounds__(256) void score_ce_fwd_kernel(const float* __restrict__ cand, const float* __restrict__ user,
                                                           const int64_t* __restrict__ target, float* __restrict__ logits,
                                                           float* __restrict__ dl, float* __restrict__ loss_rows, int64_t B, int C, int d4,
                                                           float inv_B) {
  const int64_t b = (int64_t)blockIdx.x * 4 + wave_id();
  const int l = lane_id();
  if (b >= B) return;
  const f32x4* uv = (const f32x4*)(user + b * d4 * 4);
  float mine = -3.0e38f;                              // lane c ends up with logit c
  for (int c = 0; c < C; ++c) {
    const f32x4* cv = (const f32x4*)(cand + (b * C + c) * d4 * 4);
    float acc = 0.0f;
    for (int k = l; k < d4; k += 64) {
      f32x4 x = cv[k], y = uv[k];
      acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    acc = wave_sum(acc);
    if (l == c) mine = acc;
  }
  float mx = mine;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, shfl_xor(mx, m));
  const float e = (l < C) ? expf(mine - mx) : 0.0f;
  const float se = wave_sum(e);
  const int t = target ? (int)target[b] : 0;
  const float lt = wave_sum((l == t) ? mine : 0.0f);           // logit of the target class (t outside [0, C) is refused by the host check)
  if (l < C) {
    if (logits) logits[b * C + l] = mine;
    dl[b * C + l] = (e / se - ((l == t) ? 1.0f : 0.0f)) * inv_B;
  }
  if (l == 0) loss_rows[b] = logf(se) + mx - lt;
}

// loss = inv_B * sum_b loss_rows[b], one workgroup, fixed order (bit-reproducible)
__global__ __launch_bounds__(256) void score_ce_mean_kernel(const float* __restrict__ loss_rows, float* __restrict__ loss, int64_t B, float inv_B) {
  NR_SMEM_DECL(smem);
  float* red = (float*)smem;                          // [4]
  float s = 0.0f;
  for (int64_t i = threadIdx.x; i < B; i += 256) s += loss_rows[i];
  s = wave_sum(s);
  if (lane_id() == 0) red[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) * inv_B;
}

// d_cand[b][c][:] = g * dl[b][c] * user[b][:], d_user[b][:] = g * sum_c dl[b][c] * cand[b][c][:]; g = *gscale (the gradient arriving at the
// scalar loss; NULL = 1).  d_cand / d_user rows may live inside larger buffers (row strides ldc / ldu in floats): the candidates' gradient is
// written straight into the first rows of the encoder's output gradient.
__global__ __launch_bounds__(256) void score_ce_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ gscale,
                                                           const float* __restrict__ cand, const float* __restrict__ user,
                                                           float* __restrict__ d_cand, int64_t ldc, float* __restrict__ d_user, int64_t ldu,
                                                           int64_t B, int C, int d4) {
  const float g = gscale ? gscale[0] : 1.0f;
  const int64_t total = B * d4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / d4;
    const int c4 = (int)(i - b * d4);
    const f32x4 u = *(const f32x4*)(user + (b * d4 + c4) * 4);
    f32x4 du = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      const float g_ = dl[b * C + c] * g;
      const f32x4 cv = *(const f32x4*)(cand + ((b * C + c) * d4 + c4) * 4);
      du += cv * g_;
      *(f32x4*)(d_cand + (b * C + c) * ldc + c4 * 4) = u * g_;
    }
    *(f32x4*)(d_user + b * ldu + c4 * 4) = du;
  }
}

// ---- bf16 rows -> f32 rows ------------------------------------------------------------------------------------------
// dst[r][c] = src[r][c] for c < d (d a multiple of 4; row strides in elements, multiples of 4): the dense input gradient of an encoder
// (bf16 [n][KP], the NT GEMM's output format) handed to autograd in the f32 [n][D] layout of the tensor it belongs to.
__global__ __launch_bounds__(256) void rows_to_f32_kernel(const u16* __restrict__ src, int64_t lds_, int d4, float* __restrict__ dst, int64_t ldd,
                                                          int64_t n) {
  const int64_t total = n * d4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / d4;
    const int c = (int)(i - r * d4) * 4;
    const u16x4 v = *(const u16x4*)(src + r * lds_ + c);
    *(f32x4*)(dst + r * ldd + c) = f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
  }
}

// ---- batched strided accumulate ---------------------------------------------------------------------------------
// item i: dst[r * dst_ld + c] += sum_{p < parts} src[p * part_stride + r * src_ld + c] for r < rows, c < cols (parts = 1: a plain add; parts > 1:
// the per-workgroup partial sums of a persistent kernel -- the query-vector gradient of a pooling level, one row per workgroup -- summed in a fixed
// order on the way).  blockIdx.y = item, blockIdx.x strides its elements.
struct AccumItem {
  const float* src;
  float* dst;
  int64_t src_ld, dst_ld;
  int32_t rows, cols;
  int32_t parts, pad_;
  int64_t part_stride;
};
constexpr int ACCUM_MAX_ITEMS = 48;      // 48 x 56 B of kernel arguments
struct AccumBatch {
  AccumItem it[ACCUM_MAX_ITEMS];
};

constexpr int ACCUM_PG = 16;               // part groups of an item with parts > 1 (fixed: the order of the additions is part of the result)

__global__ __launch_bounds__(256) void accum_many_kernel(AccumBatch batch) {
  NR_SMEM_DECL(smem);
  const AccumItem it = batch.it[blockIdx.y];
  const int64_t n = (int64_t)it.rows * it.cols;
  if (it.parts <= 1) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t r = i / it.cols;
      const int c = (int)(i - r * it.cols);
      it.dst[r * it.dst_ld + c] += it.src[r * it.src_ld + c];
    }
    return;
  }
  // parts > 1 (uniform per workgroup): 16 elements x 16 part groups per pass -- lane (e, pg) sums the parts [pg * per, (pg + 1) * per) of element e in
  // part order, the 16 group sums are added in group order.  A single lane walking 256 partial rows one batch of loads after the other took ~50 us.
  float* red = (float*)smem;                 // [ACCUM_PG][16]
  const int e = threadIdx.x & 15, pg = threadIdx.x >> 4;
  const int per = (it.parts + ACCUM_PG - 1) / ACCUM_PG;
  const int q0 = pg * per, q1 = (q0 + per < it.parts) ? q0 + per : it.parts;
  for (int64_t base = (int64_t)blockIdx.x * 16; base < n; base += (int64_t)gridDim.x * 16) {
    const int64_t i = base + e;
    float v = 0.0f;
    int64_t r = 0;
    int c = 0;
    if (i < n) {
      r = i / it.cols;
      c = (int)(i - r * it.cols);
      const float* sp = it.src + r * it.src_ld + c;
      int q = q0;
      for (; q + 4 <= q1; q += 4) {
        const float a0 = sp[(int64_t)q * it.part_stride], a1 = sp[(int64_t)(q + 1) * it.part_stride];
        const float a2 = sp[(int64_t)(q + 2) * it.part_stride], a3 = sp[(int64_t)(q + 3) * it.part_stride];
        v += a0; v += a1; v += a2; v += a3;
      }
      for (; q < q1; ++q) v += sp[(int64_t)q * it.part_stride];
    }
    red[pg * 16 + e] = v;
    __syncthreads();
    if (pg == 0 && i < n) {
      float t = 0.0f;
#pragma unroll
      for (int g = 0; g < ACCUM_PG; ++g) t += red[g * 16 + e];
      it.dst[r * it.dst_ld + c] += t;
    }
    __syncthreads();
  }
}

}  // namespace nr
