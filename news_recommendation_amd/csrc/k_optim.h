// Optimiser kernels: the Adam update of src/train.py:127-128,227-233 (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight
// decay, no amsgrad) on the engine's flat parameter / gradient buffers.
//
//   adam_flat_kernel        one pass over a flat fp32 region: reads p, g, m, v; writes p, m, v and (zero_grad) clears g.  The gradient is
//                           scaled by grad_scale (1 / world size) on the way in, so the data-parallel mean needs no extra pass, and the
//                           cleared gradient replaces optimizer.zero_grad() / flat.zero_() of the next step.
//   row_adam_*_kernel       the same update for a ROW-SPARSE table (LSTUR user_embedding, src/model/LSTUR/__init__.py:38-42: 711 k
//                           users x 900 at MIND-large scale = 2.56 GB, of which a step touches B rows).  Dense Adam keeps updating a
//                           row after its last gradient (m, v decay; p keeps moving along m / sqrt(v)); a row that never received
//                           a gradient has m = v = 0 and never moves.  These kernels evaluate EXACTLY that dense recurrence, but
//                           lazily: last[row] = the step up to which the row's (p, m, v) are current; before a row is read (forward
//                           gather) or updated, the zero-gradient steps it missed are replayed one by one (same fp32 operations in the
//                           same order as the dense kernel would have applied them), so HBM traffic per step is O(touched rows).
//
// Per-step scalars come from a device table sched[2 * step] = lr / (1 - beta1^step), sched[2 * step + 1] = sqrt(1 - beta2^step),
// computed on the host in double exactly as torch does (_single_tensor_adam): dense and lazy kernels read the same values.
// Element update (torch/optim/adam.py _single_tensor_adam):
//   m <- m + (g - m) * (1 - beta1);  v <- v * beta2 + (1 - beta2) * g * g;  p <- p - step_size * m / (sqrt(v) / bc2_sqrt + eps)
#pragma once
#include "nr_common.h"

namespace nr {

struct AdamCfg {
  const uint32_t* t_dev; // optional device-resident step counter (nr_set_step_counter): the dense kernel and the row-sparse step / catch-up
                        // kernels then take their step index from it instead of their by-value argument, so that a step captured into a
                        // HIP graph advances at every replay (the flush kernel is never part of a step and keeps its argument)
  const uint32_t* gate; // optional fault words of the process (nr_set_fault_words / the library's own): while word 0 or 1 is non-zero -- a
                        // persistent GRU sweep gave up a wait, its outputs and every gradient of that step are garbage -- NO update is
                        // applied: the dense kernel only clears the gradient, the row-sparse step and catch-up leave.  The first skipped step
                        // index is recorded in word 2 (the host repeats the steps from there: train_fast.py).  The flush kernel (never part
                        // of a step) is not gated.
  const float* sched;   // [2 * (max_step + 1)]
  float om_b1;          // 1 - beta1
  float b2, om_b2;      // beta2, 1 - beta2
  float eps;
};

// No FMA contraction: every operation rounds on its own, so the dense kernel, the lazy kernels and the CPU emulation build produce the
// same bits (sqrtf and / are correctly rounded on gfx950 with hipcc's defaults).
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, float om_b1, float b2, float om_b2, float eps,
                                          float step_size, float bc2_sqrt) {
#pragma clang fp contract(off)
  m = m + (g - m) * om_b1;
  v = v * b2 + om_b2 * g * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}

// the zero-gradient step a dense Adam applies to a row that was not touched
__device__ __forceinline__ void adam_elem_idle(float& p, float& m, float& v, float om_b1, float b2, float eps, float step_size,
                                               float bc2_sqrt) {
#pragma clang fp contract(off)
  m = m + (0.0f - m) * om_b1;
  v = v * b2;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}

// true = skip this optimiser step (see AdamCfg::gate); one thread of the launch records the step index of the first skipped step
__device__ __forceinline__ bool adam_gated(const AdamCfg& c, int64_t step) {
  if (c.gate == nullptr) return false;
  uint32_t* gw = const_cast<uint32_t*>(c.gate);
  if ((ld_relaxed_u32(gw) | ld_relaxed_u32(gw + 1)) == 0u) return false;
  if (blockIdx.x == 0 && threadIdx.x == 0) atomic_cas_u32(gw + 2, 0u, (uint32_t)step);
  return true;
}

__global__ __launch_bounds__(64) void step_counter_add_kernel(uint32_t* ctr, uint32_t inc) {
  if (threadIdx.x == 0) *ctr += inc;
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, AdamCfg c, int64_t step, float grad_scale,
                                                        int zero_grad) {
  if (c.t_dev != nullptr) step = (int64_t)*c.t_dev;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (adam_gated(c, step)) {               // a failed sweep upstream: the gradients are garbage -- drop them, move nothing
    if (zero_grad) {
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) *(f32x4*)(g + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      const int64_t tt = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
      if (tt < n) g[tt] = 0.0f;
    }
    return;
  }
  const float step_size = c.sched[2 * step], bc2_sqrt = c.sched[2 * step + 1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = *(f32x4*)(p + i * 4), gg = *(const f32x4*)(g + i * 4), mm = *(f32x4*)(m + i * 4), vv = *(f32x4*)(v + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = pp[j], b = mm[j], d = vv[j];
      adam_elem(a, b, d, gg[j] * grad_scale, c.om_b1, c.b2, c.om_b2, c.eps, step_size, bc2_sqrt);
      pp[j] = a; mm[j] = b; vv[j] = d;
    }
    *(f32x4*)(p + i * 4) = pp; *(f32x4*)(m + i * 4) = mm; *(f32x4*)(v + i * 4) = vv;
    if (zero_grad) *(f32x4*)(g + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // tail (n % 4 elements)
  const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    float pp = p[t], mm = m[t], vv = v[t];
    adam_elem(pp, mm, vv, g[t] * grad_scale, c.om_b1, c.b2, c.om_b2, c.eps, step_size, bc2_sqrt);
    p[t] = pp; m[t] = mm; v[t] = vv;
    if (zero_grad) g[t] = 0.0f;
  }
}

// ---- row-sparse (lazy) form ------------------------------------------------------------------------------------------------------
// One wave per row; lane l owns elements l, l + 64, ... (at most ROW_EPL per lane: d <= 64 * ROW_EPL).
constexpr int ROW_EPL = 16;      // d <= 1024 (LSTUR: 900 'ini', 450 'con')

struct RowState {
  float p[ROW_EPL], m[ROW_EPL], v[ROW_EPL];
};

__device__ __forceinline__ void row_load(RowState& r, const float* p, const float* m, const float* v, int64_t row, int d, int l) {
#pragma unroll
  for (int e = 0; e < ROW_EPL; ++e) {
    const int c = l + 64 * e;
    if (c < d) { r.p[e] = p[row * d + c]; r.m[e] = m[row * d + c]; r.v[e] = v[row * d + c]; }
  }
}
__device__ __forceinline__ void row_store(const RowState& r, float* p, float* m, float* v, int64_t row, int d, int l) {
#pragma unroll
  for (int e = 0; e < ROW_EPL; ++e) {
    const int c = l + 64 * e;
    if (c < d) { p[row * d + c] = r.p[e]; m[row * d + c] = r.m[e]; v[row * d + c] = r.v[e]; }
  }
}
// replay the idle steps (from, upto]: what dense Adam did to this row while nobody looked
__device__ __forceinline__ void row_replay(RowState& r, const AdamCfg& c, int64_t from, int64_t upto, int d, int l) {
  for (int64_t s = from + 1; s <= upto; ++s) {
    const float step_size = c.sched[2 * s], bc2_sqrt = c.sched[2 * s + 1];
#pragma unroll
    for (int e = 0; e < ROW_EPL; ++e)
      if (l + 64 * e < d) adam_elem_idle(r.p[e], r.m[e], r.v[e], c.om_b1, c.b2, c.eps, step_size, bc2_sqrt);
  }
}

// Bring the listed rows up to date with step `upto` (the number of optimiser steps taken so far).  ids may repeat: the wave that
// swaps last[row] to `upto` first owns the row, the others see it current and leave.  Rows with last == 0 never received a gradient:
// m = v = 0, dense Adam does not move them, nothing to do.
__global__ __launch_bounds__(256) void row_adam_catchup_kernel(const int64_t* __restrict__ ids, int64_t n, float* __restrict__ p,
                                                               float* __restrict__ m, float* __restrict__ v, int* __restrict__ last,
                                                               int64_t num_rows, int d, int64_t upto, AdamCfg c) {
  // with a device step counter attached (HIP-graph replays, graph.py) the counter was bumped at the START of the step this launch belongs
  // to: the steps taken so far are one less
  if (c.t_dev != nullptr) upto = (int64_t)*c.t_dev - 1;
  if (adam_gated(c, upto + 1)) return;      // (the rows stay as of the last good step: the repeated steps catch them up again)
  const int64_t i = (int64_t)blockIdx.x * 4 + wave_id();
  if (i >= n || upto <= 0) return;
  const int l = lane_id();
  const int64_t row = ids[i];
  if (row < 0 || row >= num_rows) return;
  int old = -1;
  if (l == 0) {
    const int seen = last[row];            // 0: never received a gradient (no state, nothing moves);  >= upto: current
    if (seen > 0 && seen < upto) old = atomic_exch(&last[row], (int)upto);      // claim the row; a duplicate id's wave then finds it current
  }
  old = uniform(__builtin_bit_cast(int, shfl(__builtin_bit_cast(float, old), 0)));
  if (old <= 0 || old >= upto) return;
  RowState r;
  row_load(r, p, m, v, row, d, l);
  row_replay(r, c, old, upto, d, l);
  row_store(r, p, m, v, row, d, l);
}

// every row with state (last > 0) that is behind `upto`: state_dict() / checkpoint / evaluation read the whole table
__global__ __launch_bounds__(256) void row_adam_flush_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                             int* __restrict__ last, int64_t num_rows, int d, int64_t upto, AdamCfg c) {
  const int l = lane_id();
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave_id(); row < num_rows; row += (int64_t)gridDim.x * 4) {
    const int old = uniform(last[row]);
    wave_barrier();                          // every lane has read the stamp before lane 0 rewrites it (lock-step on hardware; the emulator runs lanes in turn)
    if (old <= 0 || old >= upto) continue;
    RowState r;
    row_load(r, p, m, v, row, d, l);
    row_replay(r, c, old, upto, d, l);
    row_store(r, p, m, v, row, d, l);
    if (l == 0) last[row] = (int)upto;
  }
}

// The optimiser step `step` for the rows that received a gradient.  The (row id, gradient row) pairs of ALL ranks arrive sorted by id
// (ids_sorted ascending, perm = position of the pair in `rows`).  One wave per sorted position; the wave at the HEAD of a run of equal
// ids owns the row (the others leave): the run's rows are summed in position order (deterministic: every rank computes the same bits),
// scaled by grad_scale, the row is caught up to step - 1 and then takes the real update.  No run table, hence no host round trip to
// size one.  Rows <= pad_row (nn.Embedding padding_idx) are skipped.
__global__ __launch_bounds__(256) void row_adam_step_kernel(const int64_t* __restrict__ ids_sorted, const int64_t* __restrict__ perm,
                                                            int64_t n, const float* __restrict__ rows, int64_t ld, float* __restrict__ p,
                                                            float* __restrict__ m, float* __restrict__ v, int* __restrict__ last,
                                                            int64_t num_rows, int d, int64_t step, AdamCfg c, float grad_scale,
                                                            int pad_row) {
  if (c.t_dev != nullptr) step = (int64_t)*c.t_dev;             // as in adam_flat_kernel: the step index of a replayed graph comes from the counter
  if (adam_gated(c, step)) return;
  const int64_t a = (int64_t)blockIdx.x * 4 + wave_id();
  if (a >= n) return;
  const int l = lane_id();
  const int64_t row = ids_sorted[a];
  if (row <= pad_row || row >= num_rows) return;
  if (a > 0 && ids_sorted[a - 1] == row) return;                 // not the head of its run
  int64_t b = a + 1;
  while (b < n && ids_sorted[b] == row) ++b;
  float g[ROW_EPL];
#pragma unroll
  for (int e = 0; e < ROW_EPL; ++e) g[e] = 0.0f;
  for (int64_t i = a; i < b; ++i) {
    const float* src = rows + perm[i] * ld;
#pragma unroll
    for (int e = 0; e < ROW_EPL; ++e)
      if (l + 64 * e < d) g[e] += src[l + 64 * e];
  }
  RowState r;
  row_load(r, p, m, v, row, d, l);
  const int old = uniform(last[row]);
  wave_barrier();                            // as in the flush kernel: read by all lanes before lane 0 stamps the row
  if (old > 0 && old < step - 1) row_replay(r, c, old, step - 1, d, l);
  const float step_size = c.sched[2 * step], bc2_sqrt = c.sched[2 * step + 1];
#pragma unroll
  for (int e = 0; e < ROW_EPL; ++e)
    if (l + 64 * e < d) adam_elem(r.p[e], r.m[e], r.v[e], g[e] * grad_scale, c.om_b1, c.b2, c.om_b2, c.eps, step_size, bc2_sqrt);
  row_store(r, p, m, v, row, d, l);
  if (l == 0) last[row] = (int)step;
}

}  // namespace nr
