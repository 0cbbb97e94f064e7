// Small NAML / LSTUR kernels around the convolution and pooling kernels:
//   element_table_*   NAML ElementEncoder relu(Linear(embedding(id))) (src/model/NAML/news_encoder.py:40-47) evaluated ONCE per
//                     category row instead of once per news item: E[c][:] = relu(W emb[c] + b) for all num_categories rows
//                     (275 x 300 outputs, L2 resident); a news item then only gathers its row.
//   views_fill        rows 2 (category) and 3 (subcategory) of the [T][4] view stack that feeds final_attention (:108-114).
//   rows_scatter_add  generic fp32 row scatter-add with atomics (LSTUR user_embedding gradient: B rows of 900).
#pragma once
#include "nr_common.h"

namespace nr {

// The element tables are tiny (2 x 275 x 300 outputs of a 100-term dot product; gradients: 88 K outputs of 275- / 600-term sums), and a thread per
// output made their launches chains of dependent load -> FMA round trips: 68 + 139 us of the NAML step for 0.2 GFLOP (round 5).  Round 6: EIGHT lanes
// per output, each walking every eighth term, combined with three lane exchanges in a fixed order (deterministic).
constexpr int ET_LANES = 8;
__device__ __forceinline__ float et_reduce8(float a) {           // sum over the 8 consecutive lanes of an output
  a += shfl_xor(a, 1);
  a += shfl_xor(a, 2);
  a += shfl_xor(a, 4);
  return a;
}

// E[which][c][f] = relu(b_which[f] + sum_k emb[c][k] * W_which[f][k]);  which = 0 (category), 1 (subcategory)
__global__ __launch_bounds__(256) void element_table_fwd_kernel(const float* __restrict__ emb, int ncat, int dcat,
                                                                const float* __restrict__ W0, const float* __restrict__ b0,
                                                                const float* __restrict__ W1, const float* __restrict__ b1,
                                                                float* __restrict__ E, int F_) {
  const int total = 2 * ncat * F_;
  const int sub = threadIdx.x & (ET_LANES - 1);
  const int per = blockDim.x / ET_LANES;
  const int rounds = (total + gridDim.x * per - 1) / (gridDim.x * per);      // every lane of a wave runs the same number of rounds (lane exchanges)
  for (int r = 0; r < rounds; ++r) {
    const int i = (r * gridDim.x + blockIdx.x) * per + (threadIdx.x / ET_LANES);
    const bool live = i < total;
    const int ii = live ? i : 0;
    const int which = ii / (ncat * F_), rem = ii - which * ncat * F_;
    const int c = rem / F_, f = rem - c * F_;
    const float* W = (which ? W1 : W0) + (size_t)f * dcat;
    const float* e = emb + (size_t)c * dcat;
    float a = 0.0f;
    for (int k = sub; k < dcat; k += ET_LANES) a += e[k] * W[k];
    a = et_reduce8(a) + (which ? b1 : b0)[f];
    if (live && sub == 0) E[i] = fmaxf(a, 0.0f);
  }
}

// Backward of the table: dpre = dE * [E > 0];  dW[which][f][k] = sum_c dpre[c][f] emb[c][k];  db[which][f] = sum_c dpre[c][f];
// demb[c][k] = sum_which sum_f dpre[which][c][f] W_which[f][k] for c != 0 (padding_idx = 0 row gets no gradient).
// One launch, three output regions indexed by a flat work id.
__global__ __launch_bounds__(256) void element_table_bwd_kernel(const float* __restrict__ emb, int ncat, int dcat,
                                                                const float* __restrict__ W0, const float* __restrict__ W1,
                                                                const float* __restrict__ E, const float* __restrict__ dE, int F_,
                                                                float* __restrict__ dW /*[2][F][dcat]*/, float* __restrict__ db /*[2][F]*/,
                                                                float* __restrict__ demb /*[ncat][dcat]*/) {
  const int nW = 2 * F_ * dcat, nb = 2 * F_, ne = ncat * dcat, total = nW + nb + ne;
  const int sub = threadIdx.x & (ET_LANES - 1);
  const int per = blockDim.x / ET_LANES;
  const int rounds = (total + gridDim.x * per - 1) / (gridDim.x * per);
  for (int r = 0; r < rounds; ++r) {
    const int i0 = (r * gridDim.x + blockIdx.x) * per + (threadIdx.x / ET_LANES);
    const bool live = i0 < total;
    const int i = live ? i0 : 0;
    float a = 0.0f;
    if (i < nW) {
      const int which = i / (F_ * dcat), rem = i - which * F_ * dcat;
      const int f = rem / dcat, k = rem - f * dcat;
      const float* Ew = E + (size_t)which * ncat * F_;
      const float* dEw = dE + (size_t)which * ncat * F_;
      for (int c = sub; c < ncat; c += ET_LANES) a += (Ew[c * F_ + f] > 0.0f ? dEw[c * F_ + f] : 0.0f) * emb[(size_t)c * dcat + k];
    } else if (i < nW + nb) {
      const int j = i - nW, which = j / F_, f = j - which * F_;
      const float* Ew = E + (size_t)which * ncat * F_;
      const float* dEw = dE + (size_t)which * ncat * F_;
      for (int c = sub; c < ncat; c += ET_LANES) a += Ew[c * F_ + f] > 0.0f ? dEw[c * F_ + f] : 0.0f;
    } else {
      const int j = i - nW - nb, c = j / dcat, k = j - c * dcat;
      if (c != 0) {
        for (int which = 0; which < 2; ++which) {
          const float* Ew = E + ((size_t)which * ncat + c) * F_;
          const float* dEw = dE + ((size_t)which * ncat + c) * F_;
          const float* W = which ? W1 : W0;
          for (int f = sub; f < F_; f += ET_LANES) a += (Ew[f] > 0.0f ? dEw[f] : 0.0f) * W[(size_t)f * dcat + k];
        }
      }
    }
    a = et_reduce8(a);
    if (live && sub == 0) (i < nW ? dW + i : i < nW + nb ? db + (i - nW) : demb + (i - nW - nb))[0] = a;
  }
}

// views[(4 t + 2)][:] = E[0][cat[t]][:], views[(4 t + 3)][:] = E[1][sub[t]][:]  as bf16 ctx rows (col D = 1.0, cols > D zero)
__global__ __launch_bounds__(256) void views_fill_kernel(const int64_t* __restrict__ cat, const int64_t* __restrict__ sub,
                                                         const float* __restrict__ E, int ncat, u16* __restrict__ views, int64_t T) {
  constexpr int PC = KP / 4;
  const int64_t total = T * 2 * PC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / (2 * PC);
    const int rem = (int)(i - t * 2 * PC);
    const int which = rem / PC, c = rem - which * PC;
    u16x4 o = u16x4{0, 0, 0, 0};
    if (c < D4) {
      int64_t id = which ? sub[t] : cat[t];
      id = id < 0 ? 0 : (id >= ncat ? ncat - 1 : id);
      o = pack4(*(const f32x4*)(E + (((size_t)which * ncat + id) * D4 + c) * 4));
    } else if (c == D4) {
      o[0] = 0x3F80;
    }
    *(u16x4*)(views + ((t * 4 + 2 + which) * PC + c) * 4) = o;
  }
}

// dst[ids[i]][0:d] += scale_i * src[i][0:d] for ids[i] > pad_row (fp32 atomics).  row_scale (optional, [n]) carries a per-row factor
// (LSTUR: the dropout2d keep/(1-p) factor of the user row).
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(const int64_t* __restrict__ ids, const float* __restrict__ src, int64_t lds_,
                                                               const float* __restrict__ row_scale, float* __restrict__ dst, int64_t num_rows,
                                                               int d, int64_t n, int pad_row) {
  const int64_t total = n * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / d;
    const int c = (int)(i - r * d);
    const int64_t id = ids[r];
    if (id <= pad_row || id >= num_rows) continue;
    const float s = row_scale ? row_scale[r] : 1.0f;
    if (s != 0.0f) atomic_add(dst + id * d + c, src[r * lds_ + c] * s);
  }
}

// out[i][0:d] = table[ids[i]][0:d] * row_scale[i]  into a strided destination (row stride ldo floats): LSTUR category /
// subcategory embedding columns of the 900-d news vector and the dropout2d-masked user row.
__global__ __launch_bounds__(256) void gather_rows_strided_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                                  int64_t num_rows, int d, const float* __restrict__ row_scale,
                                                                  float* __restrict__ out, int64_t ldo, int64_t n) {
  const int d4 = d / 4;
  const int64_t total = n * d4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / d4;
    const int c = (int)(i - r * d4);
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= num_rows ? num_rows - 1 : id);
    f32x4 v = *(const f32x4*)(table + (id * d4 + c) * 4);
    if (row_scale) v = v * row_scale[r];
    *(f32x4*)(out + r * ldo + c * 4) = v;
  }
}

// scalar variant for widths that are not a multiple of 4 (LSTUR 'con': user rows of 1.5 F = 450 floats)
__global__ __launch_bounds__(256) void gather_rows_strided_scalar_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                                         int64_t num_rows, int d, const float* __restrict__ row_scale,
                                                                         float* __restrict__ out, int64_t ldo, int64_t n) {
  const int64_t total = n * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / d;
    const int c = (int)(i - r * d);
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= num_rows ? num_rows - 1 : id);
    out[r * ldo + c] = table[id * d + c] * (row_scale ? row_scale[r] : 1.0f);
  }
}

}  // namespace nr
