// Small kernels: stand-alone embedding gather, weight packing, dot-product scorers, dropout-mask export, MFMA probe.
#pragma once
#include "nr_common.h"

namespace nr {

// ---- K1: stand-alone embedding row gather (nn.Embedding forward) -------------------------------------------
// One float4 per lane; a d=300 row is 75 float4, so consecutive lanes read consecutive 16-B pieces of a row
// (coalesced 1200-B row reads).  Grid-stride over float4 pieces.
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                          float* __restrict__ out, int64_t n_tokens, int d4, int64_t num_rows) {
  int64_t total = n_tokens * d4;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t tok = i / d4;
    int c = (int)(i - tok * d4);
    int64_t id = ids[tok];
    id = id < 0 ? 0 : (id >= num_rows ? num_rows - 1 : id);
    f32x4 v = *(const f32x4*)(table + (id * d4 + c) * 4);
    *(f32x4*)(out + i * 4) = v;
  }
}

// ---- weight packing: fp32 nn.Linear parameters -> zero-padded bf16 MFMA operand layout -----------------------
// (each packing loop is a __device__ body over (block index, block count) shared by its own kernel and by pack_encoder_kernel, which runs all
//  operand packings of one encoder in ONE launch: nr_engine.hip)
__device__ __forceinline__ void pack_qkv_body(const float* __restrict__ Wq, const float* __restrict__ bq,
                                              const float* __restrict__ Wk, const float* __restrict__ bk,
                                              const float* __restrict__ Wv, const float* __restrict__ bv,
                                              u16* __restrict__ Wp, float* __restrict__ bp, int bid, int nb) {
  int total = 3 * NP * KP;
  for (int i = bid * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
    int row = i / KP, k = i - row * KP;
    int which = row / NP, n = row - which * NP;
    const float* W = which == 0 ? Wq : (which == 1 ? Wk : Wv);
    float v = (n < D && k < D) ? W[n * D + k] : 0.0f;
    Wp[tile_off(row, k, KP)] = f2bf(v);
    if (k == 0) {
      const float* b = which == 0 ? bq : (which == 1 ? bk : bv);
      bp[row] = n < D ? b[n] : 0.0f;
    }
  }
}
__global__ __launch_bounds__(256) void pack_qkv_kernel(const float* __restrict__ Wq, const float* __restrict__ bq,
                                                       const float* __restrict__ Wk, const float* __restrict__ bk,
                                                       const float* __restrict__ Wv, const float* __restrict__ bv,
                                                       u16* __restrict__ Wp, float* __restrict__ bp) {
  pack_qkv_body(Wq, bq, Wk, bk, Wv, bv, Wp, bp, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void pack_additive_body(const float* __restrict__ Wa, const float* __restrict__ ba,
                                                   const float* __restrict__ qv, int qdim, u16* __restrict__ Wap,
                                                   float* __restrict__ bap, float* __restrict__ qvp, int bid, int nb) {
  int total = QP * KP;
  for (int i = bid * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
    int n = i / KP, k = i - n * KP;
    float v = (n < qdim && k < D) ? Wa[n * D + k] : 0.0f;
    Wap[tile_off(n, k, KP)] = f2bf(v);
    if (k == 0) {
      bap[n] = n < qdim ? ba[n] : 0.0f;
      qvp[n] = n < qdim ? qv[n] : 0.0f;
    }
  }
}
__global__ __launch_bounds__(256) void pack_additive_kernel(const float* __restrict__ Wa, const float* __restrict__ ba,
                                                            const float* __restrict__ qv, int qdim, u16* __restrict__ Wap,
                                                            float* __restrict__ bap, float* __restrict__ qvp) {
  pack_additive_body(Wa, ba, qv, qdim, Wap, bap, qvp, blockIdx.x, gridDim.x);
}

// AdditiveAttention.linear weight transposed for the fused input-gradient product dctx = dpre @ Wa of the pooling backward kernels:
// WaT bf16 [KP][QKP] in tile order (QKP = QP rounded up to the MFMA k-step of 32), zero padded, with a PAIR-PERMUTED contraction
// index: slot kappa = 32 ks + 8 g + j of row d holds Wa[q][d] with q = 16 (2 ks + j / 4) + 4 g + j % 4.  The transposed projection
// product leaves a lane with rows 4g..4g+3 of each 16-row query tile; in this order the B fragment of k-step ks is just the lane's
// registers of tiles 2 ks and 2 ks + 1 back to back (k_pool2.h), or two 8-byte LDS reads (additive_bwd_kernel).
__device__ __forceinline__ void pack_additive_t_body(const float* __restrict__ Wa, int qdim, u16* __restrict__ WaT, int bid, int nb) {
  const int total = KP * QKP;
  for (int i = bid * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
    const int d = i / QKP, kap = i - d * QKP;
    const int ks = kap >> 5, g = (kap & 31) >> 3, j = kap & 7;
    const int q = 16 * (2 * ks + (j >> 2)) + 4 * g + (j & 3);
    WaT[tile_off(d, kap, QKP)] = f2bf((d < D && q < qdim) ? Wa[q * D + d] : 0.0f);
  }
}
__global__ __launch_bounds__(256) void pack_additive_t_kernel(const float* __restrict__ Wa, int qdim, u16* __restrict__ WaT) {
  pack_additive_t_body(Wa, qdim, WaT, blockIdx.x, gridDim.x);
}

// ---- weight gradients of one NRMS encoder: chunk partials -> the parameters' gradient buffers, one launch --------------------------
// The projection / pooling weight gradients come out of two batched library GEMMs as per-token-chunk fp32 partials in the PACKED
// operand geometry ([3*KP][KP] with the bias gradient in column D; [QP][KP] likewise), the query-vector gradient as one partial row
// per pooling workgroup.  autograd would sum the chunks (2 reductions), slice 9 strided views and add each into its .grad (9 more
// launches) plus the partial-row sum: 12 launches per encoder.  This kernel does all of it: one workgroup per output row sums the
// chunks and ACCUMULATES into the nine destinations (what AccumulateGrad does); summation order is fixed, so it is deterministic.
struct WgradUnpackParams {
  const float* dW;       // [ncW][3*KP][KP]  chunk partials of dqkv^T @ [X | 1]
  const float* dWa;      // [ncA][QP][KP]    chunk partials of dpre^T @ [ctx | 1]
  const float* dq;       // [nwg][QP]        per-workgroup partials of d attention_query_vector
  int ncW, ncA, qdim;
  int64_t nwg;
  float* gW[3];          // [D][D]   d W_Q / W_K / W_V .weight
  float* gb[3];          // [D]      ... .bias
  float* gWa;            // [qdim][D]
  float* gba;            // [qdim]
  float* gq;             // [qdim]
};
constexpr int WGU_ROWS = 2;                      // output rows per workgroup: 80 lanes x 4 packed columns each
constexpr int WGU_PG = 4;                        // partition groups per workgroup: group g sums partials g, g + 4, ...; combined through LDS
constexpr int WGU_LANES = WGU_ROWS * (KP / 4);   // 160 (row, column quad) slots
constexpr int WGU_THREADS = WGU_LANES * WGU_PG;  // 640
constexpr int WGU_SMEM = WGU_PG * WGU_LANES * 16;
constexpr int WGU_DQ_COLS = 16, WGU_DQ_PH = 32;  // dq kernel: 16 entries x 32 row phases per workgroup
__device__ __host__ __forceinline__ int wgrad_unpack_grid(int qdim) { return (3 * D + qdim + WGU_ROWS - 1) / WGU_ROWS; }

// VEC: all destinations are 16-byte aligned (float4 read-modify-write); otherwise element by element.
// (Round 3 gave every output quad ONE thread that walked all partials: with the 256 token partitions of the ring GEMM's pooling gradient that
// was a chain of 64 dependent trips on 15 k threads -- 93 us for 146 MB.  Four partition groups per quad, fixed combination order.)
template <bool VEC>
__global__ __launch_bounds__(WGU_THREADS) void wgrad_unpack_kernel(WgradUnpackParams p) {
  NR_SMEM_DECL(smem);
  f32x4* red = (f32x4*)smem;                                 // [WGU_PG][WGU_LANES]
  const int t = threadIdx.x % WGU_LANES, pg = threadIdx.x / WGU_LANES, rr = t / (KP / 4), c4 = t - rr * (KP / 4);
  const int b = blockIdx.x * WGU_ROWS + rr;                  // output row: [0, 3D) projections, [3D, 3D + qdim) pooling linear
  const bool live = b < 3 * D + p.qdim && c4 <= D / 4;       // columns > D are padding
  const bool proj = b < 3 * D;
  const int i = proj ? b / D : 0, r = proj ? b - i * D : b - 3 * D;
  f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    const float* src = (proj ? p.dW + ((size_t)(i * KP + r)) * KP : p.dWa + (size_t)r * KP) + c4 * 4;
    const size_t cs = proj ? (size_t)3 * KP * KP : (size_t)QP * KP;
    const int nc = proj ? p.ncW : p.ncA;
    // four independent partial sums: up to 4 x 16 B per lane in flight per trip (fixed order -> deterministic)
    f32x4 a0 = a, a1 = a, a2 = a, a3 = a;
    int c = pg;
    for (; c + 3 * WGU_PG < nc; c += 4 * WGU_PG) {
      a0 = a0 + *(const f32x4*)(src + (size_t)c * cs);
      a1 = a1 + *(const f32x4*)(src + (size_t)(c + WGU_PG) * cs);
      a2 = a2 + *(const f32x4*)(src + (size_t)(c + 2 * WGU_PG) * cs);
      a3 = a3 + *(const f32x4*)(src + (size_t)(c + 3 * WGU_PG) * cs);
    }
    for (; c < nc; c += WGU_PG) a0 = a0 + *(const f32x4*)(src + (size_t)c * cs);
    a = (a0 + a1) + (a2 + a3);
  }
  red[pg * WGU_LANES + t] = a;
  __syncthreads();
  if (pg != 0 || !live) return;
#pragma unroll
  for (int g = 1; g < WGU_PG; ++g) a = a + red[g * WGU_LANES + t];
  if (c4 == D / 4) {                                         // packed column D: the bias gradient
    float* dst = (proj ? p.gb[i] : p.gba) + r;
    *dst += a[0];
    return;
  }
  float* dst = (proj ? p.gW[i] : p.gWa) + (size_t)r * D + c4 * 4;
  if (VEC) {
    *(f32x4*)dst = *(const f32x4*)dst + a;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] += a[k];
  }
}

// d attention_query_vector: column sums of the per-workgroup partial rows, 16 columns per workgroup, 32 row phases combined through
// LDS in a fixed order.
__global__ __launch_bounds__(WGU_DQ_COLS * WGU_DQ_PH) void wgrad_dq_kernel(WgradUnpackParams p) {
  NR_SMEM_DECL(smem);
  float (*part)[WGU_DQ_COLS] = (float (*)[WGU_DQ_COLS])smem;          // [WGU_DQ_PH][WGU_DQ_COLS]
  const int t = threadIdx.x, cj = t % WGU_DQ_COLS, ph = t / WGU_DQ_COLS;
  const int j = blockIdx.x * WGU_DQ_COLS + cj;
  float a0 = 0.0f, a1 = 0.0f;
  if (j < p.qdim) {
    int64_t w = ph;
    for (; w + WGU_DQ_PH < p.nwg; w += 2 * WGU_DQ_PH) { a0 += p.dq[w * QP + j]; a1 += p.dq[(w + WGU_DQ_PH) * QP + j]; }
    if (w < p.nwg) a0 += p.dq[w * QP + j];
  }
  part[ph][cj] = a0 + a1;
  __syncthreads();
  if (ph == 0 && j < p.qdim) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < WGU_DQ_PH; ++k) s += part[k][cj];
    p.gq[j] += s;
  }
}

// ---- K7: dot-product scorers -------------------------------------------------------------------------------
// one wave per (b, c) pair; lanes stride the feature dim with float4 loads, shuffle-reduce.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
  return v;
}

__global__ __launch_bounds__(256) void score_dot_kernel(const float* __restrict__ cand, const float* __restrict__ user,
                                                        float* __restrict__ out, int64_t B, int C, int d4) {
  int64_t pair = (int64_t)blockIdx.x * 4 + wave_id();
  int64_t npairs = B * C;
  int l = lane_id();
  float acc = 0.0f;
  if (pair < npairs) {
    int64_t b = pair / C;
    const f32x4* cv = (const f32x4*)(cand + pair * d4 * 4);
    const f32x4* uv = (const f32x4*)(user + b * d4 * 4);
    for (int c = l; c < d4; c += 64) {
      f32x4 x = cv[c], y = uv[c];
      acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
  }
  acc = wave_sum(acc);
  if (pair < npairs && l == 0) out[pair] = acc;
}

// ragged (CSR) evaluation scorer: one wave per candidate slot; impression found by the block's binary search.
__global__ __launch_bounds__(256) void score_csr_kernel(const float* __restrict__ news, const float* __restrict__ users,
                                                        const int32_t* __restrict__ cand_idx, const int64_t* __restrict__ cand_ptr,
                                                        const int32_t* __restrict__ user_idx, float* __restrict__ out,
                                                        int64_t n_impr, int64_t nnz, int d4) {
  int64_t j = (int64_t)blockIdx.x * 4 + wave_id();
  int l = lane_id();
  float acc = 0.0f;
  bool live = j < nnz;
  if (live) {
    // largest i with cand_ptr[i] <= j
    int64_t lo = 0, hi = n_impr;
    while (hi - lo > 1) {
      int64_t mid = (lo + hi) >> 1;
      if (cand_ptr[mid] <= j) lo = mid; else hi = mid;
    }
    int32_t ni = cand_idx[j];
    if (ni >= 0) {
      const f32x4* cv = (const f32x4*)(news + (int64_t)ni * d4 * 4);
      const f32x4* uv = (const f32x4*)(users + (int64_t)user_idx[lo] * d4 * 4);
      for (int c = l; c < d4; c += 64) {
        f32x4 x = cv[c], y = uv[c];
        acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
      }
    }
  }
  acc = wave_sum(acc);
  if (live && l == 0) out[j] = acc;
}

// ---- dropout mask export (verification of the fused kernels' RNG) -------------------------------------------
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, int64_t n_elem, DropCfg dc, int site) {
  dc = drop_resolve(dc);
  int64_t nquad = (n_elem + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * blockDim.x) {
    uint32_t m = drop_keep4(dc, (uint32_t)site, (uint64_t)q);
    for (int j = 0; j < 4; ++j)
      if (q * 4 + j < n_elem) mask[q * 4 + j] = ((m >> j) & 1u) ? 1.0f : 0.0f;
  }
}

// ---- MFMA layout probe ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void probe_mfma_kernel(const u16* __restrict__ A, const u16* __restrict__ B, float* __restrict__ Dm) {
  int l = lane_id();
  int g = l >> 4, i = l & 15;
  u16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[i * 32 + g * 8 + j];          // A[i][k]
    b[j] = B[(g * 8 + j) * 16 + i];        // B[k][n=i]
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = mfma_16x16x32_bf16(a, b, c);
  for (int r = 0; r < 4; ++r) Dm[(g * 4 + r) * 16 + i] = c[r];
}

// ---- LDS transpose-read probe (ds_read_b64_tr_b16): LDS holds lds[i] = i (u16, 4096 entries); lane l reads the piece at byte offset
// offs[l] (8-byte aligned) and stores the four elements it receives: the parity tests hold the result to the semantics documented at
// lds_tr16_b64 (nr_prims.h), which the weight-gradient kernel's operand loads rely on ------------------------------------------------
__global__ __launch_bounds__(64) void probe_tr16_kernel(const int32_t* __restrict__ offs, u16* __restrict__ out) {
  NR_SMEM_DECL(smem);
  u16* lds = (u16*)smem;
  const int l = lane_id();
  for (int i = l; i < 4096; i += 64) lds[i] = (u16)i;
  __syncthreads();
  const u16x4 v = lds_tr16_b64(lds + (offs[l] >> 1));
  *(u16x4*)(out + l * 4) = v;
}

}  // namespace nr
