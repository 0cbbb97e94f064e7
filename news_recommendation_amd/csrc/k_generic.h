// GENERAL-GEOMETRY kernels: the model-geometry knobs of src/config.py away from the values the tuned kernels are instantiated for
// (word_embedding_dim 300, num_attention_heads 15, num_filters 300, window_size 3, query_vector_dim <= 208: everything in k_proj.h, k_conv.h,
// k_pool3.h ... is templated on those).  The reference builds ANY word_embedding_dim, any num_attention_heads that divides it, any num_filters
// and any odd window_size (src/config.py:34,45,54,55; model/general/attention/multihead_self.py:27-38; model/NAML/news_encoder.py:10-19;
// model/LSTUR/news_encoder.py:23).  This file is the engine's path for those: every dense contraction (the Q / K / V projections, the additive
// projection, the convolution as ONE GEMM over overlapping rows of a seqpad buffer, all weight and input gradients) runs in the general ring
// GEMMs of k_gemm.h on bf16 operands with fp32 accumulation -- the same numerics as the tuned path --, and the kernels below do what is not a
// GEMM: dropout, the attention core exp(QK^T / sqrt(d_k)) / (sum + 1e-8) V per (sequence, head) and its backward, the additive pooling's
// tanh / softmax / weighted sum and its backward, relu + dropout of the conv stage.  They are written for CORRECTNESS at any geometry (fp32
// VALU arithmetic, one wave per (sequence, head) / one workgroup per sequence, operands staged in LDS), not tuned: 300 / 15 / 300 / 3 stays
// on the tuned kernels (ops.py / ops_conv.py dispatch).  Limits: sequence length <= 64, d_k <= 32, dims multiples of 4.
#pragma once
#include "nr_common.h"

namespace nr {

constexpr int G_SMAX = 64;      // tokens per sequence (one lane per query / key)
constexpr int G_DKMAX = 32;     // features per head

// (wave_sum: k_misc.h)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const float o = shfl_xor(v, m); v = o > v ? o : v; }
  return v;
}

// y[i] = x[i] * keep(site, quad0 * 4 + i) / (1 - p) over n4 quads of consecutive f32 elements (F.dropout, news_encoder.py:38-40,43-45 of every
// model; its own backward: the same mask on the gradient).  The element numbering is nr_dropout_mask's, so the masks are exportable.
__global__ __launch_bounds__(256) void g_dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4, int64_t quad0, DropCfg dc,
                                                        int site) {
  dc = drop_resolve(dc);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = *(const f32x4*)(x + i * 4);
    const f32x4 m = drop_mul4(dc, (uint32_t)site, (uint64_t)(quad0 + i));
    *(f32x4*)(y + i * 4) = f32x4{v[0] * m[0], v[1] * m[1], v[2] * m[2], v[3] * m[3]};
  }
}

// ---- ScaledDotProductAttention (multihead_self.py:15-23) per (sequence, head): ctx = exp(Q K^T / sqrt(d_k)) / (sum + 1e-8) V, no max
// subtraction (the reference has none; the argument is clamped at EXP_CLAMP like in the tuned kernels), keys >= key_len[seq] masked (:60-70).
// qkv f32 [n_seq * S][ld]: Q at column 0, K at column D, V at column 2 D, head h at columns h * dk.  One wave per (sequence, head): lane =
// query position, K and V of the head staged in LDS.
struct GAttnParams {
  const float* qkv; int64_t ld;
  const float* dctx;     // backward: upstream gradient of ctx, f32 [n_seq * S][D]
  float* out;            // forward: ctx f32 [n_seq * S][D];  backward: dqkv f32 [n_seq * S][ld]
  const int* key_len;    // [n_seq] or null
  int64_t n_seq; int S, H, dk, D;
};

__global__ __launch_bounds__(64) void g_attn_fwd_kernel(GAttnParams p) {
  NR_SMEM_DECL(smem);
  float* Ks = (float*)smem;                  // [S][dk]
  float* Vs = Ks + p.S * p.dk;
  const int64_t seq = blockIdx.x / p.H;
  const int h = (int)(blockIdx.x % p.H), l = lane_id();
  const float* base = p.qkv + seq * p.S * p.ld + h * p.dk;
  for (int e = l; e < p.S * p.dk; e += 64) {
    const int j = e / p.dk, d = e - j * p.dk;
    Ks[e] = base[(int64_t)j * p.ld + p.D + d];
    Vs[e] = base[(int64_t)j * p.ld + 2 * p.D + d];
  }
  __syncthreads();
  const int kl = p.key_len ? clamp_len(p.key_len[seq], p.S) : p.S;
  if (l >= p.S) return;
  float q[G_DKMAX], acc[G_DKMAX];
#pragma unroll
  for (int d = 0; d < G_DKMAX; ++d) { q[d] = d < p.dk ? base[(int64_t)l * p.ld + d] : 0.0f; acc[d] = 0.0f; }
  const float scale = 1.0f / sqrtf((float)p.dk);
  float Z = 0.0f;
  for (int j = 0; j < kl; ++j) {
    float s = 0.0f;
#pragma unroll
    for (int d = 0; d < G_DKMAX; ++d) if (d < p.dk) s += q[d] * Ks[j * p.dk + d];
    s *= scale;
    const float e = expf(s < EXP_CLAMP ? s : EXP_CLAMP);
    Z += e;
#pragma unroll
    for (int d = 0; d < G_DKMAX; ++d) if (d < p.dk) acc[d] += e * Vs[j * p.dk + d];
  }
  const float inv = 1.0f / (Z + 1e-8f);
  float* o = p.out + (seq * p.S + l) * p.D + h * p.dk;
#pragma unroll
  for (int d = 0; d < G_DKMAX; ++d) if (d < p.dk) o[d] = acc[d] * inv;
}

// Its backward (autograd of multihead_self.py:15-23,53-75 up to the projections): with a_ij = e_ij / (Z_i + 1e-8), g_ij = dC_i . V_j and
// r_i = sum_j a_ij g_ij:  dS_ij = a_ij (g_ij - r_i);  dQ_i = sum_j dS_ij K_j / sqrt(d_k);  dK_j = sum_i dS_ij Q_i / sqrt(d_k);  dV_j = sum_i a_ij dC_i.
// Phase 1: lane = query (Z_i, r_i, dQ_i); phase 2: lane = key (dK_j, dV_j); Q, K, V, dC of the head in LDS.
__global__ __launch_bounds__(64) void g_attn_bwd_kernel(GAttnParams p) {
  NR_SMEM_DECL(smem);
  const int n = p.S * p.dk;
  float* Qs = (float*)smem;
  float* Ks = Qs + n;
  float* Vs = Ks + n;
  float* Cs = Vs + n;
  float* Zi = Cs + n;                        // [S] 1 / (Z_i + 1e-8)
  float* Ri = Zi + G_SMAX;                   // [S] r_i
  const int64_t seq = blockIdx.x / p.H;
  const int h = (int)(blockIdx.x % p.H), l = lane_id();
  const float* base = p.qkv + seq * p.S * p.ld + h * p.dk;
  const float* cb = p.dctx + seq * p.S * p.D + h * p.dk;
  for (int e = l; e < n; e += 64) {
    const int j = e / p.dk, d = e - j * p.dk;
    Qs[e] = base[(int64_t)j * p.ld + d];
    Ks[e] = base[(int64_t)j * p.ld + p.D + d];
    Vs[e] = base[(int64_t)j * p.ld + 2 * p.D + d];
    Cs[e] = cb[(int64_t)j * p.D + d];
  }
  __syncthreads();
  const int kl = p.key_len ? clamp_len(p.key_len[seq], p.S) : p.S;
  const float scale = 1.0f / sqrtf((float)p.dk);
  float* ob = p.out + seq * p.S * p.ld + h * p.dk;
  if (l < p.S) {
    float Z = 0.0f, eg = 0.0f;
    for (int j = 0; j < kl; ++j) {
      float s = 0.0f, g = 0.0f;
      for (int d = 0; d < p.dk; ++d) { s += Qs[l * p.dk + d] * Ks[j * p.dk + d]; g += Cs[l * p.dk + d] * Vs[j * p.dk + d]; }
      s *= scale;
      const float e = expf(s < EXP_CLAMP ? s : EXP_CLAMP);
      Z += e; eg += e * g;
    }
    const float inv = 1.0f / (Z + 1e-8f), r = eg * inv;
    Zi[l] = inv; Ri[l] = r;
    float dq[G_DKMAX];
#pragma unroll
    for (int d = 0; d < G_DKMAX; ++d) dq[d] = 0.0f;
    for (int j = 0; j < kl; ++j) {
      float s = 0.0f, g = 0.0f;
      for (int d = 0; d < p.dk; ++d) { s += Qs[l * p.dk + d] * Ks[j * p.dk + d]; g += Cs[l * p.dk + d] * Vs[j * p.dk + d]; }
      s *= scale;
      const float ds = expf(s < EXP_CLAMP ? s : EXP_CLAMP) * inv * (g - r) * scale;
#pragma unroll
      for (int d = 0; d < G_DKMAX; ++d) if (d < p.dk) dq[d] += ds * Ks[j * p.dk + d];
    }
#pragma unroll
    for (int d = 0; d < G_DKMAX; ++d) if (d < p.dk) ob[(int64_t)l * p.ld + d] = dq[d];
  }
  __syncthreads();
  if (l < p.S) {
    float dk_[G_DKMAX], dv[G_DKMAX];
#pragma unroll
    for (int d = 0; d < G_DKMAX; ++d) { dk_[d] = 0.0f; dv[d] = 0.0f; }
    if (l < kl) {
      for (int i = 0; i < p.S; ++i) {
        float s = 0.0f, g = 0.0f;
        for (int d = 0; d < p.dk; ++d) { s += Qs[i * p.dk + d] * Ks[l * p.dk + d]; g += Cs[i * p.dk + d] * Vs[l * p.dk + d]; }
        s *= scale;
        const float a = expf(s < EXP_CLAMP ? s : EXP_CLAMP) * Zi[i];
        const float ds = a * (g - Ri[i]) * scale;
#pragma unroll
        for (int d = 0; d < G_DKMAX; ++d) if (d < p.dk) { dk_[d] += ds * Qs[i * p.dk + d]; dv[d] += a * Cs[i * p.dk + d]; }
      }
    }
#pragma unroll
    for (int d = 0; d < G_DKMAX; ++d) if (d < p.dk) { ob[(int64_t)l * p.ld + p.D + d] = dk_[d]; ob[(int64_t)l * p.ld + 2 * p.D + d] = dv[d]; }
  }
}

// ---- AdditiveAttention (additive.py:27-53) on f32 rows: proj = x Wa^T + ba comes from the GEMM; score_t = q . tanh(proj_t), softmax over the
// first `valid` tokens (F.softmax: max-subtracted), out = sum_t w_t x_t.  One workgroup of 4 waves per sequence.
struct GPoolParams {
  const float* x; int64_t ldx; int D;        // [n_seq * S][ldx]
  const float* proj; int64_t ldp; int Q;     // [n_seq * S][ldp], bias included
  const float* qv;                           // [Q]
  const float* g_out; int64_t ldg;           // backward: gradient of out, [n_seq][ldg]
  float* out; int64_t ldo;                   // forward: [n_seq][ldo]
  float* attn_w;                             // [n_seq][S] (forward: written; backward: read)
  float* dpre; int64_t ldq;                  // backward: [n_seq * S][ldq] gradient of proj
  float* dq_part;                            // backward: [n_seq][Q]
  int64_t n_seq; int S, valid;
};

__global__ __launch_bounds__(256) void g_additive_fwd_kernel(GPoolParams p) {
  NR_SMEM_DECL(smem);
  float* sc = (float*)smem;                  // [G_SMAX]
  const int64_t seq = blockIdx.x;
  const int l = lane_id(), w = wave_id();
  for (int t = w; t < p.S; t += 4) {
    const float* pr = p.proj + (seq * p.S + t) * p.ldp;
    float s = 0.0f;
    for (int q = l; q < p.Q; q += 64) s += p.qv[q] * tanhf(pr[q]);
    s = wave_sum(s);
    if (l == 0) sc[t] = s;
  }
  __syncthreads();
  if (w == 0) {
    const bool on = l < p.valid && l < p.S;
    const float s = on ? sc[l] : -3.0e38f;
    const float m = wave_max(s);
    const float e = on ? expf(s - m) : 0.0f;
    const float z = wave_sum(e);
    const float wt = e / z;
    if (l < p.S) { sc[l] = wt; if (p.attn_w) p.attn_w[seq * p.S + l] = wt; }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < p.D; d += 256) {
    float a = 0.0f;
    for (int t = 0; t < p.S; ++t) a += sc[t] * p.x[(seq * p.S + t) * p.ldx + d];
    p.out[seq * p.ldo + d] = a;
  }
}

// backward up to the projection: dw_t = g . x_t, ds_t = w_t (dw_t - sum_s w_s dw_s), dpre[t][q] = ds_t q_q (1 - tanh^2), dq[q] = sum_t ds_t tanh
// (the caller adds the direct term w_t g to dpre @ Wa: g_rows_axpy_kernel)
__global__ __launch_bounds__(256) void g_additive_bwd_kernel(GPoolParams p) {
  NR_SMEM_DECL(smem);
  float* ds = (float*)smem;                  // [G_SMAX]
  const int64_t seq = blockIdx.x;
  const int l = lane_id(), w = wave_id();
  const float* g = p.g_out + seq * p.ldg;
  for (int t = w; t < p.S; t += 4) {
    const float* xr = p.x + (seq * p.S + t) * p.ldx;
    float s = 0.0f;
    for (int d = l; d < p.D; d += 64) s += g[d] * xr[d];
    s = wave_sum(s);
    if (l == 0) ds[t] = s;
  }
  __syncthreads();
  if (w == 0) {
    const float wt = l < p.S ? p.attn_w[seq * p.S + l] : 0.0f;
    const float dw = l < p.S ? ds[l] : 0.0f;
    const float tot = wave_sum(wt * dw);
    if (l < p.S) ds[l] = wt * (dw - tot);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < p.Q; q += 256) {
    const float qq = p.qv[q];
    float dq = 0.0f;
    for (int t = 0; t < p.S; ++t) {
      const float th = tanhf(p.proj[(seq * p.S + t) * p.ldp + q]);
      p.dpre[(seq * p.S + t) * p.ldq + q] = ds[t] * qq * (1.0f - th * th);
      dq += ds[t] * th;
    }
    p.dq_part[seq * p.Q + q] = dq;
  }
}

// y[row][0:d] (+)= a[row] * g[row / S][0:d]: the direct term w_t g_out of a pooling level's input gradient added to the GEMM part
__global__ __launch_bounds__(256) void g_rows_axpy_kernel(float* __restrict__ y, int64_t ldy, const float* __restrict__ a, const float* __restrict__ g,
                                                          int64_t ldg, int S, int d, int64_t n_rows, int accumulate) {
  const int64_t n = n_rows * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / d;
    const int c = (int)(i - r * d);
    const float v = a[r] * g[(r / S) * ldg + c];
    y[r * ldy + c] = accumulate ? y[r * ldy + c] + v : v;
  }
}

// ---- seqpad rows for a convolution of any odd window w (pad = (w - 1) / 2 zero rows between sequences and at both ends): token s of
// sequence q at row pad + q (S + pad) + s.  The window of a token is then w CONTIGUOUS rows, so Conv2d(1, F, (w, D)) (NAML / LSTUR
// news_encoder.py) is one NT GEMM whose A operand is the buffer itself with row stride Dp and K = w Dp (overlapping rows).
// src f32 [n_seq * S][lds] -> dst bf16 rows of dp elements: columns < d copied, column d = `one` (1.0 for the bias trick, 0 for gradients), rest 0.
__global__ __launch_bounds__(256) void g_rows_to_seqpad_kernel(const float* __restrict__ src, int64_t lds, int d, u16* __restrict__ dst, int dp, int S,
                                                               int pad, int64_t n_tok, u16 one) {
  const int64_t n = n_tok * dp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / dp;
    const int c = (int)(i - t * dp);
    const int64_t q = t / S;
    const int64_t row = pad + q * (S + pad) + (t - q * S);
    dst[row * dp + c] = c < d ? f2bf(src[t * lds + c]) : (c == d ? one : (u16)0);
  }
}

// act[t][f] = dropout(relu(y[row(t)][f])) (news_encoder.py: F.dropout(F.relu(conv))) from the GEMM result y f32 [virtual rows][ldy] (virtual
// row q (S + pad) + s: the bias rode the contraction); token layout out.  quad0: dropout counter of the first element (site 2).
__global__ __launch_bounds__(256) void g_relu_drop_kernel(const float* __restrict__ y, int64_t ldy, float* __restrict__ act, int F_, int S, int pad,
                                                          int64_t n_tok, DropCfg dc, int64_t quad0) {
  dc = drop_resolve(dc);
  const int f4 = F_ >> 2;
  const int64_t n = n_tok * f4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / f4;
    const int c = (int)(i - t * f4) * 4;
    const int64_t q = t / S;
    const f32x4 v = *(const f32x4*)(y + (q * (S + pad) + (t - q * S)) * ldy + c);
    f32x4 m = f32x4{1.f, 1.f, 1.f, 1.f};
    if (dc.enabled) m = drop_mul4(dc, 2u, (uint64_t)(quad0 + i));
    *(f32x4*)(act + t * F_ + c) = f32x4{v[0] > 0.f ? v[0] * m[0] : 0.f, v[1] > 0.f ? v[1] * m[1] : 0.f, v[2] > 0.f ? v[2] * m[2] : 0.f, v[3] > 0.f ? v[3] * m[3] : 0.f};
  }
}

// dy[row(t)][f] = dact[t][f] * [act[t][f] != 0] * scale as bf16 seqpad rows of fp elements (separator rows and padding columns stay zero: the
// buffer is zero-filled once): the operand of the tap-gradient GEMM and of the data-gradient GEMM
__global__ __launch_bounds__(256) void g_relu_drop_bwd_kernel(const float* __restrict__ dact, const float* __restrict__ act, u16* __restrict__ dy, int F_,
                                                              int fp, int S, int pad, int64_t n_tok, float scale) {
  const int64_t n = n_tok * F_;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / F_;
    const int c = (int)(i - t * F_);
    const int64_t q = t / S;
    dy[(pad + q * (S + pad) + (t - q * S)) * fp + c] = f2bf(act[i] != 0.0f ? dact[i] * scale : 0.0f);
  }
}

// dst[t][0:d] = src[row(t)][0:d]: virtual GEMM rows (q (S + pad) + s) back to token layout
__global__ __launch_bounds__(256) void g_unpad_rows_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int d, int S, int pad,
                                                           int64_t n_tok) {
  const int64_t n = n_tok * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / d;
    const int c = (int)(i - t * d);
    const int64_t q = t / S;
    dst[i] = src[(q * (S + pad) + (t - q * S)) * lds + c];
  }
}

// f32 rows -> bf16 rows [hi | hi | lo] of 3 dp columns (hi = bf16(x), lo = bf16(x - hi); column d of the two hi blocks = 1.0): against packed
// weights [Wh | Wl | Wh] one bf16 GEMM with K = 3 dp evaluates x W^T + b to ~2^-16 relative (xh Wh + xh Wl + xl Wh) -- for the few products
// whose SIGN decides a relu (the ElementEncoder's linear layer over the category table: a flipped unit changes a whole row of its weight gradient)
__global__ __launch_bounds__(256) void g_rows_split_kernel(const float* __restrict__ src, int64_t lds_, int d, u16* __restrict__ dst, int dp, int64_t n) {
  const int64_t total = n * dp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / dp;
    const int c = (int)(i - r * dp);
    u16 hi = 0, lo = 0;
    if (c < d) {
      const float x = src[r * lds_ + c];
      hi = f2bf(x);
      lo = f2bf(x - bf2f(hi));
    } else if (c == d) {
      hi = 0x3F80;
    }
    u16* o = dst + r * 3 * dp + c;
    o[0] = hi; o[dp] = hi; o[2 * dp] = lo;
  }
}

// y = scale * x where gate > 0 (gate = x when null), else 0: relu forward (scale 1), and the backward of dropout(relu(.)) read off the OUTPUT's
// zeros (gate = the activation, scale = 1 / (1 - p))
__global__ __launch_bounds__(256) void g_relu_kernel(const float* __restrict__ x, const float* __restrict__ gate, float* __restrict__ y, int64_t n,
                                                     float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (gate ? gate[i] > 0.0f : x[i] > 0.0f) ? x[i] * scale : 0.0f;
}

}  // namespace nr
