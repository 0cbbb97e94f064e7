// Additive-attention pooling forward for gfx950 (src/model/general/attention/additive.py:27-53):
//   out[t,:] = sum_s softmax_s( tanh(x[t,s,:] Wa^T + ba) . qv ) * x[t,s,:]
// x arrives as the bf16 ctx tile written by the MHSA kernel ([tokens][KP], K-padded with zeros).
// Workgroup = 4 waves, NSEQ sequences of S tokens staged in LDS (same 656-B row stride as the MHSA kernel).
// The [tokens,300]x[300,200] projection runs on MFMA as the TRANSPOSED product (A = Wa rows from L2 held in
// registers, B = x^T fragments from LDS): a lane then holds 4 query-dim rows of one token, so tanh * qv is
// reduced in-lane, across the 4 lane groups with two xor-shuffles, and per wave into an LDS partial-score
// row (no atomics -> deterministic).  Softmax over the S tokens is one wave per sequence; the weighted sum
// reads the LDS tile once.
#pragma once
#include "nr_common.h"
#include "k_mhsa_fwd.h"

namespace nr {

template <int S, int NSEQ, int NW = 4>
struct AddGeom {
  static constexpr int THREADS = NW * 64;
  static constexpr int TOK = S * NSEQ;
  static constexpr int MT = (TOK + 15) / 16;
  static constexpr int ROWS = MT * 16;
  static constexpr int X_BYTES = ROWS * XS * 2;
  static constexpr int SC_BYTES = NW * ROWS * 4;    // per-wave partial scores
  static constexpr int W_BYTES = ROWS * 4;          // softmax weights
  static constexpr int SMEM = X_BYTES + SC_BYTES + W_BYTES;
  static constexpr int NTQ = QP / 16;               // 13 n-tiles of the query dim
  static constexpr int DQ_FLOATS = NW * QP > 3 * ROWS ? NW * QP : 3 * ROWS;     // dq partials, also the 3 x ROWS scratch of the dw pass
  static constexpr int G_BYTES = NSEQ * D * 4;      // backward: the workgroup's g_out rows (read 25x per token by the softmax-backward pass)
  static constexpr int BWD_SMEM = X_BYTES + DQ_FLOATS * 4 + ROWS * 4 + G_BYTES;   // backward: tile + dq partials + ds + g_out rows
};

struct AdditiveParams {
  const u16* ctx;      // [n_seq*S][KP]
  const u16* Wap;      // [QP][KP]
  const float* bap;    // [QP]
  const float* qvp;    // [QP]
  float* out;          // [n_seq][out_stride] (first D columns of each row) or null
  int64_t out_stride;  // row stride of out in floats (D when the pooled vectors are dense)
  u16* out_b;          // optional bf16 copy in the ctx layout: row i at out_b + i*out_b_stride, cols 0..D-1, col D = 1.0, rest 0
  int64_t out_b_stride;
  float* attn_w;       // [n_seq][S] or null
  int64_t n_seq;
  int valid;           // tokens s >= valid of every sequence are excluded from the softmax (weight exactly 0); 0 or >= S: all S tokens.
                       // Sequences shorter than an instantiated S are zero-padded by the host and pooled with valid = their length
};

// parameters of the pooling backward kernels (k_bwd.h additive_bwd_kernel, k_pool2.h pool2_bwd_kernel)
struct AdditiveBwdParams {
  const u16* ctx;        // [n_seq*S][KP]  (forward input of the additive layer)
  const u16* Wap;        // [QP][KP]
  const float* bap;      // [QP]
  const float* qvp;      // [QP]
  const float* attn_w;   // [n_seq][S]
  const float* g_out;    // [n_seq][D]
  u16* dpre;             // [n_seq*S][QP] bf16
  float* dq_part;        // [gridDim.x][QP]  per-workgroup partial gradient of the query vector
  const u16* WaT;        // optional: bf16 [KP][QKP] = Wa^T (pack_additive_t) -> the kernel also emits dctx = dpre @ Wa
  u16* dctx;             // optional: bf16 [n_seq*S][KP] (columns < D written)
  int64_t n_seq;
  // register-resident kernels (k_pool2.h) only, instead of dctx: the gradient of the conv + relu + dropout stage that produced ctx,
  // dy_pad[seqpad row of tok] = (dpre @ Wa + attn_w (x) g_out) * [ctx != 0] * act_scale  (what conv_act_bwd_kernel computes from dctx)
  u16* dy_pad;           // optional: bf16 seqpad rows [(tok + seq + 1)][KP] (columns < D written; separators / padding stay the caller's zeros)
  float act_scale;       // 1 / (1 - p_drop)
};

// The pooling of a workgroup's NSEQ sequences once their ctx rows sit in LDS (Xs: [ROWS][XS] bf16, col D = 1.0, cols > D zero): scores,
// softmax, weighted sum.  sc ([NW][ROWS] floats) and wl ([ROWS] floats) are LDS scratch; every wave of the workgroup must call it (it
// synchronises the workgroup), whatever it did before.  Shared by additive_fwd_kernel (tile staged from HBM) and by the pooled form of
// attn_fwd_kernel (k_proj.h: the tile is the attention output the workgroup has just produced).
template <int S, int NSEQ, int NW>
__device__ __forceinline__ void additive_pool_tile(const AdditiveParams& p, const u16* Xs, float* sc, float* wl, int64_t seq0) {
  using Gm = AddGeom<S, NSEQ, NW>;
  constexpr int WG = NW * 64;          // shadows nr::WG: 4 or 8 waves
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  for (int i = tid; i < NW * Gm::ROWS; i += WG) sc[i] = 0.0f;
  __syncthreads();

  // ---- scores: sum_n tanh(x.Wa[n] + ba[n]) * qv[n] ---------------------------------------------------------
  const int w_eff = (w + (int)blockIdx.x) % NW;
  float* myrow = sc + w * Gm::ROWS;
  for (int cg = 0; cg < (Gm::NTQ + 1) / 2; ++cg) {
    int G, mb, me;
    unit_range(Gm::NTQ, Gm::MT, w_eff, NW, cg, G, mb, me);
    if (mb >= me) continue;
    // query-vector rows of this column pair: loaded once here, in the same round trip as the weight fragments (inside the epilogue
    // the load could not be hoisted over the LDS / global stores of the token-tile loop and cost an L2 round trip per tile)
    const int wrq0 = (2 * cg) * 16, wrq1 = G == 2 ? wrq0 + 16 : wrq0;
    const f32x4 qv2[2] = {*(const f32x4*)(p.qvp + wrq0 + 4 * g), *(const f32x4*)(p.qvp + wrq1 + 4 * g)};
    auto epi = [&](int wr, int m, f32x4 acc) {          // acc already holds x.Wa[n] + ba[n] (bias = accumulator init)
      const f32x4 q4 = wr == wrq0 ? qv2[0] : qv2[1];
      float s = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) s += fast_tanh(acc[r]) * q4[r];
      s += shfl_xor(s, 16);
      s += shfl_xor(s, 32);
      if (g == 0) myrow[m * 16 + li] += s;     // lanes 0..15 -> 16 distinct tokens; same wave only
    };
    if (G == 2) {
      int wrow[2] = {(2 * cg) * 16, (2 * cg + 1) * 16};
      const f32x4 binit[2] = {*(const f32x4*)(p.bap + wrow[0] + 4 * g), *(const f32x4*)(p.bap + wrow[1] + 4 * g)};
      proj_block<2, true>(p.Wap, wrow, Xs, mb, me, binit, [&](int j, int m, f32x4 acc) { epi(wrow[j], m, acc); });
    } else {
      int wrow[1] = {(2 * cg) * 16};
      const f32x4 binit[1] = {*(const f32x4*)(p.bap + wrow[0] + 4 * g)};
      proj_block<1, true>(p.Wap, wrow, Xs, mb, me, binit, [&](int j, int m, f32x4 acc) { epi(wrow[0], m, acc); });
    }
  }
  __syncthreads();

  // ---- softmax over the S tokens of each sequence (one wave per sequence) ---------------------------------
  for (int seq = w; seq < NSEQ; seq += NW) {
    const int r = seq * S + l;
    const bool live = l < ((p.valid > 0 && p.valid < S) ? p.valid : S);
    float v = -3.0e38f;
    if (live) {
      v = 0.0f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) v += sc[ww * Gm::ROWS + r];
    }
    float mx = v;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, shfl_xor(mx, m));
    float e = live ? fast_exp(v - mx) : 0.0f;
    float sum = e;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += shfl_xor(sum, m);
    float wt = e / sum;
    if (l < S) {                              // masked tokens (valid <= l < S) get weight 0: they drop out of the sum and of every gradient
      wl[r] = wt;
      if (p.attn_w != nullptr && seq0 + seq < p.n_seq) p.attn_w[(seq0 + seq) * S + l] = wt;
    }
  }
  __syncthreads();

  // ---- weighted sum: out[seq][c] = sum_s w[s] x[s][c] ------------------------------------------------------
  for (int i = tid; i < NSEQ * D4; i += WG) {
    const int seq = i / D4, c = i - seq * D4;
    if (seq0 + seq >= p.n_seq) continue;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const u16* xp = Xs + (seq * S) * XS + c * 4;
#pragma unroll 5
    for (int s = 0; s < S; ++s) {
      u16x4 x = *(const u16x4*)(xp + s * XS);
      float wt = wl[seq * S + s];
      acc[0] += wt * bf2f(x[0]); acc[1] += wt * bf2f(x[1]); acc[2] += wt * bf2f(x[2]); acc[3] += wt * bf2f(x[3]);
    }
    if (p.out != nullptr) *(f32x4*)(p.out + (seq0 + seq) * p.out_stride + c * 4) = acc;
    if (p.out_b != nullptr) *(u16x4*)(p.out_b + (seq0 + seq) * p.out_b_stride + c * 4) = pack4(acc);
  }
  if (p.out_b != nullptr) {
    constexpr int PADQ = (KP - D) / 4;
    for (int i = tid; i < NSEQ * PADQ; i += WG) {
      const int seq = i / PADQ, c = i - seq * PADQ;
      if (seq0 + seq < p.n_seq)
        *(u16x4*)(p.out_b + (seq0 + seq) * p.out_b_stride + D + c * 4) = u16x4{(u16)(c == 0 ? 0x3F80 : 0), 0, 0, 0};
    }
  }
}

template <int S, int NSEQ, int NW = 4>
__global__ __launch_bounds__(NW * 64) void additive_fwd_kernel(AdditiveParams p) {
  using Gm = AddGeom<S, NSEQ, NW>;
  constexpr int WG = NW * 64;
  NR_SMEM_DECL(smem);
  u16* Xs = (u16*)smem;
  float* sc = (float*)(smem + Gm::X_BYTES);            // [NW][ROWS]
  float* wl = (float*)(smem + Gm::X_BYTES + Gm::SC_BYTES);
  const int tid = threadIdx.x;
  const int64_t seq0 = (int64_t)blockIdx.x * NSEQ;
  const int64_t tok0 = seq0 * S, tok_total = p.n_seq * S;

  // ---- stage the ctx tile (16-B pieces, 41 per LDS row incl. the stride padding) ------------------------
  constexpr int PCS = XS / 8;   // 41 16-B pieces per LDS row
  for (int i = tid; i < Gm::ROWS * PCS; i += WG) {
    int r = i / PCS, c = i - r * PCS;
    u16x8 v = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (c < KP / 8 && r < Gm::TOK && tok0 + r < tok_total) v = *(const u16x8*)(p.ctx + (tok0 + r) * KP + c * 8);
    *(u16x8*)(Xs + r * XS + c * 8) = v;
  }
  additive_pool_tile<S, NSEQ, NW>(p, Xs, sc, wl, seq0);
}

}  // namespace nr
