// Additive-attention pooling BACKWARD over a FLAT token stream, gfx950 (src/model/general/attention/additive.py:27-53 and its autograd).
// Same outputs as the sequence-shaped kernels it replaces (dpre, dq partials, dctx = dpre @ Wa or the fused activation gradient dy_pad):
//
//   dw[tok]   = g_out[seq] . x[tok]
//   ds[tok]   = w[tok] (dw[tok] - sum_s w[s] dw[s])                       softmax backward
//   dpre[tok] = ds[tok] qv (1 - t^2),  t = tanh(x[tok] Wa^T + ba);   dq += ds[tok] t;   dctx[tok] = dpre[tok] @ Wa
//
// What made the older kernels sequence-shaped (a wave owned whole sequences: 40 of 48 rows used at S = 20, 50 of 64 at S = 50, the latter
// at one wave per SIMD) is the sum inside ds.  It does not need the sequence:
//       sum_s w[s] dw[s] = g_out[seq] . (sum_s w[s] x[s]) = g_out[seq] . y[seq]
// with y the pooled vector the FORWARD already produced from the same bf16 rows and the same weights.  A tiny kernel (rowdot_kernel) turns
// it into one scalar per sequence, and every token row becomes independent: rows are dealt to waves 48 at a time regardless of sequence
// boundaries, for any S.
//
// With no per-sequence phase left there is nothing to synchronise: the kernel is PERSISTENT (one workgroup of 8 waves per CU), Wa sits in
// LDS for its whole life (208 rows x 656 B, loaded once) and serves BOTH products -- x Wa^T through plain 16-byte fragment reads, dpre @ Wa
// through the transposing read ds_read_b64_tr_b16 of the same rows (no Wa^T operand, no weight streaming, no barrier inside the loop).
// A wave's 48 ctx rows live in registers as MFMA B fragments (as in k_pool2.h); the next group's rows are requested as soon as the
// projection has consumed the current ones, so their latency hides behind the dctx product; the two waves of a SIMD drift apart and
// overlap each other's load / MFMA / tanh / store phases.
#pragma once
#include "nr_common.h"
#include "k_additive_fwd.h"

namespace nr {

template <int MT_>
struct Pool3GeomT {
  static constexpr int NWAVE = 8, THREADS = NWAVE * 64;
  static constexpr int MT = MT_, ROWS = MT * 16;   // token rows per wave and iteration
  static constexpr int NTQ = QP / 16;            // 13 n-tiles of the query dim
  static constexpr int KS2 = QKP / 32;           // 7 k-steps of the dctx product
  static constexpr int NTD = (D + 15) / 16;      // 19 feature tiles of dctx
  static constexpr int WROW = XS * 2;            // 656 B per Wa row: conflict-free b128 fragment reads and transposing reads
  static constexpr int W_BYTES = QP * WROW;      // 136,448
  static constexpr int BQ_BYTES = 2 * QP * 4;    // bias and query vector
  static constexpr int DQ_BYTES = NWAVE * QP * 4;
  static constexpr int SMEM = W_BYTES + BQ_BYTES + DQ_BYTES;      // 144,768
  static_assert(SMEM <= 163840, "LDS");
};
using Pool3Geom = Pool3GeomT<3>;

struct Pool3Params {
  const u16* ctx;        // [n_tok][KP]  forward input of the additive layer
  const u16* Wap;        // [QP][KP] tile order
  const float* bap;      // [QP]
  const float* qvp;      // [QP]
  const float* attn_w;   // [n_tok]  forward attention weights
  const float* g_out;    // [n_seq][D]
  const float* tot;      // [n_seq]  g_out[seq] . y[seq]
  u16* dpre;             // [n_tok][QP] bf16
  float* dq_part;        // [gridDim.x][QP]
  u16* dctx;             // optional: bf16 [n_tok][KP] = dpre @ Wa (columns < D written)
  u16* dy_pad;           // optional, instead of dctx: bf16 seqpad rows (tok + seq + 1) = (dpre @ Wa + w (x) g_out) * [ctx != 0] * act_scale
  float act_scale;
  int64_t n_tok;         // n_seq * S < 2^31
  uint32_t S;            // >= 2
  uint32_t s_magic;      // floor(2^32 / S) + 1
  int dbg;               // DBG instantiation only (NR_POOL_DEBUG, tools/pool_phases.sh): 1 no ctx loads, 2 no dw phase, 4 no projection MFMAs,
                         // 8 no tanh / dpre / dq arithmetic, 16 no dctx product, 32 no global stores, 128 no dq accumulation
};

// tot[i] = a[i] . b[i] over d floats (d % 4 == 0; rows 16-byte aligned): one wave per row
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb, int64_t n,
                                                      int d, float* __restrict__ out) {
  const int l = lane_id();
  const int d4 = d >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave_id(); i < n; i += (int64_t)gridDim.x * 4) {
    float s = 0.0f;
    for (int c = l; c < d4; c += 64) {
      const f32x4 u = *(const f32x4*)(a + i * lda + c * 4), v = *(const f32x4*)(b + i * ldb + c * 4);
      s += u[0] * v[0] + u[1] * v[1] + u[2] * v[2] + u[3] * v[3];
    }
    s = sum_rows4(sum_row16(s));
    if (l == 0) out[i] = s;
  }
}

// the wave's 48 ctx rows as B-operand fragments: lane (li, g) holds features 32 ks + 8 g .. + 7 of token 16 m + li
template <int MT>
__device__ __forceinline__ void pool3_load_x(const u16* __restrict__ ctx, int64_t tok0, int64_t n_tok, u16x8 (&xf)[MT][KSTEPS]) {
  const int l = lane_id(), g = l >> 4, li = l & 15;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int64_t tok = tok0 + m * 16 + li;
    const bool live = tok < n_tok;
    const u16* row = ctx + (live ? tok : 0) * KP + g * 8;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) xf[m][ks] = live ? *(const u16x8*)(row + ks * 32) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
}

// ACT: the fused activation gradient (dy_pad) instead of dctx -- a compile-time form: its mask words and row scalars cost registers the plain
// form does not have to carry through the projection
template <bool ACT, int MT_ = 3, bool PF = true, bool DBG = false>
__global__ __launch_bounds__(Pool3GeomT<MT_>::THREADS) void pool3_bwd_kernel(Pool3Params p) {
  using Gm = Pool3GeomT<MT_>;
  const int dbg = DBG ? p.dbg : 0;        // the production instantiation folds every switch away
  constexpr int MT = Gm::MT;
  NR_SMEM_DECL(smem);
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  float* bq = (float*)(smem + Gm::W_BYTES);              // [QP] bias, [QP] query vector
  float* dqp = bq + 2 * QP + w * QP;                     // this wave's dq accumulator row
  const u16x4 Z4 = u16x4{0, 0, 0, 0};
  const bool with_dctx = (ACT || p.dctx != nullptr) && !(dbg & 16);       // the stand-alone AdditiveAttention backward stops at dpre / dq

  // ---- Wa rows -> LDS (once per workgroup): 16-byte pieces out of the tile-ordered operand -----------------------------------------------
  for (int i = tid; i < QP * (KP / 8); i += Gm::THREADS) {
    const int row = i / (KP / 8), s = i - row * (KP / 8);
    *(u16x8*)(smem + row * Gm::WROW + s * 16) = *(const u16x8*)(p.Wap + tile_off(row, s * 8, KP));
  }
  for (int i = tid; i < QP; i += Gm::THREADS) { bq[i] = p.bap[i]; bq[QP + i] = p.qvp[i]; }
  for (int i = tid; i < Gm::NWAVE * QP; i += Gm::THREADS) bq[2 * QP + i] = 0.0f;
  __syncthreads();

  const int64_t n_groups = (p.n_tok + Gm::ROWS - 1) / Gm::ROWS, gstride = (int64_t)gridDim.x * Gm::NWAVE;
  int64_t grp = (int64_t)blockIdx.x * Gm::NWAVE + w;     // neighbouring groups run on one CU at the same time: their partial lines merge in its L2
  u16x8 xf[MT][KSTEPS];
  const int64_t n_load = (dbg & 1) ? 0 : p.n_tok;        // (rows past the end are zero fragments)
  if (grp < n_groups) pool3_load_x(p.ctx, grp * Gm::ROWS, n_load, xf);

  while (grp < n_groups) {
    const int64_t tok0 = grp * Gm::ROWS;
    // ---- the lane's three rows (token tb + 16 m): ds = w (g . x - tot).  Sequence index and forward weight are dropped again after this
    // phase (the ACT epilogue fetches them a second time) instead of living through the projection ------------------------------------------
    const int64_t tb = tok0 + li;
    auto row_seq = [&](int m) -> uint32_t {
      const uint32_t tk = tb + m * 16 < p.n_tok ? (uint32_t)(tb + m * 16) : 0u;
      uint32_t q = mulhi_u32(tk, p.s_magic);   // floor(tk / S) or one more (tk < 2^31)
      return q - ((q * p.S > tk) ? 1u : 0u);
    };
    float ds[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const bool live = tb + m * 16 < p.n_tok;
      const uint32_t sq = row_seq(m);
      const float* go = p.g_out + (int64_t)sq * D + g * 8;
      float a = 0.0f;
#pragma unroll
      for (int ks = (dbg & 2) ? KSTEPS : 0; ks < KSTEPS; ++ks) {
        const u16x8 x = xf[m][ks];
        if (ks * 32 + 32 <= D || ks * 32 + g * 8 + 4 <= D) {                   // (columns >= D: the bias column of ctx, zeros)
          const f32x4 g0 = *(const f32x4*)(go + ks * 32);
          a += g0[0] * bf2f(x[0]) + g0[1] * bf2f(x[1]) + g0[2] * bf2f(x[2]) + g0[3] * bf2f(x[3]);
        }
        if (ks * 32 + 32 <= D || ks * 32 + g * 8 + 8 <= D) {
          const f32x4 g1 = *(const f32x4*)(go + ks * 32 + 4);
          a += g1[0] * bf2f(x[4]) + g1[1] * bf2f(x[5]) + g1[2] * bf2f(x[6]) + g1[3] * bf2f(x[7]);
        }
      }
      a = sum_rows4(a);
      ds[m] = live ? p.attn_w[tb + m * 16] * (a - p.tot[sq]) : 0.0f;
    }

    // ---- t = tanh(x Wa^T + ba);  dpre = ds qv (1 - t^2) (kept packed in registers + stored);  dq += ds t ---------------------------------------
    u16x4 dpk[Gm::NTQ + 1][MT];               // +1: the zero partner of the last (odd) n-tile in the dctx product
#pragma unroll
    for (int m = 0; m < MT; ++m) dpk[Gm::NTQ][m] = Z4;
#pragma unroll
    for (int nt = 0; nt < Gm::NTQ; ++nt) {
      const int wrow = nt * 16 + 4 * g;
      int bo = Gm::W_BYTES + 16 * g;          // (opaque for the same reason as `wo` below: else one hoisted address per n-tile and vector)
      NR_OPAQUE(bo);
      const f32x4 b4 = *(const f32x4*)(smem + bo + nt * 64), q4 = *(const f32x4*)(smem + bo + QP * 4 + nt * 64);
      f32x4 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = b4;
      // fragment (nt, ks) of Wa: one lane offset per SIX n-tiles + the instruction's 16-bit offset field.  (Left alone the compiler keeps one
      // loop-invariant address per n-tile, 13 registers, and spills them around the whole loop.)
      int wo = (li * Gm::WROW + g * 16) + (nt / 6) * (6 * 16 * Gm::WROW);
      NR_OPAQUE(wo);
      const unsigned char* wp = smem + wo + (nt % 6) * (16 * Gm::WROW);
      u16x8 a = *(const u16x8*)wp;
#pragma unroll
      for (int ks = (dbg & 4) ? KSTEPS : 0; ks < KSTEPS; ++ks) {
        const u16x8 an = ks + 1 < KSTEPS ? *(const u16x8*)(wp + (ks + 1) * 64) : a;
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32_bf16(a, xf[m][ks], acc[m]);
        a = an;
      }
      f32x4 dq4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4 dp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = (dbg & 8) ? acc[m][r] : fast_tanh(acc[m][r]);
          dp[r] = (dbg & 8) ? t : ds[m] * q4[r] * (1.0f - t * t);
          dq4[r] += (dbg & 8) ? 0.0f : ds[m] * t;
        }
        const u16x4 pk = pack4(dp);
        dpk[nt][m] = pk;
        if (tb + m * 16 < p.n_tok && !(dbg & 32)) *(u16x4*)(p.dpre + (tb + m * 16) * QP + wrow) = pk;
      }
      // dq: the tile's tokens live in the 16 lanes of a row -> DPP sum; the wave's own LDS row accumulates over all of its groups
#pragma unroll
      for (int r = (dbg & 128) ? 4 : 0; r < 4; ++r) {
        const float v = sum_row16(dq4[r]);
        if (li == 0) dqp[wrow + r] += v;
      }
      NR_SCHED_BARRIER();                     // keep the n-tiles apart: interleaving them stretches the live ranges past the register file
    }

    // ---- the next group's rows: the fragments are free, their latency hides behind the dctx product ----------------------------------------
    const int64_t nxt = grp + gstride;
    if (PF && !ACT && nxt < n_groups) pool3_load_x(p.ctx, nxt * Gm::ROWS, n_load, xf);

    // ---- dctx[tok][:] = dpre[tok][:] @ Wa: A = Wa^T rows of a feature tile by transposing reads of the SAME LDS rows (k-slot (g, j) of k-step ks
    // stands for query index 32 ks + 16 (j / 4) + 4 g + j % 4: the lane's packed dpre registers of tiles 2 ks, 2 ks + 1 are the B fragment) -----
    if (with_dctx) {
      uint32_t seqm[MT];
      float alpha[MT];
      if (ACT) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          seqm[m] = row_seq(m);
          alpha[m] = tb + m * 16 < p.n_tok ? p.attn_w[tb + m * 16] : 0.0f;
        }
      }
      const u16* wq = (const u16*)(smem + (4 * g + (li >> 2)) * Gm::WROW) + 4 * (li & 3);
      // one feature tile (columns 16 dt ..) of the product for the wave's rows
      auto product = [&](int dt, f32x4 (&acc)[MT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < Gm::KS2; ++ks) {
          const u16x4 lo = lds_tr16_b64(wq + (ks * 32) * XS + dt * 16);
          const u16x4 hi = 2 * ks + 1 < Gm::NTQ ? lds_tr16_b64(wq + (ks * 32 + 16) * XS + dt * 16) : lo;     // (the partner of the last n-tile is zero on the dpre side)
          const u16x8 a = cat8(lo, hi);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32_bf16(a, cat8(dpk[2 * ks][m], dpk[2 * ks + 1][m]), acc[m]);
        }
      };
      if (!ACT) {
#pragma nounroll
        for (int dt = 0; dt < Gm::NTD; ++dt) {
          f32x4 acc[MT];
          product(dt, acc);
          const int col = dt * 16 + 4 * g;
#pragma unroll
          for (int m = 0; m < MT; ++m)
            if (tb + m * 16 < p.n_tok && col < D && !(dbg & 32)) *(u16x4*)(p.dctx + (tb + m * 16) * KP + col) = pack4(acc[m]);
        }
      } else {
        // The relu / dropout mask of the conv stage is [activation != 0], and the activations are this wave's own B fragments -- in operand layout,
        // where the epilogue needs accumulator layout.  The matrix core moves them: E_h (16 x 32, E[d][k] = [k == 16 h + d]) times the
        // fragment of k-step dt / 2 is the tile x[tok][16 dt + d] exactly (1.0 x bf16 in fp32), lane for lane where acc holds the same element.
        u16x8 eh[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 8; ++j) eh[h][j] = (g == 2 * h + (li >> 3) && j == (li & 7)) ? BF16_ONE : (u16)0;
#pragma unroll
        for (int dt = 0; dt < Gm::NTD; ++dt) {
          f32x4 acc[MT];
          product(dt, acc);
          const int col = dt * 16 + 4 * g;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const f32x4 xv = mfma_16x16x32_bf16(eh[dt & 1], xf[m][dt >> 1], f32x4{0.f, 0.f, 0.f, 0.f});
            if (tb + m * 16 < p.n_tok && col < D && !(dbg & 32)) {
              // direct term from the g_out row and forward weight, straight into the seqpad row
              const f32x4 go = *(const f32x4*)(p.g_out + (int64_t)seqm[m] * D + col);
              f32x4 o;
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = xv[r] != 0.0f ? (acc[m][r] + alpha[m] * go[r]) * p.act_scale : 0.0f;
              *(u16x4*)(p.dy_pad + (tb + m * 16 + seqm[m] + 1) * KP + col) = pack4(o);
            }
          }
          NR_SCHED_BARRIER();
        }
      }
    }
    grp = nxt;
    if ((!PF || ACT) && nxt < n_groups) pool3_load_x(p.ctx, nxt * Gm::ROWS, n_load, xf);
  }
  __syncthreads();
  for (int n = tid; n < QP; n += Gm::THREADS) {
    float a = 0.0f;
#pragma unroll
    for (int ww = 0; ww < Gm::NWAVE; ++ww) a += bq[2 * QP + ww * QP + n];
    p.dq_part[(int64_t)blockIdx.x * QP + n] = a;
  }
}

}  // namespace nr
