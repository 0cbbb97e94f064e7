// Register-resident additive-attention pooling BACKWARD for titles (S = 20) and 50-token sequences, gfx950.  Same math and outputs as
// additive_fwd_kernel / additive_bwd_kernel (src/model/general/attention/additive.py:27-53 and its autograd) with the mapping of
// k_mhsa_fwd2.h: the LDS-tile kernels give a workgroup 2-4 titles and make it re-read the whole projection matrix from L2
// (133 KB per workgroup, 1.8 GB per launch at B = 512: they run at 5-10 % of the MFMA peak, bound by that traffic and by the
// stage -> project -> reduce -> softmax -> sum chain of dependent phases per tile).  Here
//
//   * one WAVE owns 4 titles = 80 tokens = 5 MFMA token tiles; its ctx rows are loaded ONCE, straight into MFMA B-operand fragments
//     (200 registers: lane (li, g) holds features 32 ks + 8 g .. +7 of token 16 m + li) and serve the projection, the score-weighted
//     sum (forward) and the g_out . x products (backward) from there -- no LDS token tile;
//   * the projection weights stream through LDS in double-buffered, fragment-major chunks copied global -> LDS directly
//     (global_load_lds_dwordx4) and shared by the 4 waves of the workgroup: one L2 read of Wa per 16 titles (8x less), every LDS
//     fragment read feeds 5 MFMAs;
//   * the transposed product (A = Wa rows, B = x^T) leaves a lane with 4 query-dim rows of one token: tanh, the dot with the query
//     vector and (backward) dpre = ds q (1 - t^2) are in-lane; per-title softmax / softmax-backward go through 640 B of wave-private LDS;
//   * backward: dpre never leaves registers before the fused input-gradient product dctx = dpre @ Wa.  Its B operand wants 8 consecutive
//     query indices per lane; the lane holds rows 4g..4g+3 of two neighbouring n-tiles instead.  As in k_mhsa_fwd2.h the contraction
//     index is simply PERMUTED on both sides: k-slot (g, j) of k-step ks stands for query index 16 (2 ks + j / 4) + 4 g + j % 4, and
//     pack_additive_t_kernel (k_misc.h) stores Wa^T in that order, so the fragment is the concatenation of two registers the lane already has.
//   The only workgroup barriers are the one per weight chunk.
#pragma once
#include "nr_common.h"
#include "k_additive_fwd.h"

namespace nr {

// TPW titles per wave x NWAVE waves per workgroup.  Instantiations: <20, 2, 8> -- two waves per SIMD on 3 token tiles = 48 rows for 40 tokens (17 %
// padding, but one wave's loads / tanh / reductions overlap the other's MFMAs; the one-wave-per-SIMD form <20, 4, 4> lost in rounds 2 and 3 and is
// gone) -- and <50, 1, 4>: one 50-token sequence (NAML abstracts, the NRMS click history) per wave on 4 token tiles, 4 sequences per workgroup.
// Both are the SHORT-launch kernels since round 4: large launches run the flat kernel of k_pool3.h.
template <int S_, int TPW_, int NWAVE_>
struct Pool2Geom {
  static constexpr int S = S_;
  static constexpr int TPW = TPW_;               // titles per wave
  static constexpr int NWAVE = NWAVE_;
  static constexpr int PER_WG = TPW * NWAVE;     // sequences per workgroup
  static constexpr int THREADS = NWAVE * 64;
  static constexpr int TOKW = S * TPW;           // 80 tokens per wave
  static constexpr int MT = (TOKW + 15) / 16;    // 5 (or 3) token tiles
  static constexpr int NTQ = QP / 16;            // 13 n-tiles of the query dim
  static constexpr int CH_NT = 5;                // n-tiles of Wa per chunk
  static constexpr int NCH = (NTQ + CH_NT - 1) / CH_NT;      // 3 chunks (5, 5, 3 n-tiles)
  static constexpr int CH_BYTES = CH_NT * KSTEPS * 1024;     // 51,200 B
  static constexpr int KS2 = QKP / 32;           // 7 k-steps of the dctx product (query dim 208 -> 224)
  static constexpr int NTD = (D + 15) / 16;      // 19 feature tiles of dctx
  static constexpr int CH_DT = 7;                // feature tiles of Wa^T per chunk: 7 * 7 KiB = 50,176 B <= CH_BYTES
  static constexpr int NCH2 = (NTD + CH_DT - 1) / CH_DT;     // 3 chunks (7, 7, 5)
  static constexpr int WV_FLOATS = 2 * TOKW;     // per-wave scratch: scores / dw, weights / ds
  static constexpr int GROW = KP;                // floats per staged g_out row (zero padded beyond D)
  static constexpr int FWD_SMEM = 2 * CH_BYTES + NWAVE * WV_FLOATS * 4;
  static constexpr int BWD_SMEM = FWD_SMEM + NWAVE * TPW * GROW * 4 + NWAVE * QP * 4;
  static_assert(CH_DT * KS2 * 1024 <= CH_BYTES && TOKW <= 128, "geometry");
};

// the wave's 80 ctx rows as B-operand fragments
template <int MT, int TOKW>
__device__ __forceinline__ void pool2_load_x(const u16* __restrict__ ctx, int64_t tok0, int64_t tok_total, u16x8 (&xf)[MT][KSTEPS]) {
  const int l = lane_id(), g = l >> 4, li = l & 15;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int64_t tok = tok0 + m * 16 + li;
    const bool live = tok < tok_total && m * 16 + li < TOKW;            // rows beyond the wave's tokens (48 rows for 40 tokens) are zero
    const u16* row = ctx + (live ? tok : 0) * KP + g * 8;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) xf[m][ks] = live ? *(const u16x8*)(row + ks * 32) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
}

// (The register-resident pooling FORWARD of rounds 2-3 -- pool2_fwd_kernel, NR_POOL2_FWD=1 -- measured 286 / 271 / 304 us in its three forms against
// 252-265 us for the LDS-tile additive_fwd_kernel, two rounds running, and is gone: profiles/r02_ab_switches.txt)

// ---------------------------------------------------------------------------------------------------------------------------------------------
// DBG = true: a second instantiation with phase switches for timing decompositions (NR_POOL_DEBUG -> dbgv; tools/prof_kernel.py):
// 1 no ctx loads, 2 no dw / softmax-backward phase, 4 no projection MFMAs, 8 no tanh / dpre / dq arithmetic, 16 no dctx product,
// 32 no global stores.  In the production instantiation `dbg` is the constant 0 and every switch folds away.
template <typename Gm, bool DBG = false>
__global__ __launch_bounds__(Gm::THREADS) void pool2_bwd_kernel(AdditiveBwdParams p, int dbgv) {
  constexpr int S = Gm::S, MT = Gm::MT;
  const int dbg = DBG ? dbgv : 0;
  NR_SMEM_DECL(smem);
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  const int64_t seq0 = ((int64_t)blockIdx.x * Gm::NWAVE + w) * Gm::TPW;
  const int64_t tok0 = seq0 * S, tok_total = p.n_seq * S;
  float* sc = (float*)(smem + 2 * Gm::CH_BYTES) + w * Gm::WV_FLOATS;       // [80] dw, later ds
  float* wl = sc + Gm::TOKW;                                               // [80] forward attention weights
  float* gl = (float*)(smem + Gm::FWD_SMEM) + w * Gm::TPW * Gm::GROW;      // [4][320] g_out rows of this wave's titles, zero padded
  float* dqp = (float*)(smem + Gm::FWD_SMEM + Gm::NWAVE * Gm::TPW * Gm::GROW * 4);     // [4][QP] per-wave dq partials
  const u16x4 Z4 = u16x4{0, 0, 0, 0};
  const bool with_dctx = p.dctx != nullptr || p.dy_pad != nullptr;        // the stand-alone AdditiveAttention backward stops at dpre / dq

  // chunks 0..NCH-1: Wa (n-tiles 5c..), chunks NCH..NCH+NCH2-1: pair-permuted Wa^T (feature tiles 7c'..); both operands are in tile order
  auto chunk_fetch = [&](int c, int buf) {
    unsigned char* dst = smem + buf * Gm::CH_BYTES;
    if (c < Gm::NCH) {
      const int nt0 = c * Gm::CH_NT, ntc = (Gm::NTQ - nt0) < Gm::CH_NT ? (Gm::NTQ - nt0) : Gm::CH_NT;
      const u16* src = p.Wap + (size_t)nt0 * KSTEPS * 512 + l * 8;
      for (int blk = w; blk < ntc * KSTEPS; blk += Gm::NWAVE) NR_GLDS16(src + (size_t)blk * 512, dst + blk * 1024);
    } else {
      const int dt0 = (c - Gm::NCH) * Gm::CH_DT, dtc = (Gm::NTD - dt0) < Gm::CH_DT ? (Gm::NTD - dt0) : Gm::CH_DT;
      const u16* src = p.WaT + (size_t)dt0 * Gm::KS2 * 512 + l * 8;
      for (int blk = w; blk < dtc * Gm::KS2; blk += Gm::NWAVE) NR_GLDS16(src + (size_t)blk * 512, dst + blk * 1024);
    }
  };
  chunk_fetch(0, 0);
  u16x8 xf[MT][KSTEPS];
  pool2_load_x<MT, Gm::TOKW>(p.ctx, tok0, (dbg & 1) ? 0 : tok_total, xf);
  // fused activation gradient (dy_pad): the relu / dropout mask of the conv stage is [activation != 0].  The activations are this wave's own
  // fragments: one bit per element (bit 8 ks + j of a token row = feature 32 ks + 8 g + j), 80 bits per lane and token tile, kept until the
  // epilogue -- which used to fetch the activation values again with an 8-byte gather per (token, 4 features): +0.4 ms on NAML's abstracts
  uint32_t mk[MT][3];
  if (p.dy_pad != nullptr) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      mk[m][0] = mk[m][1] = mk[m][2] = 0u;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) mk[m][(ks * 8 + j) >> 5] |= (xf[m][ks][j] & 0x7FFF) ? 1u << ((ks * 8 + j) & 31) : 0u;
    }
  }
  // g_out rows and forward weights of this wave's titles -> wave-private LDS
  for (int i = l; i < Gm::TPW * (Gm::GROW / 4); i += 64) {
    const int sq = i / (Gm::GROW / 4), c = i - sq * (Gm::GROW / 4);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < D4 && seq0 + sq < p.n_seq) v = *(const f32x4*)(p.g_out + (seq0 + sq) * D + c * 4);
    *(f32x4*)(gl + sq * Gm::GROW + c * 4) = v;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int t = l + 64 * k;
    if (t < Gm::TOKW) wl[t] = tok0 + t < tok_total ? p.attn_w[tok0 + t] : 0.0f;
  }
  wave_barrier();

  // ---- dw[tok] = g_out[title] . x[tok] --------------------------------------------------------------------------------------------------
#pragma unroll
  for (int m = (dbg & 2) ? MT : 0; m < MT; ++m) {
    const int trow = m * 16 + li < Gm::TOKW ? m * 16 + li : Gm::TOKW - 1;          // rows past the wave's tokens (zero fragments) stay in range
    const float* go = gl + (trow / S) * Gm::GROW + g * 8;
    float a = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const f32x4 g0 = *(const f32x4*)(go + ks * 32), g1 = *(const f32x4*)(go + ks * 32 + 4);
      const u16x8 x = xf[m][ks];
      a += g0[0] * bf2f(x[0]) + g0[1] * bf2f(x[1]) + g0[2] * bf2f(x[2]) + g0[3] * bf2f(x[3]);
      a += g1[0] * bf2f(x[4]) + g1[1] * bf2f(x[5]) + g1[2] * bf2f(x[6]) + g1[3] * bf2f(x[7]);
    }
    a = sum_rows4(a);
    if (g == 0 && m * 16 + li < Gm::TOKW) sc[m * 16 + li] = a;
  }
  wave_barrier();
  // ---- softmax backward: ds = w (dw - sum_s w dw) ---------------------------------------------------------------------------------------------
  float dsn[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int t = l + 64 * k;
    dsn[k] = 0.0f;
    if (t < Gm::TOKW) {
      const int base = (t / S) * S;
      float tot = 0.0f;
#pragma unroll
      for (int j = 0; j < S; ++j) tot += wl[base + j] * sc[base + j];
      dsn[k] = wl[t] * (sc[t] - tot);
    }
  }
  wave_barrier();                           // every lane has read dw before it is overwritten by ds
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int t = l + 64 * k;
    if (t < Gm::TOKW) sc[t] = dsn[k];
  }
  wave_barrier();
  float ds[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) ds[m] = m * 16 + li < Gm::TOKW ? sc[m * 16 + li] : 0.0f;
  __syncthreads();                          // Wa chunk 0 visible

  // ---- recompute t = tanh(x Wa^T + ba);  dpre = ds qv (1 - t^2) (kept packed in registers + stored);  dq += ds t ---------------------------
  u16x4 dpk[Gm::NTQ + 1][MT];               // +1: the zero partner of the last (odd) n-tile in the dctx product
#pragma unroll
  for (int m = 0; m < MT; ++m) dpk[Gm::NTQ][m] = Z4;
#pragma unroll
  for (int c = 0; c < Gm::NCH; ++c) {
    if (c + 1 < Gm::NCH || with_dctx) chunk_fetch(c + 1, (c + 1) & 1);        // chunk NCH is the first Wa^T chunk
    const int nt0 = c * Gm::CH_NT;
    const u16* Wc = (const u16*)(smem + (c & 1) * Gm::CH_BYTES);
#pragma unroll
    for (int nt = 0; nt < Gm::CH_NT; ++nt) {
      if (nt0 + nt < Gm::NTQ) {
        const int wrow = (nt0 + nt) * 16 + 4 * g;
        const f32x4 b4 = *(const f32x4*)(p.bap + wrow), q4 = *(const f32x4*)(p.qvp + wrow);
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = b4;
        const u16* wp = Wc + (nt * KSTEPS) * 512 + l * 8;
        u16x8 a = *(const u16x8*)wp;
#pragma unroll
        for (int ks = (dbg & 4) ? KSTEPS : 0; ks < KSTEPS; ++ks) {
          const u16x8 an = ks + 1 < KSTEPS ? *(const u16x8*)(wp + (ks + 1) * 512) : a;
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
          dpk[nt0 + nt][m] = pk;
          const int64_t tok = tok0 + m * 16 + li;
          if (tok < tok_total && m * 16 + li < Gm::TOKW && !(dbg & 32)) *(u16x4*)(p.dpre + tok * QP + wrow) = pk;
        }
        // dq partial of this wave: the tile's tokens live in the 16 lanes of a row -> DPP sum; the 4 waves are combined through LDS below
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = sum_row16(dq4[r]);
          if (li == 0) dqp[w * QP + wrow + r] = v;
        }
        NR_SCHED_BARRIER();                 // keep the n-tiles apart: interleaving them stretches the live ranges past the register file
      }
    }
    __syncthreads();
  }

  // ---- dctx[tok][:] = dpre[tok][:] @ Wa  (transposed product: A = Wa^T rows of a feature tile, B = the packed dpre registers) -----------
  if (!with_dctx) __syncthreads();          // orders the dqp stores before the reduction below
  for (int c2 = 0; c2 < ((with_dctx && !(dbg & 16)) ? Gm::NCH2 : 0); ++c2) {
    const int c = Gm::NCH + c2;
    if (c2 + 1 < Gm::NCH2) chunk_fetch(c + 1, (c + 1) & 1);
    const int dt0 = c2 * Gm::CH_DT;
    const u16* Wc = (const u16*)(smem + (c & 1) * Gm::CH_BYTES);
    for (int dt = 0; dt < Gm::CH_DT; ++dt) {
      if (dt0 + dt < Gm::NTD) {
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const u16* wp = Wc + (dt * Gm::KS2) * 512 + l * 8;
#pragma unroll
        for (int ks = 0; ks < Gm::KS2; ++ks) {
          const u16x8 a = *(const u16x8*)(wp + ks * 512);      // (an explicit one-ahead prefetch of this fragment measures the same: A/B r02v)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32_bf16(a, cat8(dpk[2 * ks][m], dpk[2 * ks + 1][m]), acc[m]);
        }
        const int col = (dt0 + dt) * 16 + 4 * g;
        // mask bits of (token li of tile m, features col .. col + 3): k-step (dt0 + dt) / 2, k-slots 4 (g & 1) .. + 3 of lane group
        // 2 ((dt0 + dt) & 1) + (g >> 1) of the same token -- one cross-lane fetch of the 32-bit word that holds them
        const int ksd = (dt0 + dt) >> 1, src_lane = (2 * ((dt0 + dt) & 1) + (g >> 1)) * 16 + li, mshift = (ksd & 3) * 8 + 4 * (g & 1);
        uint32_t mbits[MT];
        if (p.dy_pad != nullptr) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const uint32_t word = (ksd >> 2) == 0 ? mk[m][0] : ((ksd >> 2) == 1 ? mk[m][1] : mk[m][2]);
            mbits[m] = shfl_u32(word, src_lane) >> mshift;          // (every lane of the wave takes part)
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int tl = m * 16 + li;
          const int64_t tok = tok0 + tl;
          if (tok < tok_total && tl < Gm::TOKW && col < D && !(dbg & 32)) {
            if (p.dy_pad == nullptr) {
              *(u16x4*)(p.dctx + tok * KP + col) = pack4(acc[m]);
            } else {
              // fused activation gradient (conv_act_bwd_kernel): direct term from the staged g_out row and forward weight, relu / dropout
              // mask from the activation itself (ctx == 0 <=> dropped or relu-clipped), straight into the seqpad row
              const int sq = tl / S;
              const f32x4 go = *(const f32x4*)(gl + sq * Gm::GROW + col);
              const float wt = wl[tl];
              f32x4 o;
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = ((mbits[m] >> r) & 1u) ? (acc[m][r] + wt * go[r]) * p.act_scale : 0.0f;
              *(u16x4*)(p.dy_pad + (tok + seq0 + sq + 1) * KP + col) = pack4(o);
            }
          }
        }
      }
    }
    __syncthreads();                        // also orders the dqp stores above before the reduction below
  }
  for (int n = tid; n < QP; n += Gm::THREADS) {
    float a = 0.0f;
#pragma unroll
    for (int ww = 0; ww < Gm::NWAVE; ++ww) a += dqp[ww * QP + n];
    p.dq_part[(int64_t)blockIdx.x * QP + n] = a;
  }
}

}  // namespace nr
