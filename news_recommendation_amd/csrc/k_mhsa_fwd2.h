// Register-resident multi-head self-attention forward for titles (S = 20), gfx950.  Same math and outputs as
// k_mhsa_fwd.h (src/model/general/attention/multihead_self.py:15-23,46-75 + the gather / dropout front of
// src/model/NRMS/news_encoder.py:38-45) with a different mapping, chosen after the PMC profile of the LDS-tile kernel
// (12 % MFMA busy, 47 % of wave cycles parked on barriers / waitcnt, 45 % of LDS cycles bank-conflicted):
//
//   * one WAVE owns 4 titles = 80 tokens = 5 MFMA token tiles and never exchanges data with another wave;
//   * its token matrix lives in REGISTERS for the whole kernel as MFMA operand fragments (200 VGPRs: lane (li, g) holds
//     features 32 ks + 8 g .. +7 of token 16 m + li), gathered straight from the fp32 table -- no LDS tile, no conversions pass;
//   * the weights stream through LDS in double-buffered chunks of 5 n-tiles (80 rows x 320), copied global -> LDS directly
//     (global_load_lds_dwordx4) in fragment-major order and shared by the 4 waves of the workgroup (one L2 read per 16
//     titles); one conflict-free LDS fragment read feeds 5 MFMAs (one per token tile);
//   * Q, K (transposed product: lane = token, 4 consecutive features) and V (plain product: lane = feature, 4 consecutive
//     tokens) of a 4-head group stay in registers in exactly the fragment layouts the attention MFMAs want:
//       S^T tile  = K-tile-pair (A) x Q-tile-pair (B): k-slot (g, r) of the first tile and (g, 4 + r) of the second carry the
//                   same feature on both sides, so no transposition or LDS round trip is needed; features of neighbouring
//                   heads that share the 16-row tiles are zeroed on the K side; one now-free k-slot carries the key mask
//                   (q = 1, k = -29952 on tokens of other titles), so foreign keys leave exp2 as exact zeros;
//       ctx^T tile = V-tile-pair (A, k = key token) x P^T (B): P is packed in place, V is already key-major.
//   There are no workgroup barriers except the one per weight chunk.
#pragma once
#include "nr_common.h"
#include "k_mhsa_fwd.h"
#include <type_traits>

namespace nr {

struct Mhsa2Geom {
  static constexpr int S = 20;
  static constexpr int TPW = 4;                  // titles per wave
  static constexpr int NWAVE = 4;
  static constexpr int TOKW = S * TPW;           // 80 tokens per wave
  static constexpr int MT = TOKW / 16;           // 5 token tiles
  static constexpr int CH_ROWS = HG * DK;        // 80 weight rows per chunk (5 n-tiles)
  static constexpr int CH_BYTES = CH_ROWS * KP * 2;   // 51,200 B: 50 fragment blocks of 1 KiB
  static constexpr int B_BYTES = 3 * NP * 4;     // 3,840 B: the packed bias vector, read once per n-tile by every wave
  static constexpr int SMEM = 2 * CH_BYTES + B_BYTES;
  static_assert(TOKW % 16 == 0 && S % 4 == 0, "geometry");
};

// DBG = true: a second instantiation with the NR_MHSA_DEBUG phase switches (MhsaParams::debug) for timing decompositions; in the
// production instantiation (DBG = false) `dbg` is the constant 0 and every switch folds away.
template <bool DBG>
__global__ __launch_bounds__(256) NR_ONE_WAVE_PER_SIMD void mhsa_fwd2_kernel(MhsaParams p) {
  using Gm = Mhsa2Geom;
  const int dbg = DBG ? p.debug : 0;
  p.dc = drop_resolve(p.dc);
  constexpr int S = Gm::S, MT = Gm::MT;
  NR_SMEM_DECL(smem);
  auto wl = [&](int buf) -> u16* { return (u16*)(smem + buf * Gm::CH_BYTES); };
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  const int64_t seq0 = ((int64_t)blockIdx.x * Gm::NWAVE + w) * Gm::TPW;      // first title of this wave
  const int64_t tok0 = seq0 * S, tok_total = p.n_seq * S;
  const u16x4 Z4 = u16x4{0, 0, 0, 0};

  // ---- weight chunk c = 3 * head_group + {Q, K, V}: global -> LDS directly (global_load_lds_dwordx4, no staging registers) -----
  // LDS image is FRAGMENT-MAJOR: block (nt, ks) = 1 KiB holding, lane-linear, the 16 B that lane (li, g) needs as its MFMA
  // operand fragment W[row 16 nt + li][32 ks + 8 g .. +7].  Reads are one conflict-free ds_read_b128 at block + 16 lane.
  auto chunk_fetch = [&](int c, int buf) {
    const u16* src = p.Wp + ((size_t)(c % 3) * NP + (size_t)(c / 3) * Gm::CH_ROWS) * KP + l * 8;       // Wp is in tile order already
    unsigned char* dst = smem + buf * Gm::CH_BYTES;
    for (int blk = w; blk < 5 * KSTEPS; blk += Gm::NWAVE) {       // 50 blocks per chunk, 12-13 per wave
      const int nt = blk / KSTEPS, ks = blk - nt * KSTEPS;
      NR_GLDS16(src + (size_t)nt * 16 * KP + ks * 512, dst + blk * 1024);
    }
  };
  chunk_fetch(0, 0);
  // biases -> LDS (visible after the barrier that follows the token gather).  As global loads they sat in front of every n-tile's
  // MFMA chain (load, s_waitcnt vmcnt(0), then the accumulator init): 60 exposed L2 round trips per wave at one wave per SIMD.
  float* bl = (float*)(smem + 2 * Gm::CH_BYTES);
  for (int i = tid; i < 3 * NP / 4; i += 256) *(f32x4*)(bl + i * 4) = *(const f32x4*)(p.bp + i * 4);

  // ---- gather the wave's 80 tokens into operand fragments -----------------------------------------------------------------
  u16x8 xf[MT][KSTEPS];
  f32x4 lo[2][KSTEPS], hi[2][KSTEPS];        // two token tiles of raw fp32 rows in flight
  auto issue = [&](int m, int slot) {
    const int64_t tok = tok0 + m * 16 + li;
    const bool live = tok < tok_total;
    const float* row = nullptr;
    if (live) {
      if (p.ids != nullptr) {
        int64_t id = p.ids[tok];
        id = id < 0 ? 0 : (id >= p.num_rows ? p.num_rows - 1 : id);
        row = p.table + (size_t)id * D;
      } else {
        row = p.x_dense + (size_t)tok * D;
      }
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = ks * 32 + g * 8;
      lo[slot][ks] = (live && c < D && !(dbg & 1)) ? *(const f32x4*)(row + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      hi[slot][ks] = (live && c + 4 < D && !(dbg & 1)) ? *(const f32x4*)(row + c + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  issue(0, 0);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (m + 1 < MT) issue(m + 1, (m + 1) & 1);
    const int64_t tok = tok0 + m * 16 + li;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = ks * 32 + g * 8;
      f32x4 a = lo[m & 1][ks], b = hi[m & 1][ks];
      if (p.dc.enabled) {
        if (c < D) a = a * drop_mul4(p.dc, 1u, (uint64_t)tok * D4 + (c >> 2));
        if (c + 4 < D) b = b * drop_mul4(p.dc, 1u, (uint64_t)tok * D4 + (c >> 2) + 1);
      }
      xf[m][ks] = cat8(pack4(a), pack4(b));
      if (p.x_save != nullptr && tok < tok_total && !(dbg & 8)) {   // the weight-gradient GEMM operand, straight from the fragment registers
        u16x8 o = xf[m][ks];
        if (c <= D && D < c + 8) o[D - c] = BF16_ONE;        // column D = 1.0: the GEMM then also yields the bias gradient
        *(u16x8*)(p.x_save + tok * KP + c) = o;
      }
    }
  }
  __syncthreads();

  const float c2 = LOG2E / sqrtf((float)DK), clamp2 = EXP_CLAMP * LOG2E;
  constexpr int NCHUNK = 3 * NGROUPS;
  int klen[Gm::TPW];                              // key length of each of the wave's titles (MhsaParams::key_len), S when absent
#pragma unroll
  for (int sq = 0; sq < Gm::TPW; ++sq)
    klen[sq] = (p.key_len != nullptr && seq0 + sq < p.n_seq) ? uniform(clamp_len(p.key_len[seq0 + sq], S)) : S;

  u16x4 qr[5][MT], kr[5][MT], vr[5][MT];       // [n-tile of the head group][token tile]

  // one weight chunk: fetch the next one, project this one (WHICH = 0 Q, 1 K, 2 V is a compile-time constant so that the three
  // register tiles are addressed statically), attention after V, then commit the prefetched chunk
  auto project = [&](auto WHICH, int hg, int NT, const u16* Wc) {
    constexpr int which = decltype(WHICH)::value;
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
      if (nt < NT) {
        const int wrow = which * NP + hg * Gm::CH_ROWS + nt * 16;      // row in the packed matrix / bias vector
        f32x4 acc[MT];
        if (which < 2) {
          const f32x4 b4 = *(const f32x4*)(bl + wrow + 4 * g);          // lane owns features 4g..4g+3
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = b4;
        } else {
          const float b1 = bl[wrow + li];                                // lane owns feature li
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = f32x4{b1, b1, b1, b1};
        }
        const u16* wp = Wc + (nt * KSTEPS) * 512 + l * 8;        // fragment block (nt, ks) at +512 elements per ks
        u16x8 a = *(const u16x8*)wp;
#pragma unroll
        for (int ks = (dbg & 2) ? KSTEPS : 0; ks < KSTEPS; ++ks) {
          const u16x8 an = ks + 1 < KSTEPS ? *(const u16x8*)(wp + (ks + 1) * 512) : a;     // next fragment in flight during the MFMAs
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m] = which < 2 ? mfma_16x16x32_bf16(a, xf[m][ks], acc[m]) : mfma_16x16x32_bf16(xf[m][ks], a, acc[m]);
          a = an;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const u16x4 v = pack4(acc[m]);
          if (which == 0) qr[nt][m] = v; else if (which == 1) kr[nt][m] = v; else vr[nt][m] = v;
          if (p.q_save != nullptr && !(dbg & 8)) {
            if (which < 2) {               // row-major Q / K: lane = token 16m + li, features col .. col+3
              const int64_t tok = tok0 + m * 16 + li;
              const int col = hg * Gm::CH_ROWS + nt * 16 + 4 * g;
              if (tok < tok_total && col < D) *(u16x4*)((which == 0 ? p.q_save : p.k_save) + tok * KP + col) = v;
            } else {                       // dv-major V blocks: lane = feature, tokens 16m + 4g .. +3 (inside one title: 20 % 4 == 0)
              const int t0 = m * 16 + 4 * g;
              const int sq = t0 / S, tis = t0 - sq * S;
              const int col = hg * Gm::CH_ROWS + nt * 16 + li;
              const int hd = col / DK, dv = col - hd * DK;
              if (seq0 + sq < p.n_seq && col < D) *(u16x4*)(p.vt_save + (((seq0 + sq) * H + hd) * DK + dv) * S + tis) = v;
            }
          }
        }
      }
    }
  };
  auto next_chunk = [&](int) {
    __syncthreads();                        // drains the in-flight global->LDS copies: next chunk visible; and everybody is done
  };                                        // reading the buffer that the chunk after that will overwrite
  using std::integral_constant;

  for (int hg = 0; hg < NGROUPS; ++hg) {
    const int nh = (H - hg * HG) < HG ? (H - hg * HG) : HG;
    const int NT = (nh * DK + 15) / 16;                     // 5, or 4 in the last (3-head) group
    int c = 3 * hg;
    chunk_fetch(c + 1, (c + 1) & 1);                        // the next chunk's global->LDS copies fly during this chunk's MFMAs
    project(integral_constant<int, 0>{}, hg, NT, wl(c & 1));
    next_chunk(c);
    ++c;
    chunk_fetch(c + 1, (c + 1) & 1);
    project(integral_constant<int, 1>{}, hg, NT, wl(c & 1));
    next_chunk(c);
    ++c;
    if (c + 1 < NCHUNK) chunk_fetch(c + 1, (c + 1) & 1);
    project(integral_constant<int, 2>{}, hg, NT, wl(c & 1));

    // ---- attention of the head group once Q, K, V are complete -------------------------------------------------------------
    if (!(dbg & 4)) {
#pragma unroll
      for (int hd = 0; hd < HG; ++hd) {
        if (hd < nh) {
          const int c0 = hd * DK;                     // first column of the head inside the group
          const int ta = c0 / 16, tb = ta + 1;        // the two n-tiles the head's 20 features live in
          const int ra = c0 - ta * 16;                // tile a: rows ra..15 belong to the head
          const int rb = c0 + DK - tb * 16;           // tile b: rows 0..rb-1 belong to the head
          const bool in_a = 4 * g >= ra, in_b = 4 * g < rb;
          // free k-slot for the key mask: a slot whose feature belongs to a neighbouring head
          const bool mslot_a = (hd > 0) && g == 0;    // (tile a, row 0) is foreign for heads 1..3
          const bool mslot_b = (hd == 0) && g == 1;   // (tile b, row 4) is foreign for head 0
#pragma unroll
          for (int sq = 0; sq < Gm::TPW; ++sq) {
            const int i0 = (sq * S) / 16;             // the title's tokens live in token tiles i0, i0 + 1
            const int kend = sq * S + klen[sq];       // keys of this title beyond its key length are masked like foreign tokens
            u16x8 ka[2], qa[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int i = i0 + t;
              const int tokl = i * 16 + li;           // wave-local token of this lane (A row / B column)
              const bool mine = tokl >= sq * S && tokl < kend;
              u16x4 klo = in_a ? kr[ta][i] : Z4, khi = in_b ? kr[tb][i] : Z4;
              u16x4 qlo = qr[ta][i], qhi = qr[tb][i];
              if (mslot_a) { klo[0] = mine ? (u16)0 : BF16_NEG_BIG; qlo[0] = BF16_ONE; }
              if (mslot_b) { khi[0] = mine ? (u16)0 : BF16_NEG_BIG; qhi[0] = BF16_ONE; }
              ka[t] = cat8(klo, khi);
              qa[t] = cat8(qlo, qhi);
            }
            u16x4 pt[2][2];                            // P^T [key tile][query tile]
#pragma unroll
            for (int qj = 0; qj < 2; ++qj) {
              f32x4 e[2];
              float sum = 0.0f;
#pragma unroll
              for (int ki = 0; ki < 2; ++ki) {
                e[ki] = mfma_16x16x32_bf16(ka[ki], qa[qj], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  e[ki][r] = fast_exp2(fminf(e[ki][r] * c2, clamp2));
                  sum += e[ki][r];
                }
              }
              sum = sum_rows4(sum);
              const float rden = fast_rcp(sum + 1e-8f);           // exp / (sum + 1e-8): multihead_self.py:16-20 verbatim
              pt[0][qj] = pack4(e[0] * rden);
              pt[1][qj] = pack4(e[1] * rden);
            }
            // ctx^T = V^T P^T for the two feature tiles of the head
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int tt = t == 0 ? ta : tb;
              const u16x8 va = cat8(vr[tt][i0], vr[tt][i0 + 1]);
              const bool rows_mine = t == 0 ? in_a : in_b;
#pragma unroll
              for (int qj = 0; qj < 2; ++qj) {
                f32x4 acc = mfma_16x16x32_bf16(va, cat8(pt[0][qj], pt[1][qj]), f32x4{0.f, 0.f, 0.f, 0.f});
                const int tokl = (i0 + qj) * 16 + li;
                const int64_t tok = tok0 + tokl;
                const int col = hg * Gm::CH_ROWS + tt * 16 + 4 * g;
                if (rows_mine && tokl >= sq * S && tokl < (sq + 1) * S && tok < tok_total && !(dbg & 16)) {
                  if (p.dc.enabled) acc = acc * drop_mul4(p.dc, 2u, (uint64_t)tok * D4 + (col >> 2));
                  *(u16x4*)(p.ctx + tok * KP + col) = pack4(acc);
                }
              }
            }
          }
        }
      }
    }

    next_chunk(c);
  }

  // K padding of the ctx rows of this wave: col D = 1.0 (bias-gradient column), cols D+1..KP-1 = 0
  constexpr int PADQ = (KP - D) / 4;
  for (int i = l; i < Gm::TOKW * PADQ; i += 64) {
    const int r = i / PADQ, cq = i - r * PADQ;
    const int64_t tok = tok0 + r;
    if (tok < tok_total) *(u16x4*)(p.ctx + tok * KP + D + cq * 4) = u16x4{(u16)(cq == 0 ? 0x3F80 : 0), 0, 0, 0};
  }
}

}  // namespace nr
