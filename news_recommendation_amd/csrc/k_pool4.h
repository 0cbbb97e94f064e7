// Additive-attention pooling FORWARD over whole sequences per wave, persistent, gfx950 (src/model/general/attention/additive.py:27-53):
//   out[seq][:] = sum_s softmax_s( tanh(x[seq][s][:] Wa^T + ba) . qv ) x[seq][s][:]
// The LDS-tile kernel (k_additive_fwd.h) gives a workgroup one 50-token sequence (or two titles): every workgroup re-reads the 128 KB
// projection matrix from L2 and walks a stage -> project -> softmax -> sum chain with three workgroup barriers -- 0.55 + 0.27 ms per NAML step
// for the abstracts and titles, ~0.17 of the HBM roofline.  This is the forward counterpart of k_pool3.h:
//   * PERSISTENT, one workgroup of 8 waves per CU; Wa (200 rows x 640 B, swizzled 16-byte slots) is loaded into LDS once and never moves;
//   * a wave owns WHOLE sequences: a group of 64 token rows = floor(64 / S) sequences (3 titles of 20 = 60 rows; one abstract of 50), so the
//     softmax needs no other wave and the kernel has NO barrier inside its loop;
//   * the group's rows are loaded as row pieces (four consecutive lanes on one row: 64 contiguous bytes), turned into MFMA B fragments through 4 KB
//     of wave-private LDS and stay in registers (160); the projection x Wa^T + ba runs as the transposed product (A = Wa fragments from LDS), so a
//     lane holds 4 query rows of one token: tanh * qv is reduced in-lane and across the four lane groups (two lane swaps);
//   * the weighted sum runs on the matrix core too: per 32-feature k-step the fragments go back to the scratch as rows and come out again through
//     the TRANSPOSING read (ds_read_b64_tr_b16) as B fragments whose contraction index is the TOKEN; A = the softmax weights of the group's
//     sequences, one (hi, lo) bf16 pair of rows per sequence (w = hi + lo to 2^-17; products with the bf16 rows exact, fp32 accumulation).
// Any S in [16, 64] (a group must hold at least one sequence and at most 4); query_vector_dim <= 200 (the rows of Wa kept in LDS), like k_pool3.h.
#pragma once
#include "nr_common.h"
#include "k_additive_fwd.h"
#include "k_pool3.h"

namespace nr {

struct Pool4Geom {
  static constexpr int NWAVE = 8, THREADS = NWAVE * 64;
  static constexpr int MT = 4, ROWS = MT * 16;   // token rows per wave and iteration
  static constexpr int NTQ = QP / 16;            // 13 n-tiles of the query dim
  static constexpr int WROWS = Pool3Geom::WROWS, WROW = Pool3Geom::WROW, W_BYTES = Pool3Geom::W_BYTES;
  static constexpr int BQ_BYTES = 2 * QP * 4;    // bias and query vector
  static constexpr int SC_WAVE = ROWS * 64;      // 4,096 B of layout-change scratch per wave: [64 rows][4 slots of 16 B], slots swizzled (sc_swz);
                                                 // between the two uses of the rows it also holds the group's 64 softmax weights (f32)
  static constexpr int SMEM = W_BYTES + BQ_BYTES + NWAVE * SC_WAVE;      // 163,072
  static constexpr int MAXSLOT = 4;              // sequences per group: S >= 16
  static_assert(SMEM <= 163840, "LDS");
};

struct Pool4Params {
  const u16* ctx;        // [n_seq * S][KP]
  const u16* Wap;        // [QP][KP] tile order
  const float* bap;      // [QP]
  const float* qvp;      // [QP]
  float* out;            // [n_seq][out_stride] (first D columns) or null
  int64_t out_stride;
  u16* out_b;            // optional bf16 copy in the ctx layout (row i at out_b + i * out_b_stride: cols 0..D-1, col D = 1.0, rest 0)
  int64_t out_b_stride;
  float* attn_w;         // [n_seq][S] or null
  int64_t n_seq;
  int S, valid;          // tokens s >= valid of every sequence get weight exactly 0
  int dbg;               // DBG instantiation only (NR_POOL_DEBUG, tools/prof_kernel.py pool_fwd_flat*): 1 no row loads, 2 no rows -> fragments pass,
                         // 4 no projection MFMAs, 8 no tanh, 16 no weighted sum, 32 no global stores
};

__device__ __forceinline__ float row16_max(float v) {
  float o;
  o = row_xchg<0>(v); v = o > v ? o : v;
  o = row_xchg<1>(v); v = o > v ? o : v;
  o = row_xchg<2>(v); v = o > v ? o : v;
  o = row_xchg<3>(v); v = o > v ? o : v;
  return v;
}

template <bool DBG = false>
__global__ __launch_bounds__(Pool4Geom::THREADS) void pool4_fwd_kernel(Pool4Params p) {
  using Gm = Pool4Geom;
  const int dbg = DBG ? p.dbg : 0;        // the production instantiation folds every switch away
  constexpr int MT = Gm::MT;
  NR_SMEM_DECL(smem);
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  float* bq = (float*)(smem + Gm::W_BYTES);              // [QP] bias, [QP] query vector
  unsigned char* sc = smem + Gm::W_BYTES + Gm::BQ_BYTES + w * Gm::SC_WAVE;
  float* wl = (float*)sc;                                // the group's softmax weights: live only between the two uses of the scratch as a row buffer

  // ---- Wa rows -> LDS (once per workgroup), exactly as pool3_bwd_kernel places them ---------------------------------------------------------------
  for (int e = tid; e < Gm::NTQ * KSTEPS * 64; e += Gm::THREADS) {
    const int blk = e >> 6, ln = e & 63;
    const int row = (blk / KSTEPS) * 16 + (ln & 15), s = (blk % KSTEPS) * 4 + (ln >> 4);
    if (row < Gm::WROWS) *(u16x8*)(smem + row * Gm::WROW + ((s ^ w_swz(row)) * 16)) = *(const u16x8*)(p.Wap + (size_t)e * 8);
  }
  for (int i = tid; i < KP / 8; i += Gm::THREADS) *(u16x8*)(smem + Gm::WROWS * Gm::WROW + i * 16) = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  // tanh(a) = 1 - 2 / (exp(2 a) + 1): sum_n qv[n] tanh(a_n) = sum_n qv[n] - 2 sum_n qv[n] / (exp2(C2 a_n) + 1), and the softmax over the tokens of a
  // sequence does not see the first (token-independent) term.  LDS keeps C2 ba and -2 qv: per element one multiply-add, exp2, add, rcp, multiply-add.
  constexpr float C2 = 2.0f * 1.4426950408889634f;
  for (int i = tid; i < QP; i += Gm::THREADS) { bq[i] = p.bap[i] * C2; bq[QP + i] = -2.0f * p.qvp[i]; }
  __syncthreads();

  const int S = p.S;
  const int nslot = Gm::ROWS / S;                        // sequences per group (1 .. 4)
  const int grows = nslot * S;                           // live rows of a full group
  const int64_t n_groups = (p.n_seq + nslot - 1) / nslot, gstride = (int64_t)gridDim.x * Gm::NWAVE;
  int64_t grp = (int64_t)blockIdx.x * Gm::NWAVE + w;
  // the lane's four rows (row 16 m + li of the group): slot and position inside the sequence never change (a group starts at a sequence)
  int slot_m[MT], pos_m[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int r = m * 16 + li;
    const int s = (r >= S) + (r >= 2 * S) + (r >= 3 * S);
    slot_m[m] = r < grows ? s : -1;
    pos_m[m] = r - s * S;
  }
  const int valid = (p.valid > 0 && p.valid < S) ? p.valid : S;
  // output addressing: lane groups 0 / 1 hold the sequences (slots) 2 g, 2 g + 1 of the group; the other lanes' stores fall outside every resource
  constexpr uint32_t OOR = 0x80000000u;
  uint32_t oo[2], ob[2];
#pragma unroll
  for (int e2 = 0; e2 < 2; ++e2) {
    oo[e2] = g < 2 ? (uint32_t)(((2 * g + e2) * p.out_stride + li) * 4) : OOR;
    ob[e2] = g < 2 ? (uint32_t)(((2 * g + e2) * p.out_b_stride + li) * 2) : OOR;
  }

  // row pieces: xr[ks][t] = columns 32 ks + 8 (l & 3) .. + 7 of row 16 t + (l >> 2) (16 runs of 64 contiguous bytes per instruction)
  u16x8 xr[KSTEPS][MT];
  auto group_rsrc = [&](int64_t gi) -> BufRsrc {
    const int64_t seq0 = gi < n_groups ? gi * nslot : 0, tok0 = seq0 * S;
    const int64_t left = gi < n_groups ? (p.n_seq - seq0 < nslot ? p.n_seq - seq0 : nslot) * S : 0;       // live rows (the last group may hold fewer sequences; past the last group: none)
    return make_buf(p.ctx + tok0 * KP, (uint32_t)((left > 0 && !(dbg & 1) ? left : 0) * KP * 2));         // rows past the end read as zeros
  };
  auto load_ks = [&](BufRsrc rx, int ks) {
    int lq = l;
    NR_OPAQUE(lq);
#pragma unroll
    for (int t = 0; t < MT; ++t) xr[ks][t] = buf_load16<0>(rx, (uint32_t)((t * 16 + (lq >> 2)) * (KP * 2) + (lq & 3) * 16), (uint32_t)(ks * 64));
  };
  if (grp < n_groups) {
    const BufRsrc rx = group_rsrc(grp);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) load_ks(rx, ks);
  }

  // scratch addresses: row-piece writes (row 16 t + (l >> 2), slot l & 3) and fragment-shaped accesses (row 16 m + li, slot g)
  unsigned char* const pw = sc + (l >> 2) * 64 + (((l & 3) ^ sc_swz(l >> 2)) * 16);      // + t * 1024
  unsigned char* const fr = sc + li * 64 + ((g ^ sc_swz(li)) * 16);                        // + m * 1024   (16 m does not move the swizzle)

  while (grp < n_groups) {
    const int64_t seq0 = grp * nslot, tok0 = seq0 * S;
    const int live_seq = p.n_seq - seq0 < nslot ? (int)(p.n_seq - seq0) : nslot;
    // ---- row pieces -> B fragments (features 32 ks + 8 g .. + 7 of token 16 m + li), one k-step at a time through the scratch ----------------------
#pragma unroll
    for (int ks = (dbg & 2) ? KSTEPS : 0; ks < KSTEPS; ++ks) {
#pragma unroll
      for (int t = 0; t < MT; ++t) *(u16x8*)(pw + t * 1024) = xr[ks][t];
      wave_barrier();
#pragma unroll
      for (int m = 0; m < MT; ++m) xr[ks][m] = *(const u16x8*)(fr + m * 1024);
      wave_barrier();
    }
    // ---- scores: s[tok] = sum_n qv[n] tanh(x[tok] . Wa[n] + ba[n]): transposed product, the lane holds query rows 16 nt + 4 g .. + 3 of token li ----
    f32x2 sp2[MT];                            // two partial sums per token (packed fp32 arithmetic: even / odd query rows of the lane's four)
#pragma unroll
    for (int m = 0; m < MT; ++m) sp2[m] = f32x2{0.0f, 0.0f};
    // Wa fragment (nt, ks): row 16 nt + li (rows >= 200: the zero row), slot (4 ks + g) ^ w_swz(row) = 4 (ks ^ b) + c with b, c lane constants.  The reads
    // run as ONE pipeline over all 130 (nt, ks) steps, two steps ahead of the MFMAs that consume them (three fragment registers; asynchronous reads
    // with counted waits: see lds_read16_async), across the tanh phases too.  Instruction offsets are 16 bits: one base per 6 n-tiles and k-step parity.
    {
      int lq = l;
      NR_OPAQUE(lq);
      const int lg = lq >> 4, lr = lq & 15;
      const int wb = w_swz(lr) >> 2, wc = (lg ^ w_swz(lr)) & 3;
      const unsigned char* w0 = smem + lr * Gm::WROW + wc * 16;
      const unsigned char* w12 = smem + (12 * 16 + lr < Gm::WROWS ? 12 * 16 + lr : Gm::WROWS) * Gm::WROW + wc * 16;
      // (ks ^ wb) * 64 = ks * 64 + 64 wb for even ks, ks * 64 - 64 wb for odd ks
      const unsigned char* we[3] = {w0 + 64 * wb, w0 + 6 * 16 * Gm::WROW + 64 * wb, w12 + 64 * wb};
      const unsigned char* wo[3] = {w0 - 64 * wb, w0 + 6 * 16 * Gm::WROW - 64 * wb, w12 - 64 * wb};
      constexpr int NSTEP = Gm::NTQ * KSTEPS;
      u16x8 af[3];
      auto issue = [&](auto tag) {
        constexpr int e = decltype(tag)::value, nt = e / KSTEPS, ks = e % KSTEPS;
        constexpr int off = (nt % 6) * 16 * Gm::WROW + ks * 64;
        af[e % 3] = lds_read16_async<off>((ks & 1) ? wo[nt / 6] : we[nt / 6]);
      };
      issue(StaticIdx<0>{});
      issue(StaticIdx<1>{});
      f32x4 acc[MT];
      u16x8 bqr[2];                            // the n-tile's four (scaled) bias and query-vector entries: requested three k-steps before the tanh phase
      const unsigned char* bqp = smem + Gm::W_BYTES + 16 * lg;
      constexpr int KBQ = KSTEPS - 3;
      static_for<NSTEP>([&](auto tag) {
        constexpr int e = decltype(tag)::value, nt = e / KSTEPS, ks = e % KSTEPS;
        if (e + 2 < NSTEP) issue(StaticIdx<(e + 2 < NSTEP ? e + 2 : 0)>{});
        if (ks == KBQ) { bqr[0] = lds_read16_async<nt * 64>(bqp); bqr[1] = lds_read16_async<QP * 4 + nt * 64>(bqp); }
        constexpr int ahead = e + 2 < NSTEP ? 2 : NSTEP - 1 - e;        // fragment reads issued after the one of (nt, ks)
        NR_SCHED_BARRIER();
        NR_WAIT_LGKMCNT(ahead + (ks >= KBQ ? 2 : 0));                   // this wave's LDS operations return in order: at most the reads after (nt, ks) are outstanding
        NR_SCHED_BARRIER();
        if (!(dbg & 4)) {
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32_bf16(af[e % 3], xr[ks][m], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[m]);
        } else if (ks == 0) {
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (ks == KSTEPS - 1) {
          NR_SCHED_BARRIER();
          NR_WAIT_LGKMCNT(ahead);
          NR_SCHED_BARRIER();
          const f32x4 b4 = __builtin_bit_cast(f32x4, bqr[0]), q4 = __builtin_bit_cast(f32x4, bqr[1]);
          const f32x2 c2 = f32x2{C2, C2}, one2 = f32x2{1.0f, 1.0f};
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x2 a2 = f32x2{acc[m][2 * h], acc[m][2 * h + 1]} * c2 + f32x2{b4[2 * h], b4[2 * h + 1]};
              const f32x2 d2 = ((dbg & 8) ? a2 : f32x2{fast_exp2(a2[0]), fast_exp2(a2[1])}) + one2;
              const f32x2 r2 = (dbg & 8) ? d2 : f32x2{fast_rcp(d2[0]), fast_rcp(d2[1])};
              sp2[m] = r2 * f32x2{q4[2 * h], q4[2 * h + 1]} + sp2[m];
            }
        }
      });
    }
    // ---- softmax over the tokens of each sequence of the group (every lane group g ends up with the same numbers) ---------------------------------
    float wt[MT];
    {
      float sv[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const bool on = slot_m[m] >= 0 && slot_m[m] < live_seq && pos_m[m] < valid;
        const float tot = sum_rows4(sp2[m][0] + sp2[m][1]);       // (every lane takes part in the lane swaps)
        sv[m] = on ? tot : -3.0e38f;
      }
      float mx[Gm::MAXSLOT], sm[Gm::MAXSLOT];
#pragma unroll
      for (int s = 0; s < Gm::MAXSLOT; ++s) {
        float v = -3.0e38f;
#pragma unroll
        for (int m = 0; m < MT; ++m) v = (slot_m[m] == s && sv[m] > v) ? sv[m] : v;
        mx[s] = row16_max(v);
      }
      float e[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int s = slot_m[m] < 0 ? 0 : slot_m[m];
        const float mxs = s == 0 ? mx[0] : s == 1 ? mx[1] : s == 2 ? mx[2] : mx[3];
        e[m] = sv[m] > -1.0e38f ? fast_exp(sv[m] - mxs) : 0.0f;
      }
#pragma unroll
      for (int s = 0; s < Gm::MAXSLOT; ++s) {
        float v = 0.0f;
#pragma unroll
        for (int m = 0; m < MT; ++m) v += slot_m[m] == s ? e[m] : 0.0f;
        sm[s] = sum_row16(v);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int s = slot_m[m] < 0 ? 0 : slot_m[m];
        const float sms = s == 0 ? sm[0] : s == 1 ? sm[1] : s == 2 ? sm[2] : sm[3];
        wt[m] = e[m] > 0.0f ? e[m] / sms : 0.0f;
      }
    }
    // weights -> LDS (one row of 64 floats) for the A fragments below, and to memory (row r of the group = token tok0 + r: sequences are contiguous)
    if (g == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m) wl[m * 16 + li] = wt[m];
    }
    {                                         // (rows of sequences past the last one and the rows above `grows`: outside the resource, the store vanishes)
      const BufRsrc r_aw = make_buf(p.attn_w != nullptr ? p.attn_w + tok0 : nullptr, (uint32_t)(p.attn_w != nullptr && !(dbg & 32) ? live_seq * S * 4 : 0));
#pragma unroll
      for (int m = 0; m < MT; ++m) buf_store4f(r_aw, g == 1 ? (uint32_t)((m * 16 + li) * 4) : OOR, wt[m]);
    }
    wave_barrier();
    // ---- A fragments of the weighted sum, one per pair of token tiles tp (rows 32 tp .. + 31): row li = (slot li >> 1, part li & 1), k-slot (g, j) =
    // token 32 tp + 8 g + j: the hi / lo bf16 part of its weight when the token belongs to that slot, else 0 -------------------------------------------
    u16x8 aw[2];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      const f32x4 w0 = *(const f32x4*)(wl + tp * 32 + g * 8), w1 = *(const f32x4*)(wl + tp * 32 + g * 8 + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = tp * 32 + g * 8 + j;
        const int st = (t >= S) + (t >= 2 * S) + (t >= 3 * S);
        const float v = j < 4 ? w0[j] : w1[j - 4];
        const u16 hi = f2bf(v);
        const u16 part = (li & 1) ? f2bf(v - bf2f(hi)) : hi;
        aw[tp][j] = (st == (li >> 1) && t < grows) ? part : (u16)0;
      }
    }
    wave_barrier();                           // the weights have been read: the scratch becomes a row buffer again
    // ---- y[slot][d] = sum_tok w[tok] x[tok][d]: per k-step the fragments return to the scratch as rows and come back transposed (k = token).  A
    // k-step's registers are free once its rows are in the scratch: the NEXT group's rows of that k-step are requested right there, so their
    // latency hides behind the rest of this phase ---------------------------------------------------------------------------------------------------
    const BufRsrc rx_next = group_rsrc(grp + gstride);
    // outputs: rows seq0 .. seq0 + live_seq - 1 through buffer resources (a null output: an empty resource), no branch around any store
    const bool st_on = !(dbg & 32);
    const BufRsrc r_out = make_buf(p.out != nullptr ? p.out + seq0 * p.out_stride : nullptr,
                                   (uint32_t)(p.out != nullptr && st_on ? ((int64_t)(live_seq - 1) * p.out_stride + D) * 4 : 0));
    const BufRsrc r_outb = make_buf(p.out_b != nullptr ? p.out_b + seq0 * p.out_b_stride : nullptr,
                                    (uint32_t)(p.out_b != nullptr && st_on ? ((int64_t)(live_seq - 1) * p.out_b_stride + KP) * 2 : 0));
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (dbg & 16) { load_ks(rx_next, ks); continue; }
#pragma unroll
      for (int m = 0; m < MT; ++m) *(u16x8*)(fr + m * 1024) = xr[ks][m];
      wave_barrier();
      load_ks(rx_next, ks);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
          // piece P_{li} of the lane's group: row k0 + (li >> 2), columns 16 h + 4 (li & 3) .. + 3 -> slot 2 h + ((li & 3) >> 1), byte (li & 1) * 8
          const int q = li & 3;
          auto piece = [&](int k0) -> const u16* {
            const int r = k0 + (li >> 2);
            return (const u16*)(sc + r * 64 + (((2 * h + (q >> 1)) ^ sc_swz(r)) * 16) + (q & 1) * 8);
          };
          const u16x4 lo = lds_tr16_b64(piece(tp * 32 + g * 8)), hi = lds_tr16_b64(piece(tp * 32 + g * 8 + 4));
          acc = mfma_16x16x32_bf16(aw[tp], cat8(lo, hi), acc);
        }
        // accumulator rows 4 g + r = (slot 2 g + (r >> 1), part r & 1): lane groups 0 and 1 hold slots 0 - 1 and 2 - 3 for feature 32 ks + 16 h + li
        const int col0 = ks * 32 + h * 16;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          const float y = acc[2 * e2] + acc[2 * e2 + 1];
          if (col0 < D) buf_store4f(r_out, (col0 + 16 <= D || col0 + li < D) ? oo[e2] : OOR, y, (uint32_t)(col0 * 4));
          buf_store2(r_outb, ob[e2], f2bf(y), (uint32_t)(col0 * 2));        // (col D: sum of the weights = 1.0; cols > D: 0)
        }
      }
      wave_barrier();
    }
    grp += gstride;
  }
}

}  // namespace nr
