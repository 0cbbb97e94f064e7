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
// LDS for its whole life (200 rows x 640 B, loaded once) and serves BOTH products -- x Wa^T through plain 16-byte fragment reads, dpre @ Wa
// through the transposing read ds_read_b64_tr_b16 of the same rows (no Wa^T operand, no weight streaming, no barrier inside the loop).
// A wave's 48 ctx rows live in registers as MFMA B fragments (as in k_pool2.h).
//
// Global memory is only touched LANE-CONTIGUOUSLY.  The MFMA layouts put consecutive lanes on consecutive TOKENS (rows 640 B apart): a fragment-
// shaped load or an accumulator-shaped store is 64 separate requests to the texture addresser, and the first version of this kernel, like
// k_pool2.h, spent two thirds of its time issuing them (phase switches: 428 us, 259 without the stores, 329 without the loads; the matrix
// and vector pipes together busy 40 % of the time).  Now every vector-memory instruction moves 16 bytes per lane with four consecutive lanes
// on one row -- 16 runs of 64 contiguous bytes -- and the change of layout happens in 3 KB of wave-private LDS: loaded pieces are written
// there as they arrive and read back as fragments; accumulator tiles are written there two column tiles at a time and read back as row
// pieces.  (Same wave on both sides: LDS operations of a wave execute in order, no barrier.)
#pragma once
#include "nr_common.h"
#include "k_additive_fwd.h"

namespace nr {

struct Pool3Geom {
  static constexpr int NWAVE = 8, THREADS = NWAVE * 64;
  static constexpr int MT = 3, ROWS = MT * 16;   // token rows per wave and iteration
  static constexpr int NTQ = QP / 16;            // 13 n-tiles of the query dim
  static constexpr int KS2 = QKP / 32;           // 7 k-steps of the dctx product
  static constexpr int NTD = (D + 15) / 16;      // 19 feature tiles of dctx
  static constexpr int WROWS = 200;              // rows of Wa kept (query_vector_dim <= 200 real rows of the 208 packed ones); row WROWS = zeros
  static constexpr int WROW = KP * 2;            // 640 B per row, 16-byte slots swizzled (w_swz): no padding
  static constexpr int W_BYTES = (WROWS + 1) * WROW;               // 128,640
  static constexpr int BQ_BYTES = 2 * QP * 4;    // bias and query vector
  static constexpr int DQ_BYTES = NWAVE * QP * 4;
  static constexpr int SC_WAVE = ROWS * 64;      // 3,072 B of layout-change scratch per wave: [48 rows][4 slots of 16 B], slots swizzled (sc_swz)
  static constexpr int EH_BYTES = 2 * 64 * 16;   // the two selector fragments of the activation-gradient form (see there)
  static constexpr int SMEM = W_BYTES + BQ_BYTES + DQ_BYTES + NWAVE * SC_WAVE + EH_BYTES;      // 163,584
  static_assert(SMEM <= 163840 && QP - WROWS == 8, "LDS; the dropped rows are the upper half of the last n-tile");
};

// Swizzle of the 16-byte slots of Wa row r (XOR into the slot index, inside aligned groups of 8 slots = 128 B): slot ^= r & 6.  The chip services a
// b128 read in four NON-contiguous 16-lane groups and a transposing read in two 32-lane halves, on 64 banks (MI355X_MICROARCH.md, LDS): under that
// model (tools/lds_bank_model.py) this swizzle is conflict-free for both access shapes of the two products.  [Rounds 4-5 shipped
// ((r & 3) << 1) | ((r >> 2) & 1), laid out for 8-lane groups on 32 banks: two-way on both shapes -- 37 % of the kernel's LDS cycles were conflicts.]
__device__ __forceinline__ int w_swz(int r) { return r & 6; }
// scratch rows are 64 B (4 slots): consecutive row pairs fill the 128 B of banks, the pair index picks the slot rotation
__device__ __forceinline__ int sc_swz(int r) { return (r >> 1) & 3; }

template <int V> struct IntTag { static constexpr int value = V; };

struct Pool3Params {
  const u16* ctx;        // [n_tok][KP]  forward input of the additive layer
  const u16* Wap;        // [QP][KP] tile order
  const float* bap;      // [QP]
  const float* qvp;      // [QP]
  const float* attn_w;   // [n_tok]  forward attention weights
  const float* g_out;    // [n_seq][D], row r at g_out + r * g_row_bytes / 4 (a column block of wider rows: LSTUR's [category | subcategory | title] gradient)
  uint32_t g_row_bytes;  // >= D * 4, a multiple of 16
  const float* tot;      // [n_seq]  g_out[seq] . y[seq]
  u16* dpre;             // [n_tok][QP] bf16
  float* dq_part;        // [gridDim.x][QP]
  u16* dctx;             // optional: bf16 [n_tok][KP] = dpre @ Wa (columns < D written)
  u16* dy_pad;           // optional, instead of dctx: bf16 seqpad rows (tok + seq + 1) = (dpre @ Wa + w (x) g_out) * [ctx != 0] * act_scale
  float act_scale;
  int64_t n_seq;
  int64_t n_tok;         // n_seq * S < 2^31
  uint32_t S;            // >= 2
  uint32_t s_magic;      // floor(2^32 / S) + 1
  unsigned long long* stamps;   // DBG instantiation only (nr_debug_pool3_stamps): [4 workgroups][2 waves][8 iterations][8] cycle-counter stamps
  int dbg;               // DBG instantiation only (NR_POOL_DEBUG, tools/pool3_phases.py): 1 no ctx loads, 2 no dw phase, 4 no projection MFMAs,
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

// The wave's 48 ctx rows as row pieces: xr[ks][t] = columns 32 ks + 8 (l & 3) .. + 7 of row 16 t + (l >> 2) -- per instruction 16 runs of 64
// contiguous bytes.  pool3_to_fragments() turns them into B-operand fragments in place.
__device__ __forceinline__ BufRsrc pool3_x_rsrc(const u16* __restrict__ ctx, int64_t tok0, int64_t n_tok) {
  const int64_t left = n_tok - tok0;                                    // (<= 0: past the last group, or a phase switch of the debug build)
  const int nrows = left < Pool3Geom::ROWS ? (left > 0 ? (int)left : 0) : Pool3Geom::ROWS;
  return make_buf(ctx + (left > 0 ? tok0 : 0) * KP, (uint32_t)(nrows * KP * 2));      // rows past the end read as zeros
}
__device__ __forceinline__ void pool3_load_x_ks(BufRsrc rx, int ks, u16x8 (&x)[Pool3Geom::MT]) {
  int lq = lane_id();
  NR_OPAQUE(lq);                                                        // (offsets rebuilt from an opaque lane id: not worth three registers)
#pragma unroll
  for (int t = 0; t < Pool3Geom::MT; ++t) x[t] = buf_load16<0>(rx, (uint32_t)((t * 16 + (lq >> 2)) * (KP * 2) + (lq & 3) * 16), (uint32_t)(ks * 64));
}
__device__ __forceinline__ void pool3_load_x(const u16* __restrict__ ctx, int64_t tok0, int64_t n_tok, u16x8 (&xr)[KSTEPS][Pool3Geom::MT]) {
  const BufRsrc rx = pool3_x_rsrc(ctx, tok0, n_tok);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) pool3_load_x_ks(rx, ks, xr[ks]);
}

// row pieces -> fragments through the wave's scratch, one k-step (48 rows x 64 B) at a time: afterwards xr[ks][m] holds features
// 32 ks + 8 g .. + 7 of token 16 m + li (lane = 16 g + li)
__device__ __forceinline__ void pool3_to_fragments(unsigned char* sc, u16x8 (&xr)[KSTEPS][Pool3Geom::MT]) {
  const int l = lane_id(), g = l >> 4, li = l & 15;
  unsigned char* wr = sc + (l >> 2) * 64 + (((l & 3) ^ sc_swz(l >> 2)) * 16);         // + t * 1024
  const unsigned char* rd = sc + li * 64 + ((g ^ sc_swz(li)) * 16);                     // + m * 1024   (16 m does not move the swizzle)
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
    for (int t = 0; t < Pool3Geom::MT; ++t) *(u16x8*)(wr + t * 1024) = xr[ks][t];
    wave_barrier();
#pragma unroll
    for (int m = 0; m < Pool3Geom::MT; ++m) xr[ks][m] = *(const u16x8*)(rd + m * 1024);
    wave_barrier();
  }
}

// ACT: the fused activation gradient (dy_pad) instead of dctx -- a compile-time form
template <bool ACT, bool DBG = false>
__global__ __launch_bounds__(Pool3Geom::THREADS) void pool3_bwd_kernel(Pool3Params p) {
  using Gm = Pool3Geom;
  const int dbg = DBG ? p.dbg : 0;        // the production instantiation folds every switch away
  constexpr int MT = Gm::MT;
  NR_SMEM_DECL(smem);
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  float* bq = (float*)(smem + Gm::W_BYTES);              // [QP] bias, [QP] query vector
  float* dqp = bq + 2 * QP + w * QP;                     // this wave's dq accumulator row
  unsigned char* sc = smem + Gm::W_BYTES + Gm::BQ_BYTES + Gm::DQ_BYTES + w * Gm::SC_WAVE;
  const u16x4 Z4 = u16x4{0, 0, 0, 0};
  const bool with_dctx = (ACT || p.dctx != nullptr) && !(dbg & 16);       // the stand-alone AdditiveAttention backward stops at dpre / dq

  // ---- Wa rows -> LDS (once per workgroup): the tile-ordered operand is read front to back (piece e = block (row tile, k-step), lane) ------------
  for (int e = tid; e < Gm::NTQ * KSTEPS * 64; e += Gm::THREADS) {
    const int blk = e >> 6, ln = e & 63;
    const int row = (blk / KSTEPS) * 16 + (ln & 15), s = (blk % KSTEPS) * 4 + (ln >> 4);
    if (row < Gm::WROWS) *(u16x8*)(smem + row * Gm::WROW + ((s ^ w_swz(row)) * 16)) = *(const u16x8*)(p.Wap + (size_t)e * 8);
  }
  for (int i = tid; i < KP / 8; i += Gm::THREADS) *(u16x8*)(smem + Gm::WROWS * Gm::WROW + i * 16) = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  // tanh through r = 1 / (exp(2 a) + 1):  t = 1 - 2 r,  1 - t^2 = 4 r (1 - r).  LDS keeps C2 ba (the exp2 argument is one multiply-add away from the
  // accumulator) and 4 qv; per element: multiply-add, exp2, add, rcp, r - r^2, two multiplies for dpre and one multiply-add for dq, all but the two
  // transcendentals as packed fp32 pairs (round 6; before: ~9 scalar VALU operations + the same two transcendentals)
  constexpr float C2 = 2.0f * 1.4426950408889634f;
  for (int i = tid; i < QP; i += Gm::THREADS) { bq[i] = p.bap[i] * C2; bq[QP + i] = 4.0f * p.qvp[i]; }
  for (int i = tid; i < Gm::NWAVE * QP; i += Gm::THREADS) bq[2 * QP + i] = 0.0f;
  unsigned char* eht = smem + Gm::SMEM - Gm::EH_BYTES;
  if (ACT && tid < 128) {                 // E_h as an A fragment: lane (d = li, g) holds E_h[d][8 g + j] = [8 g + j == 16 h + d]
    const int h = tid >> 6, ln = tid & 63;
    u16x8 e;
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = ((ln >> 4) == 2 * h + ((ln >> 3) & 1) && j == (ln & 7)) ? BF16_ONE : (u16)0;
    *(u16x8*)(eht + tid * 16) = e;
  }
  __syncthreads();

  const int64_t n_groups = (p.n_tok + Gm::ROWS - 1) / Gm::ROWS, gstride = (int64_t)gridDim.x * Gm::NWAVE;
  int64_t grp = (int64_t)blockIdx.x * Gm::NWAVE + w;     // neighbouring groups run on one CU at the same time: their partial lines merge in its L2
  u16x8 xr[KSTEPS][MT];
  const int64_t n_load = (dbg & 1) ? 0 : p.n_tok;
  if (grp < n_groups) pool3_load_x(p.ctx, grp * Gm::ROWS, n_load, xr);

  // lane constants of the scratch: accumulator-shaped writes (token li of tile m, 4 columns 4 g ..: half h of a column-tile pair) and row-piece
  // reads (row 16 t + (l >> 2), slot l & 3 = columns 8 (l & 3) .. + 7 of the pair)
  const int cw0 = li * 64 + ((((g >> 1) ^ sc_swz(li))) * 16) + (g & 1) * 8;           // + m * 1024, ^ 32 for the second tile of a pair
  const int pr0 = (l >> 2) * 64 + (((l & 3) ^ sc_swz(l >> 2)) * 16);                  // + t * 1024

  int it = 0;
  auto stamp = [&](int k) {               // (timeline of the first iterations of a few waves: tools/pool3_phases.py --timeline)
    if (DBG && p.stamps != nullptr && l == 0 && w < 2 && blockIdx.x < 4 && it < 8)
      p.stamps[(((size_t)blockIdx.x * 2 + w) * 8 + it) * 8 + k] = __builtin_readcyclecounter();
  };
  while (grp < n_groups) {
    const int64_t tok0 = grp * Gm::ROWS;
    stamp(0);
    pool3_to_fragments(sc, xr);
    stamp(1);
    // ---- the lane's three rows (row 16 m + li of the group): ds = w (g . x - tot).  Sequence index and forward weight are dropped again after this
    // phase (the ACT epilogue fetches them a second time) instead of living through the projection ------------------------------------------
    const int nrows = p.n_tok - tok0 < Gm::ROWS ? (int)(p.n_tok - tok0) : Gm::ROWS;       // live rows of the group (wave-uniform)
    const uint32_t t0u = (uint32_t)tok0;
    auto seq_of = [&](int row) -> uint32_t {  // sequence of the group's row (rows past the end: of row 0)
      const uint32_t tk = t0u + (row < nrows ? (uint32_t)row : 0u);
      uint32_t q = mulhi_u32(tk, p.s_magic);   // floor(tk / S) or one more (tk < 2^31)
      return q - ((q * p.S > tk) ? 1u : 0u);
    };
    const int rowl = l >> 2;                   // + 16 t: the lane's row in the row-piece layout
    // every global access of the group: a buffer resource on the group's rows (rows past the end: loads give 0, stores vanish) + 32-bit offsets
    const BufRsrc r_aw = make_buf(p.attn_w + tok0, (uint32_t)(nrows * 4));
    const BufRsrc r_g = make_buf(p.g_out, (uint32_t)((p.n_seq - 1) * p.g_row_bytes + D * 4)), r_tot = make_buf(p.tot, (uint32_t)(p.n_seq * 4));
    const BufRsrc r_dpre = make_buf(p.dpre + tok0 * QP, (uint32_t)(nrows * QP * 2));
    // dw[tok] = g_out[seq(tok)] . x[tok] on the matrix core: the group's rows belong to at most 8 sequences (S >= 7; 16 for S >= 3: a second tile), "slots" sq0 .. sq0 + 7.  A tile:
    // row li = (slot li >> 1, part li & 1) of g_out split into two bf16 numbers (g = hi + lo to 2^-17: the products with the bf16 rows are exact,
    // the accumulation fp32); B = the rows' own fragments.  A token then picks the two accumulator rows of ITS slot.  [As 3 x 80 multiply-adds per
    // lane with 60 sixteen-byte loads of g_out in front of them this phase took a fifth of the kernel -- latency of six load batches per group.]
    const uint32_t sq0 = (uint32_t)uniform((int)seq_of(0));
    const bool two = p.S < 7;                 // sequences shorter than 7 tokens (NAML's 4 views): up to 16 per group -> a second tile of slots 8 .. 15 (round 6)
    float ds[MT];
    {
      f32x4 accg[MT], accg2[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) { accg[m] = f32x4{0.f, 0.f, 0.f, 0.f}; accg2[m] = accg[m]; }
      uint32_t go = (sq0 + (uint32_t)(li >> 1)) * p.g_row_bytes + (uint32_t)(g * 32);      // (slots past the last sequence: outside the buffer, zeros)
      NR_OPAQUE(go);
      auto gstep = [&](auto ks_tag) {
        constexpr int ks = decltype(ks_tag)::value;
        auto slots = [&](uint32_t off, f32x4 (&acc)[MT]) {
          f32x4 g0 = f32x4{0.f, 0.f, 0.f, 0.f}, g1 = g0;
          if (ks * 32 + 32 <= D || ks * 32 + g * 8 + 4 <= D) g0 = buf_load16f<ks * 128>(r_g, off);          // (columns >= D: the bias column of ctx, zeros)
          if (ks * 32 + 32 <= D || ks * 32 + g * 8 + 8 <= D) g1 = buf_load16f<ks * 128 + 16>(r_g, off);
          u16x8 fr;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const u16 h0 = f2bf(g0[j]), h1 = f2bf(g1[j]);
            fr[j] = (li & 1) ? f2bf(g0[j] - bf2f(h0)) : h0;
            fr[4 + j] = (li & 1) ? f2bf(g1[j] - bf2f(h1)) : h1;
          }
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32_bf16(fr, xr[ks][m], acc[m]);
        };
        slots(go, accg);
        if (two) slots(go + 8u * p.g_row_bytes, accg2);
      };
      if (!(dbg & 2)) {
        gstep(IntTag<0>{}); gstep(IntTag<1>{}); gstep(IntTag<2>{}); gstep(IntTag<3>{}); gstep(IntTag<4>{});
        gstep(IntTag<5>{}); gstep(IntTag<6>{}); gstep(IntTag<7>{}); gstep(IntTag<8>{}); gstep(IntTag<9>{});
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint32_t sq = seq_of(m * 16 + li), slot = sq - sq0;                 // accumulator rows 4 g + r = (slot 2 g + (r >> 1), part r & 1)
        const f32x4 ag = slot < 8u ? accg[m] : accg2[m];
        const float pair = (slot & 1) ? ag[2] + ag[3] : ag[0] + ag[1];
        const float dw = sum_rows4((int)((slot & 7u) >> 1) == g ? pair : 0.0f);
        ds[m] = buf_load4f(r_aw, (uint32_t)((m * 16 + li) * 4)) * (dw - buf_load4f(r_tot, sq * 4u));      // (rows past the end: weight 0)
      }
    }

    stamp(2);
    // ---- t = tanh(x Wa^T + ba);  dpre = ds qv (1 - t^2) (kept packed in registers + stored);  dq += ds t ---------------------------------------
    f32x2 dsm2[MT];                           // -2 ds (dq += ds t = ds - 2 ds r: the first term once per n-tile, below)
    float dssum = 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m) { dsm2[m] = f32x2{-2.0f * ds[m], -2.0f * ds[m]}; dssum += ds[m]; }
    u16x4 dpk[Gm::NTQ][MT];
#pragma unroll
    for (int nt = 0; nt < Gm::NTQ; ++nt) {
      const int wrow = nt * 16 + 4 * g;
      // (every lane-constant address below is rebuilt per n-tile from an opaque lane id, ~8 integer instructions: kept across the loop they are the
      // first values the register allocator spills, and a spill reload sits in the in-order memory counter behind the dpre stores)
      int lq = l;
      NR_OPAQUE(lq);
      const int lg = lq >> 4, lr = lq & 15;
      const int bo = Gm::W_BYTES + 16 * lg;
      const f32x4 b4 = *(const f32x4*)(smem + bo + nt * 64), q4 = *(const f32x4*)(smem + bo + QP * 4 + nt * 64);
      f32x4 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      // fragment (nt, ks) of Wa: row 16 nt + li (rows >= 200: the zero row), slot (4 ks + g) ^ w_swz(row) = 4 (ks ^ b) + c with b, c lane constants
      const int wr_ = nt * 16 + lr < Gm::WROWS ? nt * 16 + lr : Gm::WROWS;
      const int wb = w_swz(lr) >> 2, wc = (lg ^ w_swz(lr)) & 3;                     // (16 nt does not move the swizzle; the zero row is zero in every slot)
      const unsigned char* wp = smem + wr_ * Gm::WROW + wc * 16;
      auto frag = [&](int ks) -> u16x8 { return *(const u16x8*)(wp + ((ks ^ wb) * 64)); };
      u16x8 a = frag(0);
#pragma unroll
      for (int ks = (dbg & 4) ? KSTEPS : 0; ks < KSTEPS; ++ks) {
        const u16x8 an = ks + 1 < KSTEPS ? frag(ks + 1) : a;
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32_bf16(a, xr[ks][m], acc[m]);
        a = an;
      }
      f32x2 dq2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
      const f32x2 c2 = f32x2{C2, C2}, one2 = f32x2{1.0f, 1.0f};
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4 dp;
        const f32x2 ds2 = f32x2{ds[m], ds[m]};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const f32x2 a2 = f32x2{acc[m][2 * hh], acc[m][2 * hh + 1]} * c2 + f32x2{b4[2 * hh], b4[2 * hh + 1]};
          const f32x2 d2 = ((dbg & 8) ? a2 : f32x2{fast_exp2(a2[0]), fast_exp2(a2[1])}) + one2;
          const f32x2 r2 = (dbg & 8) ? d2 : f32x2{fast_rcp(d2[0]), fast_rcp(d2[1])};
          const f32x2 u2 = r2 - r2 * r2;                               // (1 - t^2) / 4
          const f32x2 x2 = (u2 * f32x2{q4[2 * hh], q4[2 * hh + 1]}) * ds2;
          dp[2 * hh] = x2[0]; dp[2 * hh + 1] = x2[1];
          dq2[hh] = dsm2[m] * r2 + dq2[hh];
        }
        dpk[nt][m] = pack4(dp);
      }
      const f32x4 dq4 = f32x4{dq2[0][0] + dssum, dq2[0][1] + dssum, dq2[1][0] + dssum, dq2[1][1] + dssum};
      // dq: the tile's tokens live in the 16 lanes of a row -> DPP sums; lane li < 4 of every row then adds query row 4 g + li to the wave's own
      // LDS accumulator with ONE fire-and-forget ds_add_f32 [before: four read - wait - add - write round trips per n-tile under a lane mask]
      if (!(dbg & 128)) {
        const float v0 = sum_row16(dq4[0]), v1 = sum_row16(dq4[1]), v2 = sum_row16(dq4[2]), v3 = sum_row16(dq4[3]);
        const float v = li == 0 ? v0 : li == 1 ? v1 : li == 2 ? v2 : v3;
        if (li < 4) lds_add_f32(dqp + wrow + li, v);
      }
      // dpre rows leave two n-tiles at a time: 32 columns = 64 contiguous bytes of every row
      if ((nt & 1) || nt == Gm::NTQ - 1) {
        const int nt0 = nt & ~1;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          *(u16x4*)(sc + m * 1024 + cw0) = dpk[nt0][m];
          if (nt & 1) *(u16x4*)(sc + m * 1024 + (cw0 ^ 32)) = dpk[nt][m];
        }
        wave_barrier();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const u16x8 v = *(const u16x8*)(sc + t * 1024 + pr0);
          int lq = l;                         // (the offset is recomputed here from an opaque lane id: hoisted, the 3 of them -- 21 with the column --
          NR_OPAQUE(lq);                      //  are spilled, and a spill reload in front of a store waits for every store before it)
          const uint32_t o = (uint32_t)((t * 16 + (lq >> 2)) * (QP * 2) + (lq & 3) * 16);
          if (((nt & 1) || (l & 3) < 2) && !(dbg & 32)) buf_store16(r_dpre, o, v, (uint32_t)(nt0 * 32));
        }
        wave_barrier();
      }
      if (nt == 5) stamp(3);
      NR_SCHED_BARRIER();                     // keep the n-tiles apart: interleaving them stretches the live ranges past the register file
    }
    stamp(4);

    // ---- the next group's rows: the fragments are free, their latency hides behind the dctx product (ACT needs them for its mask: below) ----------
    const int64_t nxt = grp + gstride;
    if (!ACT && nxt < n_groups) pool3_load_x(p.ctx, nxt * Gm::ROWS, n_load, xr);
    stamp(5);

    // ---- dctx[tok][:] = dpre[tok][:] @ Wa: A = Wa^T rows of a feature tile by transposing reads of the SAME LDS rows (k-slot (g, j) of k-step ks
    // stands for query index 32 ks + 16 (j / 4) + 4 g + j % 4: the lane's packed dpre registers of tiles 2 ks, 2 ks + 1 are the B fragment) -----
    if (with_dctx) {
      // piece P_{4 r + qq} of the lane's group: row 32 ks + 16 h + 4 g + r, columns 16 dt + 4 qq .. + 3 -> slot 2 dt + (qq >> 1), swizzled by
      // w_swz(row) = row & 6 = 4 (g & 1) + 2 (r >> 1): the low bit of the slot stays, its next two bits = (dt & 3) ^ (2 (g & 1) + (r >> 1))
      const int tr_r = li >> 2, tr_q = li & 3;
      const int tr_lo = ((tr_q >> 1) * 16) + (tr_q & 1) * 8;
      auto product = [&](int dt, f32x4 (&acc)[MT], bool zero) {
        int lq = l;                           // (opaque lane id: with the tile loop unrolled, one hoisted address per feature tile otherwise)
        NR_OPAQUE(lq);
        const int r_ = (lq >> 2) & 3;
        const unsigned char* wq = smem + (4 * (lq >> 4) + r_) * Gm::WROW + (dt >> 2) * 128 + (((dt & 3) ^ ((((lq >> 4) & 1) << 1) | (r_ >> 1))) * 32) + tr_lo;
        const unsigned char* wz = smem + Gm::WROWS * Gm::WROW;        // rows >= 200 of the last k-step: the zero row
        if (zero) {
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < Gm::KS2; ++ks) {
          const bool in0 = ks * 32 + 15 < Gm::WROWS || ks * 32 + 4 * (lq >> 4) + r_ < Gm::WROWS;
          const u16x4 lo = lds_tr16_b64((const u16*)(in0 ? wq + (ks * 32) * Gm::WROW : wz));
          const u16x4 hi = 2 * ks + 1 < Gm::NTQ ? lds_tr16_b64((const u16*)(wq + (ks * 32 + 16) * Gm::WROW)) : lo;     // (the partner of the last n-tile is zero on the dpre side)
          const u16x8 a = cat8(lo, hi);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32_bf16(a, cat8(dpk[2 * ks][m], 2 * ks + 1 < Gm::NTQ ? dpk[2 * ks + 1][m] : Z4), acc[m]);
        }
      };
      // two feature tiles (32 columns) of the wave's rows leave together: accumulator-shaped 8-byte writes into the scratch, 16-byte row pieces out
      uint32_t seql[MT];                      // ACT: sequence of row 16 t + (l >> 2) (the seqpad row is tok + seq + 1)
      if (ACT) {
#pragma unroll
        for (int t = 0; t < MT; ++t) seql[t] = seq_of(t * 16 + rowl);
      }
      // ACT: seqpad row of a token = tok + seq + 1; the resource starts at the seqpad row of the group's first token
      const int64_t out_rows = ACT ? (p.n_tok + p.n_seq) - (tok0 + sq0) : nrows;
      const BufRsrc r_out = make_buf(ACT ? p.dy_pad + (tok0 + sq0 + 1) * KP : p.dctx + tok0 * KP,
                                     (uint32_t)((out_rows < 3000000 ? out_rows : 3000000) * (KP * 2)));
      auto flush = [&](int dt0, bool pair) {
        wave_barrier();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const u16x8 v = *(const u16x8*)(sc + t * 1024 + pr0);
          const int col = dt0 * 16 + (l & 3) * 8;
          if ((!ACT || t * 16 + rowl < nrows) && (pair || (l & 3) < 2) && !(dbg & 32)) {
            int lq = l;
            NR_OPAQUE(lq);
            const uint32_t row = (uint32_t)(t * 16 + (lq >> 2)) + (ACT ? seql[t] - sq0 : 0u);
            const uint32_t off = row * (uint32_t)(KP * 2) + (uint32_t)((lq & 3) * 16);
            if (col + 8 <= D) buf_store16(r_out, off, v, (uint32_t)(dt0 * 32));
            else if (col + 4 <= D) buf_store8(r_out, off, u16x4{v[0], v[1], v[2], v[3]}, (uint32_t)(dt0 * 32));
          }
        }
        wave_barrier();
      };
      if (!ACT) {
#pragma nounroll
        for (int dt0 = 0; dt0 < Gm::NTD; dt0 += 2) {
          f32x4 acc[MT];
          product(dt0, acc, true);
#pragma unroll
          for (int m = 0; m < MT; ++m) *(u16x4*)(sc + m * 1024 + cw0) = pack4(acc[m]);
          const bool pair = dt0 + 1 < Gm::NTD;
          if (pair) {
            product(dt0 + 1, acc, true);
#pragma unroll
            for (int m = 0; m < MT; ++m) *(u16x4*)(sc + m * 1024 + (cw0 ^ 32)) = pack4(acc[m]);
          }
          flush(dt0, pair);
        }
      } else {
        // The relu / dropout mask of the conv stage is [activation != 0], and the activations are this wave's own B fragments -- in operand layout,
        // where the epilogue needs accumulator layout.  The matrix core moves them: E_h (16 x 32, E[d][k] = [k == 16 h + d]) times the
        // fragment of k-step dt / 2 is the tile x[tok][16 dt + d] exactly (1.0 x bf16 in fp32), lane for lane where acc holds the same element.
        // The direct term w[tok] g_out[seq(tok)][d] enters the accumulators as one more k-step: A[d][8 s + j] = parts of g_out[sq0 + s][d] (the group's
        // rows belong to at most 4 sequences: S >= 16), B[8 s + j][tok] = parts of w[tok] where s is the token's slot, else 0, both numbers split
        // into two bf16: hi hi + lo hi + hi lo.  One 4-byte load per lane and feature tile (row sq0 + g, column 16 dt + li) instead of a
        // 16-byte g_out piece per lane, tile AND row tile with its wait right behind the previous tile's stores.
        u16x4 bd[MT];                          // k-slots 0..3 of the lane's eight (the other four are zero)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float a_ = buf_load4f(r_aw, (uint32_t)((m * 16 + li) * 4));
          const u16 ah = f2bf(a_), al = f2bf(a_ - bf2f(ah));
          const bool mine = (int)(seq_of(m * 16 + li) - sq0) == g;
          bd[m] = u16x4{mine ? ah : (u16)0, mine ? ah : (u16)0, mine ? al : (u16)0, 0};
        }
        uint32_t gq = (sq0 + (uint32_t)g) * p.g_row_bytes + (uint32_t)(li * 4);
        NR_OPAQUE(gq);
        auto gload = [&](int dt) -> float { return dt * 16 + li < D ? buf_load4f(r_g, gq + (uint32_t)(dt * 64)) : 0.0f; };
        float gn = gload(0), gn2 = gload(1);      // two tiles ahead: the wait for a value then never covers the stores and row requests of the tile before
                                                 // (the memory counter is in order: one tile ahead, every second wait was a wait for the HBM latency of the rows)
        const BufRsrc rx_next = pool3_x_rsrc(p.ctx, (grp + gstride) * Gm::ROWS, n_load);        // (past the last group: an empty resource, zeros nobody reads)
#pragma unroll
        for (int dt = 0; dt < Gm::NTD; ++dt) {
          const float gv = gn;
          gn = gn2;
          if (dt + 2 < Gm::NTD) gn2 = gload(dt + 2);
          const u16x8 eh = *(const u16x8*)(eht + (dt & 1) * 1024 + l * 16);
          const u16 gh = f2bf(gv), gl = f2bf(gv - bf2f(gh));
          const u16x8 ad = u16x8{gh, gl, gh, 0, 0, 0, 0, 0};
          f32x4 acc[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32_bf16(ad, cat8(bd[m], Z4), f32x4{0.f, 0.f, 0.f, 0.f});
          product(dt, acc, false);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const f32x4 xv = mfma_16x16x32_bf16(eh, xr[dt >> 1][m], f32x4{0.f, 0.f, 0.f, 0.f});
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = xv[r] != 0.0f ? acc[m][r] * p.act_scale : 0.0f;
            *(u16x4*)(sc + m * 1024 + ((dt & 1) ? (cw0 ^ 32) : cw0)) = pack4(o);
          }
          if ((dt & 1) || dt == Gm::NTD - 1) {
            flush(dt & ~1, (dt & 1) != 0);
            pool3_load_x_ks(rx_next, dt >> 1, xr[dt >> 1]);     // this k-step's fragments have served their last tile: the next group's rows take their registers
          }
          NR_SCHED_BARRIER();
        }
      }
    }
    stamp(6);
    grp = nxt;
    stamp(7);
    ++it;
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
