// The data gradient of the 3-tap token convolution of the text encoders as a PERSISTENT ring GEMM for gfx950
// (autograd of nn.Conv2d(1, F, (3, D), padding=(1, 0)): src/model/NAML/news_encoder.py:15-17,27-36; src/model/LSTUR/news_encoder.py:24-28,62-69):
//
//   dX^T[d][i] = sum_{tap, f} Wd2[d][tap * KP + f] * dY[i + tap][f]      i = virtual row = seqpad row i + 1 (k_conv.h);  Wd2[d][tap KP + f] = W[f][2 - tap][d]
//
// Round 4's form of this product (gemm_ring_kernel MODE 2, k_gemm.h) gives every 320 x 256 output tile its own workgroup and rebuilds base, tap and
// column block of every copy from scratch: ~440 non-MFMA instructions per chunk and wave around 20 MFMAs.  With MFMAs, copies and stores switched
// off it still took 0.63 of its time (profiles/r06_convgemm_phases.txt): instruction ISSUE -- not the matrix pipe, the LDS or memory -- was the bound
// (two waves per SIMD, in lockstep through one barrier per chunk, have ~140 issue slots each under the chunk's MFMAs).  Here:
//   * ONE workgroup of 8 waves per CU walks tiles blockIdx.x, + gridDim.x, ...; the copy ring (four 36 KB chunk buffers, LDS-DMA three chunks ahead,
//     one counted vmcnt wait + one raw barrier per chunk) never drains: while the last chunks of a tile are multiplied the first chunks of the next
//     tile are already landing;
//   * the scalar side of a copy is a choice between two RUNNING pointers (filter bank / token rows of the fetch stream's tile, advanced once per
//     chunk: tap-inner order) + the slot address: everything lane- and block-dependent sits in one 32-bit lane offset per copy, computed once
//     (again for the partial last tile); the copy itself is a 3-instruction asm (glds16_lean);
//   * fragment reads are asynchronous (lds_read16_async) with counted lgkmcnt waits: before the MFMA group of long-side fragment o exactly
//     NO - 1 + NI reads may still be in flight (the same bookkeeping as the TN form of k_gemm.h);
//   * a finished tile leaves through the ring slot the stream has just left (free until the copies four chunks ahead are issued): wave-private rows,
//     8-byte accumulator quads in, 16-byte pieces out -- 8 tokens x 128 contiguous bytes per store instruction.  [Direct 8-byte stores from the
//     accumulator layout -- 32 rows x 16 bytes per instruction -- cost 0.2 ms per launch: store-issue bound.]  No asynchronous read is in flight across
//     that epilogue (its code is the compiler's, which may move the destination registers of a read it believes complete).
// Stores enter the same in-order memory counter as the copies; they are always YOUNGER than the copies a counted wait is for, so the waits stay
// correct (at worst they also wait for a few stores of the previous tile).
// A forward form of the same kernel (gather pass + GEMM with the bias / relu / dropout epilogue) was built and measured twice in round 6.  First
// version (bias fragments in registers): 1.29 ms against the LDS-tile kernel's 1.13 ms for the abstracts (the gather pass alone is 0.23 ms, the
// epilogue's registers spilled inside the loop), removed.  Second version (EPI below + conv_gather_kernel: bias from LDS, no spill on the per-chunk
// path): 1.47 against 1.20 ms, titles 0.65 against 0.57 ms (profiles/r06_ab_conv_fwd_gemm.txt) -- one counter hash per output quad (~35 VALU
// instructions) by all eight waves at once behind the tile's last barrier is ~0.45 ms that nothing overlaps.  Kept behind NR_CONV_FWD_GEMM=1
// (parity-tested: tests/kernel_checks_conv.py gemm=True), NOT the default.
#pragma once
#include "nr_common.h"

namespace nr {

struct ConvGemmGeom {
  static constexpr int BM = KP, BN = 256;                          // output channels x virtual token rows of a tile
  static constexpr int TM = 5, TN = 2;                             // 32 x 32 accumulator tiles of a wave: waves 2 (channels) x 4 (rows)
  static constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;       // a chunk = 32 contraction indices = 64 B per row
  static constexpr int BUF = A_BYTES + B_BYTES;                    // 36,864
  static constexpr int NB = 4, RING = NB * BUF;                    // 147,456
  static constexpr int ABLK = A_BYTES / 1024, NBLK = BUF / 1024;   // 20 + 16 blocks of 1 KB (16 rows) = one copy instruction each
  static constexpr int CP = 5;                                     // copies per wave and chunk (36 blocks over 8 waves; waves 4..7 repeat one)
  static constexpr int NCH = 3 * KP / 32;                          // 30 chunks per tile, TAP-INNER: chunk c = tap c % 3, columns 32 (c / 3)
  static constexpr int SMEM = RING;
  static constexpr int SMEM_EPI = RING + KP * 4;                   // + the bias row of the forward epilogue
  static_assert(SMEM_EPI <= 163840 && BM == 2 * TM * 32 && BN == 4 * TN * 32, "LDS; wave grid");
};

struct ConvGemmParams {
  const u16* A;          // [KP][3 * KP] row-major filter bank (nr_pack_conv_fwd2 / nr_pack_conv_dgrad)
  const u16* R;          // seqpad rows [n_rows + 2][KP]
  u16* C;                // [n_tok][KP] plain token layout
  int64_t n_rows;        // virtual rows = n_seq (S + 1) - 1
  int64_t n_tok;         // n_seq S
  int n_tiles;
  uint32_t S1;           // S + 1
  uint32_t s1_magic;     // floor(2^32 / (S + 1)) + 1
  int debug;             // profiling only (NR_CONVGEMM_DEBUG): 1 no copies of A, 2 no copies of the token rows, 4 no MFMAs, 8 no stores, 16 A is chunk-major
  // EPI (the FORWARD convolution over the masked token rows, nr_conv3_fwd_gemm): C = dropout2(relu(y + bias)), column D = 1.0, columns > D zero
  const float* bias;     // f32 [KP] (zero padded)
  int64_t tok_offset;    // added to the token index in the dropout counters (as in conv3_kernel)
  DropCfg dc;
};

struct CgYes { static constexpr bool v = true; };
struct CgNo { static constexpr bool v = false; };

// PLAIN: the same stream over an ordinary [n_rows][3 KP] row-major operand (row stride 3 KP instead of the overlapping KP of the seqpad rows, no
// separator rows, virtual row = output row): the input gradient of the title encoder's projections, dX = dqkv [Wq; Wk; Wv] (nr_dx_gemm, k_proj.h:
// A[n][which KP + f] = W_which[f][n]).  Chunks in straight column order.
// PAIRS (conv form only): chunk order (tap 0, block 2 j), (tap 0, block 2 j + 1), (tap 1, 2 j), (tap 1, 2 j + 1), (tap 2, 2 j), (tap 2, 2 j + 1): the
// two 64-byte halves of a 128-byte line of the rows are requested by consecutive chunks, the taps' re-reads stay two chunks apart (A/B:
// NR_CONVGEMM_PAIRS).
// EPI: the epilogue of the forward convolution (NAML news_encoder.py:27-32, LSTUR news_encoder.py:62-67) on the way through the staging rows:
// + bias (f32, from 1,280 bytes of LDS behind the ring: nothing stays in registers across the stream), relu, the counter-based dropout of
// conv3_kernel's output stage (site 2, quad = (tok_offset + token) D4 + column / 4: the same masks), column D = 1.0, columns > D zero.
template <bool DBG = false, bool PLAIN = false, bool PAIRS = false, bool EPI = false>
__global__ __launch_bounds__(512, 2) void conv_gemm_kernel(ConvGemmParams p) {
  using Gm = ConvGemmGeom;
  constexpr int TM = Gm::TM, TN = Gm::TN, CP = Gm::CP, NCH = Gm::NCH;
  constexpr int LDR = PLAIN ? 3 * KP : KP;                         // row stride of the token-row operand
  NR_SMEM_DECL(smem);
  const int l = lane_id(), w = wave_id(), h = l >> 5, li = l & 31;
  const int wr = w >> 2, wc = w & 3;
  const int grid = (int)gridDim.x;
  const int n_my = (int)blockIdx.x < p.n_tiles ? (p.n_tiles - 1 - (int)blockIdx.x) / grid + 1 : 0;      // tiles of this workgroup
  if (n_my == 0) return;
  const int T = n_my * NCH;                                        // chunks of this workgroup's stream
  const int dbg = DBG ? p.debug : 0;       // the production instantiation folds every switch away
  const uint32_t smem32 = lds_addr32(smem);
  float* const bias_s = (float*)(smem + Gm::RING);
  if (EPI) {
    p.dc = drop_resolve(p.dc);
    if (threadIdx.x < KP) bias_s[threadIdx.x] = p.bias[threadIdx.x];         // (published by the stream's barriers long before the first tile leaves)
  }

  // ---- copies.  Block b of a chunk buffer = rows 16 b .. + 15 of A (b < 20) / of the tile's token rows (b - 20); the lane's piece inside a block:
  // row l >> 2, physical 16-byte slot l & 3 <- logical k-slot (l & 3) ^ ((row >> 2) & 3) (16 b does not move the swizzle).  Wave w copies blocks
  // w, w + 8, w + 16, w + 24 and (waves 0..3) w + 32; waves 4..7 repeat block w + 24 (same bytes, same place).  Per copy ONE lane offset that holds
  // everything lane- and block-dependent (A: row of the filter bank; token rows: row of the tile, clamped to the live rows of a partial tile), so
  // that the scalar side of a copy is a choice between two running pointers.  [The first version of this kernel -- and k_gemm.h's -- rebuilt base,
  // tap and column block per copy: ~440 non-MFMA instructions per chunk and wave around 20 MFMAs; with MFMAs, copies and stores all switched off
  // it still took 0.63 of its time (profiles/r06_convgemm_phases.txt): instruction issue, not the matrix pipe, LDS or memory, was the bound.] ------------
  const int sl = (l & 3) ^ ((l >> 4) & 3);
  const int blk4 = w < 4 ? w + 32 : w + 24;
  const bool a2 = w < 4;                                           // copy 2 (block w + 16) is an A block for waves 0..3
  uint32_t voff[CP];
  voff[0] = (uint32_t)(((w * 16 + (l >> 2)) * (3 * KP) + sl * 8) * 2);
  voff[1] = (uint32_t)((((w + 8) * 16 + (l >> 2)) * (3 * KP) + sl * 8) * 2);
  auto set_row_offsets = [&](int lim) __attribute__((always_inline)) {      // lim: live virtual rows of the tile being fetched (rows past the end repeat the last one)
    auto rowoff = [&](int blk) -> uint32_t {
      int r = (blk - Gm::ABLK) * 16 + (l >> 2);
      r = r < lim ? r : lim - 1;
      return (uint32_t)((r * LDR + sl * 8) * 2);
    };
    voff[2] = a2 ? (uint32_t)((((w + 16) * 16 + (l >> 2)) * (3 * KP) + sl * 8) * 2) : rowoff(w + 16);
    voff[3] = rowoff(w + 24);
    voff[4] = rowoff(blk4);
  };
  auto tile_rows = [&](int tord) -> int64_t { return ((int64_t)blockIdx.x + (int64_t)tord * grid) * Gm::BN; };
  auto tile_lim = [&](int tord) -> int { const int64_t left = p.n_rows - tile_rows(tord); return left < Gm::BN ? (int)left : Gm::BN; };
  // the fetch stream: running pointers at (tile, chunk) = (ft, fc); tap-inner order: + KP columns twice, then back and + 32
  const u16* fa = p.A;
  const u16* fr = p.R + tile_rows(0) * LDR;
  int ft = 0, fc = 0, ftap = 0;
  uint32_t fdst = 0;                                               // byte offset of the ring slot being filled
  set_row_offsets(tile_lim(0));
  auto piece = [&](int i) __attribute__((always_inline)) {         // copy i of the chunk the fetch stream stands at
    const uint32_t dst = fdst + (uint32_t)((i < 4 ? w + 8 * i : blk4) * 1024);
    const bool isA = i < 2 || (i == 2 && a2);
    if (DBG && (dbg & 64)) NR_GLDS16_S(isA ? fa : fr, voff[i], smem + dst);      // (64: the save / restore form of the copy)
    else if (!DBG || (isA ? !(dbg & 1) : !(dbg & 2))) NR_GLDS16_L(isA ? fa : fr, voff[i], smem, smem32, dst);
  };
  auto advance = [&]() __attribute__((always_inline)) {            // the fetch stream moves on by one chunk
    fdst = fdst + Gm::BUF < (uint32_t)Gm::RING ? fdst + Gm::BUF : 0u;
    int d;
    if (PLAIN) {
      d = 32;                                                      // straight order: the other half of a 128-byte line is the next chunk's
    } else if (PAIRS) {
      d = (ftap & 1) == 0 ? 32 : (ftap == 5 ? 32 - 2 * KP : KP - 32);
      ftap = ftap == 5 ? 0 : ftap + 1;
    } else {
      d = ftap == 2 ? 32 - 2 * KP : KP;
      ftap = ftap == 2 ? 0 : ftap + 1;
    }
    fa += d; fr += d;
    if (++fc == NCH) {
      fc = 0; ++ft;
      fa = p.A;
      if (ft < n_my) {
        fr = p.R + tile_rows(ft) * LDR;
        const int lim = tile_lim(ft);
        if (lim < Gm::BN) set_row_offsets(lim);                    // (only the last tile of the problem, i.e. the last of this stream)
      }
    }
  };
  auto fetch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CP; ++i) piece(i);
    advance();
  };

  f32x16 acc[TM][TN];

  // ---- fragments: row 32 tile + li of the operand's chunk rows, k-slot (2 ks + h) ^ ((row >> 2) & 3) (32 tile does not move the swizzle) -----------
  const int sw = (li >> 2) & 3;
  const int f0 = li * 64 + ((h ^ sw) << 4);                        // k-step 0; k-step 1: slot ^ 2 = byte offset ^ 32
  const unsigned char* const fragA = smem + wr * TM * 2048;        // + buffer, + lane offset, + 2048 a
  const unsigned char* const fragB = smem + Gm::A_BYTES + wc * TN * 2048;
  u16x8 lg[TM], sh[2][TN];
  auto read_short = [&](auto par_tag, int rb, int ks) __attribute__((always_inline)) {      // the TN fragments of the token rows for k-step ks of buffer rb into set P
    constexpr int P_ = decltype(par_tag)::v ? 1 : 0;
    const unsigned char* b = fragB + rb * Gm::BUF + (ks ? (f0 ^ 32) : f0);
    sh[P_][0] = lds_read16_async<0>(b);
    sh[P_][1] = lds_read16_async<2048>(b);
  };
  // One k-step of MFMAs on the fragments in registers (short side: set P).  NEXT: on the way the fragments of k-step 1 - P of buffer rb replace
  // them.  copy: the CP copies of the fetch stream's chunk are issued between the MFMAs.
  auto multiply = [&](auto next_tag, auto par_tag, auto copy_tag, int rb) __attribute__((always_inline)) {
    constexpr bool next = decltype(next_tag)::v, copy = decltype(copy_tag)::v;
    constexpr int P = decltype(par_tag)::v ? 1 : 0;
    using Other = typename std::conditional<P == 0, CgYes, CgNo>::type;
    const unsigned char* a = fragA + rb * Gm::BUF + (P == 0 ? (f0 ^ 32) : f0);
    if (next) {
      read_short(Other{}, rb, 1 - P);
    } else {
      NR_SCHED_BARRIER();
      NR_WAIT_LGKMCNT(0);
    }
    NR_SCHED_BARRIER();
    static_for<TM>([&](auto ot) __attribute__((always_inline)) {
      constexpr int o = decltype(ot)::value;
      if (next) { NR_SCHED_BARRIER(); NR_WAIT_LGKMCNT(TM - 1 + TN); NR_SCHED_BARRIER(); }     // (pinned: the operands come from asm the scheduler cannot see through)
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if (!DBG || !(dbg & 4)) acc[o][i] = mfma_32x32x16_bf16(lg[o], sh[P][i], acc[o][i]);
        if (o * TN + i < CP && copy) piece(o * TN + i);
      }
      if (next) lg[o] = lds_read16_async<o * 2048>(a);
      NR_SCHED_BARRIER();
    });
    if (copy) advance();
  };
  auto arrive = [&](bool more) __attribute__((always_inline)) {    // the next stream chunk complete for everybody; everybody has read the one before it
    if (more) NR_WAIT_VMCNT(2 * CP);                               // (the copies of the two chunks after it may stay in flight)
    else NR_WAIT_VMCNT(0);
    NR_WAIT_LGKMCNT(0);
    NR_BARRIER_RAW();
  };

  // ---- results of a tile through the ring slot the stream has just left (free until the copies of the chunk four ahead are issued): wave-private
  // rows of 144 B; accumulator quads (4 consecutive channels of token li) are written 8 bytes at a time, two channel tiles = 128 B per token, and
  // leave as 16-byte pieces, 8 tokens x 128 contiguous bytes per store instruction (the fifth tile: 16 tokens x 64 bytes) --------------------------------
  const BufRsrc r_c = make_buf(p.C, (DBG && (dbg & 8)) ? 0u : (uint32_t)(p.n_tok * (KP * 2)));
  constexpr uint32_t OOR = 0x80000000u;
  constexpr int SROW = 144;
  auto store_tile = [&](int tord, int slot) __attribute__((always_inline)) {
    const int64_t n0 = tile_rows(tord);
    unsigned char* stg = smem + slot * Gm::BUF + w * (32 * SROW);
    auto token_of = [&](int tl, int jt, uint32_t& tok) -> bool {   // token row of local row tl of the wave's tile jt; false: separator / past the end
      const int64_t vr = n0 + (wc * TN + jt) * 32 + tl;            // virtual row = seqpad row vr + 1
      if (PLAIN) { tok = (uint32_t)vr; return vr < p.n_rows; }
      const uint32_t sp = (uint32_t)(vr + 1);
      uint32_t sq = mulhi_u32(sp, p.s1_magic);                     // floor(sp / (S + 1)) or one more
      sq -= (sq * p.S1 > sp) ? 1u : 0u;
      tok = sp - sq - 1u;
      return vr < p.n_rows && sp != sq * p.S1;
    };
#pragma unroll
    for (int jt = 0; jt < TN; ++jt) {
      uint32_t tokq = 0;                                             // EPI: the token this lane's accumulator quads belong to
      if (EPI) token_of(li, jt, tokq);
      auto put = [&](int a, int pos) __attribute__((always_inline)) {       // channel tile a of the wave -> bytes 64 pos .. + 63 of the staging rows
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 y = f32x4{acc[a][jt][4 * q], acc[a][jt][4 * q + 1], acc[a][jt][4 * q + 2], acc[a][jt][4 * q + 3]};
          if (EPI) {
            const int col = (wr * TM + a) * 32 + q * 8 + h * 4;     // 4 consecutive filters of token tokq
            y = y + *(const f32x4*)(bias_s + col);
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.0f);
            if (col < D) {
              if (p.dc.enabled) y = y * drop_mul4(p.dc, 2u, (uint64_t)(p.tok_offset + (int64_t)tokq) * D4 + (uint32_t)(col >> 2));
            } else {
              y = col == D ? f32x4{1.0f, 0.0f, 0.0f, 0.0f} : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
          }
          *(u16x4*)(stg + li * SROW + pos * 64 + q * 16 + h * 8) = pack4(y);
        }
      };
#pragma unroll
      for (int ap = 0; ap + 1 < TM; ap += 2) {
        put(ap, 0); put(ap + 1, 1);
        wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int tl = k * 8 + (l >> 3), pc = l & 7;
          uint32_t tok;
          const bool live = token_of(tl, jt, tok);
          const u16x8 v = *(const u16x8*)(stg + tl * SROW + pc * 16);
          buf_store16(r_c, live ? tok * (uint32_t)(KP * 2) + (uint32_t)(pc * 16 + ((wr * TM + ap) * 32) * 2) : OOR, v);      // (no scalar offset: see below)
        }
        wave_barrier();
      }
      put(TM - 1, 0);
      wave_barrier();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int tl = k * 16 + (l >> 2), pc = l & 3;
        uint32_t tok;
        const bool live = token_of(tl, jt, tok);
        const u16x8 v = *(const u16x8*)(stg + tl * SROW + pc * 16);
        // (the wave-uniform channel offset sits in the LANE offset, not in the scalar-offset field: see buf_store16 in nr_prims.h)
        buf_store16(r_c, live ? tok * (uint32_t)(KP * 2) + (uint32_t)(pc * 16 + ((wr * TM + TM - 1) * 32) * 2) : OOR, v);
      }
      wave_barrier();
    }
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.0f;
  };

  // ---- the stream --------------------------------------------------------------------------------------------------------------------------------
  zero_acc();
  fetch(); fetch(); fetch();
  arrive(true);
  fetch();                                                         // (T >= 30: the fourth chunk exists)
  int t = 0, c = 0, rb = 0;                                        // tile number and chunk of stream chunk g (k-step 0 in registers), its ring slot
  // one stream chunk: MFMAs of (g, k-step 0) while (g, 1) is read [+ the copies of chunk g + 3, whose slot the last arrive freed]; the next chunk
  // arrives; MFMAs of (g, 1) while (g + 1, 0) is read; a complete tile leaves through the slot of chunk g, which nobody reads any more
  auto first_fragments = [&](int slot) __attribute__((always_inline)) {       // k-step 0 of the chunk in `slot` -> registers (set 0 of the short side), waited for
    const unsigned char* a = fragA + slot * Gm::BUF + f0;
    static_for<TM>([&](auto ot) __attribute__((always_inline)) { constexpr int o = decltype(ot)::value; lg[o] = lds_read16_async<o * 2048>(a); });
    read_short(CgNo{}, slot, 0);
    NR_SCHED_BARRIER(); NR_WAIT_LGKMCNT(0); NR_SCHED_BARRIER();
  };
  auto chunk = [&](auto copy_tag, bool more) __attribute__((always_inline)) {
    multiply(CgYes{}, CgNo{}, copy_tag, rb);
    arrive(more);
    const int rb_done = rb;
    rb = (rb + 1) & (Gm::NB - 1);
    multiply(CgYes{}, CgYes{}, CgNo{}, rb);
    if (++c == NCH) {
      // the tile is complete.  The fragments of the next chunk must have LANDED before the epilogue: its code is the compiler's, which may move
      // or spill those registers -- and believes they were written when the asynchronous reads were issued
      NR_SCHED_BARRIER(); NR_WAIT_LGKMCNT(0); NR_SCHED_BARRIER();
      store_tile(t, rb_done);
      NR_WAIT_LGKMCNT(0);
      NR_BARRIER_RAW();                                            // (the next copies into that slot come from OTHER waves' instruction streams)
      zero_acc();
      c = 0;
      ++t;
    }
  };
  first_fragments(0);
  int g = 0;
  chunk(CgNo{}, true);                                             // (chunk 3 of the stream is already under way)
  for (g = 1; g + 3 < T; ++g) chunk(CgYes{}, true);
  for (; g + 1 < T; ++g) chunk(CgNo{}, false);                     // the last chunks of the stream: nothing left to copy
  multiply(CgYes{}, CgNo{}, CgNo{}, rb);
  multiply(CgNo{}, CgYes{}, CgNo{}, 0);
  NR_WAIT_LGKMCNT(0);
  NR_BARRIER_RAW();                                                // everybody has read the last chunk: its slot is the staging area
  store_tile(n_my - 1, rb);
}

// ---- the input stage of the forward convolution as its own pass (the GEMM form's R operand IS the x_save buffer the weight-gradient GEMMs read):
// x_save seqpad rows [n_seq (S + 1) + 1][KP] = bf16(F.dropout(table[ids])) (NAML news_encoder.py:23-25 / LSTUR :58-60), column D = 1.0 in token
// rows, zero separators; positions s >= valid are zero vectors.  Same counters, same arithmetic, same bits as conv3_kernel's staging phase.
struct ConvGatherParams {
  const int64_t* ids;
  const float* table;
  int64_t num_rows;
  u16* x_save;
  int64_t n_seq;
  int S, valid;
  int64_t tok_offset;
  DropCfg dc;
};
__global__ __launch_bounds__(256) void conv_gather_kernel(ConvGatherParams p) {
  p.dc = drop_resolve(p.dc);
  constexpr int QR = KP / 4;                                       // 80 quads per row
  const int64_t rows_total = p.n_seq * (p.S + 1) + 1;
  const int64_t total = rows_total * QR;
  const uint32_t S1 = (uint32_t)(p.S + 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t gr = i / QR;
    const int c = (int)(i - gr * QR);
    const int64_t seq = gr / S1;
    const int s = (int)(gr - seq * S1) - 1;
    u16x4 o = u16x4{0, 0, 0, 0};
    if (s >= 0 && seq < p.n_seq) {
      if (c < D4) {
        if (p.valid <= 0 || s < p.valid) {
          const int64_t tok = seq * p.S + s;
          int64_t id = p.ids[tok];
          id = id < 0 ? 0 : (id >= p.num_rows ? p.num_rows - 1 : id);
          f32x4 x = *(const f32x4*)(p.table + ((size_t)id * D4 + c) * 4);
          if (p.dc.enabled) x = x * drop_mul4(p.dc, 1u, (uint64_t)(p.tok_offset + tok) * D4 + c);
          o = pack4(x);
        }
      } else if (c == D4) {
        o[0] = 0x3F80;
      }
    }
    *(u16x4*)(p.x_save + gr * KP + c * 4) = o;
  }
}

// A operand of the forward GEMM form: Wf2[f][tap * KP + d] = W[f][tap][d] (Conv2d weight f32 [F][1][3][D]), bf16 row-major [KP][3 * KP], zero padded
__global__ __launch_bounds__(256) void pack_conv_fwd2_kernel(const float* __restrict__ W, int F_, int D_, u16* __restrict__ Wf2) {
  const int total = KP * 3 * KP;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int f = i / (3 * KP), rem = i - f * 3 * KP, tap = rem / KP, d = rem - tap * KP;
    Wf2[i] = f2bf((f < F_ && d < D_) ? W[((size_t)f * 3 + tap) * D_ + d] : 0.0f);
  }
}

}  // namespace nr
