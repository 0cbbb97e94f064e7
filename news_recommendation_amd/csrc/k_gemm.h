// General bf16 GEMMs of the backward / recurrent paths for gfx950, fp32 accumulation and fp32 results -- the products the engine used to hand
// to hipBLASLt (LSTUR: src/model/LSTUR/user_encoder.py:11-14,30-45 = nn.GRU's x W_ih^T, and autograd's dX = dGi W_ih, dW_ih = dGi^T X,
// dW_hh = dGh^T H; NAML / LSTUR: autograd of the Conv2d(1, F, (3, D)) weight, src/model/NAML/news_encoder.py:27-28; NRMS: autograd of the
// nn.Linear weights, multihead_self.py:53-55, additive.py:35).
//
// One kernel skeleton, the ring of dx_gemm_ring_kernel (k_proj.h): one workgroup of 8 waves per CU, FOUR chunk buffers in LDS filled by
// LDS-DMA three chunks ahead, per chunk one counted s_waitcnt + one raw s_barrier, the k-steps of neighbouring chunks interleaved so that
// every block of MFMAs hides the LDS latency of the next fragment set.  Two operand forms:
//   NT   C[m][n] = sum_k A[m][k] B[n][k]        both operands K-contiguous (activations x packed weights): plain 16-byte fragment reads
//   TN   C[p][m][n] = sum_{tok in part p} G[tok][m] X[tok + n / tapw][n % tapw]    both operands token-major (weight gradients, split K over
//        token partitions): fragments by the transposing LDS read ds_read_b64_tr_b16; with taps = 3 the X operand is the VIRTUAL row
//        [x[tok], x[tok + 1], x[tok + 2]] of a seqpad buffer (k_conv.h), so that the three tap gradients of a convolution are ONE GEMM with
//        N = 3 * 320 whose G tile is fetched once for all taps.
//   NT3  (MODE 2) the data gradient of the 3-tap convolution (k_conv.h) as a GEMM: C^T[d][row] = sum_{tap, f} Wd2[d][tap * 320 + f] dY[row + tap][f] --
//        A = the flipped / transposed filter bank, row-major [320][960]; B = the VIRTUAL row [dy[i], dy[i + 1], dy[i + 2]] of the seqpad gradient
//        buffer (chunk c of the contraction reads columns 32 (c / 3) of row i + c % 3: tap-inner); the result leaves as bf16 rows in plain token layout
//        (separator rows dropped), staged through LDS so that every store is a 16-byte piece of a contiguous 640-byte row.
// Three tile shapes (8 waves as WR x WC, a wave owns TM x TN 32 x 32 accumulator tiles): 256 x 256 (4 x 2 waves of 2 x 4 tiles), 320 x 256
// (2 x 4 waves of 5 x 2 tiles: the conv tap gradients have M = 320 rows) and 256 x 320 (4 x 2 waves of 2 x 5 tiles: N = 320 columns).
#pragma once
#include "nr_common.h"

namespace nr {

struct GemmParams {
  const u16* A;          // NT: [M][lda]   TN: G [n_tok][lda]
  const u16* B;          // NT: [N][ldb]   TN: X [n_tok (+ taps - 1)][ldb]
  float* C;              // NT: [M][ldc]   TN: [P][M][ldc]
  int64_t lda, ldb, ldc;
  int64_t M;             // NT: rows of A;  TN: output rows = columns of G used (<= lda)
  int N;                 // output columns (TN: taps * tapw columns of the virtual X row)
  int K;                 // NT: contraction length (multiple of 32)
  // TN only
  const u16* zeros;      // >= 16 zero bytes: what lanes beyond a partition's tokens / an operand's columns copy
  int64_t n_tok;
  int64_t tok_per_part;  // multiple of 32
  int P;                 // token partitions (multiple of 8)
  int tapw;              // columns per tap of the virtual X row (taps = 1: >= N)
  int tiles_m, tiles_n;
  // MODE 2 only (data gradient of the 3-tap convolution): bf16 result rows in plain token layout
  u16* Cb;               // [n_seq * S][KP]
  int S;                 // tokens per sequence: virtual row i is seqpad row i + 1; rows with (i + 1) % (S + 1) == 0 are separators (not stored)
};

template <int MODE, int WR, int TM, int TN_>
struct GemmGeom {
  static constexpr int WC = 8 / WR;
  static constexpr int BM = WR * TM * 32, BN = WC * TN_ * 32;
  static constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;       // a chunk = 32 contraction indices: 64 B per row (NT) / 32 token rows (TN)
  static constexpr int BUF = A_BYTES + B_BYTES;
  static constexpr int NB = 4;
  static constexpr int SMEM = NB * BUF;                            // 131,072 (256 x 256) / 147,456 (320 x 256)
  static constexpr int NBLK = BUF / 1024;                          // 1 KB blocks = one LDS-DMA instruction each
  static constexpr int ABLK = A_BYTES / 1024;
  static constexpr int CP = (NBLK + 7) / 8;                        // copy instructions per wave and chunk
  static constexpr int SLA = BM / 8, SLB = BN / 8;                 // TN: 16-byte slots per token row
  static_assert(SMEM <= 163840 && CP * (NB - 1) < 64 && NBLK >= 8, "ring fits the LDS; vmcnt range");
};

// TN swizzle of the 16-byte slots of token row r: the rows a transposing read touches must sit in different banks.  Row length a multiple of
// 256 B (SL % 16 == 0): four consecutive rows start at the same bank -> shift by 0 / 64 / 128 / 192 B; an odd multiple of 128 B
// (SL % 16 == 8): rows r, r + 2 collide -> shift the second pair by 64 B.  (XOR keeps a slot inside its aligned group of 16 / 8.)
template <int SL>
__device__ __forceinline__ int tn_swz(int r) { return (SL % 16 == 0) ? 4 * (r & 3) : 4 * ((r >> 1) & 1); }

struct GemmYes { static constexpr bool v = true; };
struct GemmNo { static constexpr bool v = false; };
template <bool B> struct GemmTag { using type = GemmYes; };
template <> struct GemmTag<false> { using type = GemmNo; };

template <int MODE, int WR, int TM, int TN_>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(GemmParams p) {
  using Gm = GemmGeom<MODE, WR, TM, TN_>;
  constexpr bool TN = MODE == 1;
  NR_SMEM_DECL(smem);
  const int l = lane_id(), w = wave_id(), h = l >> 5, li = l & 31;
  const int wr = w / Gm::WC, wc = w % Gm::WC;
  // block id -> (xcd, tile, partition): ids that differ by a multiple of 8 share an XCD (round-robin dispatch).  NT: the column tiles of a
  // row tile get consecutive such ids (the A rows come from HBM once, the packed weights live in every L2); TN: the tiles of a partition.
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  int tm, tn, part = 0;
  if (TN) {
    const int nt = p.tiles_m * p.tiles_n, t = jb % nt;
    part = (jb / nt) * 8 + xcd;
    tm = t % p.tiles_m; tn = t / p.tiles_m;
  } else if (MODE == 2) {
    tm = 0; tn = blockIdx.x;                                      // one row tile (the 320 filters): every block is a tile of 256 token rows
  } else {
    tm = (jb / p.tiles_n) * 8 + xcd; tn = jb % p.tiles_n;
    if (tm >= p.tiles_m) return;
  }
  const int64_t m0 = (int64_t)tm * Gm::BM;
  const int n0 = tn * Gm::BN;
  int64_t t_begin = 0;
  int ntok = 0, nchunk = p.K / 32;
  if (TN) {
    t_begin = (int64_t)part * p.tok_per_part;
    int64_t t_end = t_begin + p.tok_per_part;
    t_end = t_end < p.n_tok ? t_end : p.n_tok;
    ntok = t_end > t_begin ? (int)(t_end - t_begin) : 0;
    nchunk = (ntok + 31) / 32;
  }

  // ---- the CP copies of this wave: block blk = w + 8 i of the chunk (A blocks first).  Wave-uniform per copy: the block, its operand and that
  // operand's base pointer at the tile / partition; per lane only a 32-bit element offset from that base (-1: the lane copies zeros) and, for
  // TN, the token row of its piece (rows beyond the partition copy zeros).  [64-bit per-lane pointers and strides: 30 registers, spills] ----
  const u16* const baseA = TN ? p.A + t_begin * p.lda : p.A + m0 * p.lda;
  const u16* const baseB = TN ? p.B + t_begin * p.ldb : p.B + (int64_t)n0 * p.ldb;
  auto blk_of = [&](int i) { const int b_ = w + 8 * i; return b_ < Gm::NBLK ? b_ : b_ - 8; };     // surplus copies repeat a block (same bytes, same place)
  int off[Gm::CP], trow[Gm::CP];
#pragma unroll
  for (int i = 0; i < Gm::CP; ++i) {
    const int blk = blk_of(i);
    const bool isA = blk < Gm::ABLK;
    const int g = (isA ? blk : blk - Gm::ABLK) * 64 + l;          // 16-byte piece index inside the operand's chunk
    if (!TN) {
      const int r = g >> 2, s = (g & 3) ^ ((r >> 2) & 3);         // row of the tile, LOGICAL k-slot of the piece that lands in physical slot g & 3
      const int64_t lim = isA ? p.M - m0 : (int64_t)p.N - n0;     // rows past the end repeat the last row (never stored)
      const int rr = r < lim ? r : (int)lim - 1;
      off[i] = rr * (int)(isA ? p.lda : p.ldb) + s * 8;
      trow[i] = 0;
    } else {
      const int SL = isA ? Gm::SLA : Gm::SLB;
      const int r = g / SL, ps = g - r * SL;
      const int s = ps ^ (isA ? tn_swz<Gm::SLA>(r) : tn_swz<Gm::SLB>(r));
      trow[i] = r;
      if (isA) {
        const int64_t col = m0 + s * 8;
        off[i] = col < p.lda ? r * (int)p.lda + (int)col : -1;
      } else {
        const int col = n0 + s * 8;                               // column of the virtual X row: tap = col / tapw
        const int tap = col / p.tapw, c = col - tap * p.tapw;
        off[i] = (col < p.N && c < p.ldb) ? (r + tap) * (int)p.ldb + c : -1;
      }
    }
  }
  auto piece = [&](int c, int i) {
    unsigned char* buf = smem + (c & (Gm::NB - 1)) * Gm::BUF;
    const int blk = blk_of(i);
    const bool isA = blk < Gm::ABLK;
    const u16* s_;
    if (TN) {
      const u16* cb = isA ? baseA + (int64_t)c * 32 * p.lda : baseB + (int64_t)c * 32 * p.ldb;      // wave-uniform
      s_ = (off[i] >= 0 && c * 32 + trow[i] < ntok) ? cb + off[i] : p.zeros;
    } else if (MODE == 2) {
      // TAP-INNER order of the contraction: chunk c = tap c % 3, columns 32 (c / 3): the three chunks that read the 64-byte pieces of rows
      // i, i + 1, i + 2 at one column block follow each other, so two of the three fetches of every piece hit the L2 line the first one brought
      // in (tap-outer -- chunk c = tap c / 10 -- re-read each tile's rows 10 and 20 chunks later, through a 4 MB L2 shared by 64 such tiles:
      // 3.82 GB fetched for 1.22 GB of rows, profiles/r04_pmc_step_traffic_NAML.txt)
      const int tap = c % 3, cb = c / 3;
      s_ = isA ? baseA + tap * KP + cb * 32 + off[i] : baseB + (int64_t)tap * p.ldb + cb * 32 + off[i];
    } else {
      s_ = (isA ? baseA : baseB) + c * 32 + off[i];
    }
    NR_GLDS16(s_, buf + blk * 1024);
  };
  auto fetch = [&](int c) {
#pragma unroll
    for (int i = 0; i < Gm::CP; ++i) piece(c, i);
  };

  f32x16 acc[TM][TN_];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.0f;

  // Fragment sets of one k-step (16 contraction indices): TM fragments of A, TN_ of B.  The LONGER operand side lives in ONE set that is
  // refilled in place -- while the MFMAs of a group (one fragment of the long side x all fragments of the short side) run, the fragment they
  // just consumed is re-read for the NEXT k-step into the same registers; the short side alternates between two sets (parity P of the
  // k-step).  (Two whole sets, as dx_gemm_ring_kernel keeps them, cost 2 x 28 registers beside the 160 accumulator registers of the
  // 320 x 256 tile: 98 spills, and spill reloads travel through vmcnt and drain the copy ring.)
  constexpr bool AMAJ = TM >= TN_;
  constexpr int NO = AMAJ ? TM : TN_, NI = AMAJ ? TN_ : TM;       // long / short side
  u16x8 lg[NO], sh[2][NI];
  // TN: geometry of the transposing reads (see lds_tr16_b64): the lane supplies the piece (row k0 + prow, columns n0 + pcol .. + 3) of the
  // 4 x 16 block its 16-lane group transposes.  The swizzle term of a row depends on its low two bits only, i.e. on prow (k-step, half and
  // piece pair move the row by multiples of 4): per fragment ONE lane-dependent byte offset, everything else is the instruction's offset field.
  const int prow = (l & 15) >> 2, pcol = 16 * ((l >> 4) & 1) + 4 * (l & 3);
  int tbase[TM + TN_];                                            // TN: byte offset of fragment idx (A: 0 .. TM - 1, B: TM ..) inside a chunk buffer, k-step 0, piece pair 0
  if (TN) {
#pragma unroll
    for (int i = 0; i < TM + TN_; ++i) {
      const bool isA = i < TM;
      const int tile = isA ? wr * TM + i : wc * TN_ + (i - TM);
      const int SL = isA ? Gm::SLA : Gm::SLB, row = 8 * h + prow, col = 32 * tile + pcol;
      const int swz = isA ? tn_swz<Gm::SLA>(row) : tn_swz<Gm::SLB>(row);
      tbase[i] = (isA ? 0 : Gm::A_BYTES) + (row * SL + ((col >> 3) ^ swz)) * 16 + (col & 7) * 2;
    }
  }
  // fragment `idx` of operand side A / B (tag) for k-step 0 / 1 (tag) of chunk c.  TN: asynchronous transposing reads, waited for by hand
  auto read_op = [&](auto a_tag, auto ks_tag, int c, int idx) -> u16x8 {
    constexpr bool isA = decltype(a_tag)::v;
    constexpr int ks = decltype(ks_tag)::v ? 1 : 0;
    const unsigned char* buf = smem + (c & (Gm::NB - 1)) * Gm::BUF;
    if (!TN) {
      const int row = (isA ? wr * TM + idx : wc * TN_ + idx) * 32 + li;
      return *(const u16x8*)(buf + (isA ? 0 : Gm::A_BYTES) + row * 64 + (((ks * 2 + h) ^ ((row >> 2) & 3)) * 16));
    }
    constexpr int SL = isA ? Gm::SLA : Gm::SLB;
    const u16* pa = (const u16*)(buf + tbase[isA ? idx : TM + idx]);
    return cat8(lds_tr16_b64_async<(16 * ks) * SL * 16>(pa), lds_tr16_b64_async<(16 * ks + 4) * SL * 16>(pa));
  };
  static_assert(TM * TN_ >= Gm::CP, "one copy per MFMA at most");
  // One k-step of MFMAs on the fragments in registers (short side: set P).  NEXT: on the way the fragments of k-step (nc, nks) replace them
  // (short side into set 1 - P).  cnext >= 0: the CP copies of chunk cnext are issued between the MFMAs, not as a burst behind the barrier.
  // TN waits (reads return in issue order, 2 per fragment): before group o the refill of lg[o] from the PREVIOUS call must have landed;
  // younger than it are the previous call's refills o + 1 .. NO - 1, this call's short-side reads and its refills 0 .. o - 1:
  // 2 (NO - 1 + NI) reads may stay in flight -- the same count for every group.
  using LongTag = typename GemmTag<AMAJ>::type;                   // operand side of the long / short fragment set
  using ShortTag = typename GemmTag<!AMAJ>::type;
  // par_tag: parity of the k-step in registers (= which short-side set holds it); the k-step being fetched is the other one: 1 - P
  auto multiply = [&](auto next_tag, auto par_tag, int nc, int cnext) {
    constexpr bool next = decltype(next_tag)::v;                  // compile time: no control flow around the accumulators
    constexpr int P = decltype(par_tag)::v ? 1 : 0;
    using NextKs = typename GemmTag<P == 0>::type;
    if (next) {
#pragma unroll
      for (int i = 0; i < NI; ++i) sh[1 - P][i] = read_op(ShortTag{}, NextKs{}, nc, i);
    } else if (TN) {
      NR_SCHED_BARRIER();
      NR_WAIT_LGKMCNT(0);
    }
    NR_SCHED_BARRIER();
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      if (TN && next) { NR_SCHED_BARRIER(); NR_WAIT_LGKMCNT(2 * (NO - 1 + NI)); NR_SCHED_BARRIER(); }      // (pinned: the scheduler may otherwise hoist an MFMA above the wait -- its operands come from asm it cannot see through)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int a = AMAJ ? o : i, j = AMAJ ? i : o;
        acc[a][j] = mfma_32x32x16_bf16(AMAJ ? lg[o] : sh[P][i], AMAJ ? sh[P][i] : lg[o], acc[a][j]);     // C[m][n]: lane = column n, rows 8 q + 4 h + e
        if (o * NI + i < Gm::CP && cnext >= 0 && cnext < nchunk) piece(cnext, o * NI + i);           // (compile-time copy index)
      }
      if (next) lg[o] = read_op(LongTag{}, NextKs{}, nc, o);
      NR_SCHED_BARRIER();
    }
  };
  auto arrive = [&](int c) {
    // this wave's copies of chunk c have landed once at most the copies of the (up to two) chunks after it are outstanding
    if (c + 2 < nchunk) NR_WAIT_VMCNT(2 * Gm::CP);
    else if (c + 1 < nchunk) NR_WAIT_VMCNT(Gm::CP);
    else NR_WAIT_VMCNT(0);
    NR_WAIT_LGKMCNT(0);                                           // this wave's reads of chunk c - 1 have returned
    NR_BARRIER_RAW();                                             // chunk c complete for everybody; everybody has read chunk c - 1: its slot is free
  };
  if (nchunk > 0) {
    for (int c = 0; c < Gm::NB - 1; ++c)
      if (c < nchunk) fetch(c);
    arrive(0);
    if (Gm::NB - 1 < nchunk) fetch(Gm::NB - 1);
#pragma unroll
    for (int o = 0; o < NO; ++o) lg[o] = read_op(LongTag{}, GemmNo{}, 0, o);
#pragma unroll
    for (int i = 0; i < NI; ++i) sh[0][i] = read_op(ShortTag{}, GemmNo{}, 0, i);
    if (TN) { NR_SCHED_BARRIER(); NR_WAIT_LGKMCNT(0); NR_SCHED_BARRIER(); }      // (the counted waits in multiply assume refill order)
    int c = 0;
    for (; c + 1 < nchunk; ++c) {
      // registers hold k-step 0 of chunk c; the slot of chunk c - 1 is free (arrive(c) has passed): chunk c + 3 goes there
      multiply(GemmYes{}, GemmNo{}, c, c >= 1 ? c + Gm::NB - 1 : -1);          // MFMAs of (c, 0), fetching (c, 1)
      arrive(c + 1);
      multiply(GemmYes{}, GemmYes{}, c + 1, -1);                                // MFMAs of (c, 1), fetching (c + 1, 0)
    }
    multiply(GemmYes{}, GemmNo{}, c, -1);                         // the last chunk: nothing left to copy
    multiply(GemmNo{}, GemmYes{}, 0, -1);
  }
  if constexpr (MODE == 2) {
    // ---- bf16 rows in plain token layout: TN_ passes of 128 virtual rows (token tile jt of the four wave columns) through LDS.  The lane holds
    // 4 consecutive filters-in (d) of ONE row: 8-byte staging writes; the rows then leave as 16-byte pieces of contiguous 640-byte runs -------
    constexpr int OROW = 656;
    static_assert(Gm::WC * 32 * OROW <= Gm::SMEM && Gm::BM == KP, "staging fits the ring; one row tile = all KP columns");
#pragma unroll
    for (int jt = 0; jt < TN_; ++jt) {
      __syncthreads();                                            // the ring (jt = 0) / the previous pass has been read
      unsigned char* row = smem + (wc * 32 + li) * OROW;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(u16x4*)(row + ((wr * TM + a) * 32 + 8 * q + 4 * h) * 2) =
              pack4(f32x4{acc[a][jt][4 * q], acc[a][jt][4 * q + 1], acc[a][jt][4 * q + 2], acc[a][jt][4 * q + 3]});
      __syncthreads();
      constexpr int PCS = KP / 8;                                 // 40 sixteen-byte pieces per row
#pragma unroll
      for (int i = 0; i < Gm::WC * 32 * PCS / 512; ++i) {
        const int idx = threadIdx.x + 512 * i;
        const int r = idx / PCS, pc = idx - r * PCS;
        const int64_t vr = n0 + (r >> 5) * (TN_ * 32) + jt * 32 + (r & 31);      // virtual row = seqpad row vr + 1
        const int64_t sp = vr + 1, sq = sp / (p.S + 1);
        if (vr < p.N && sp != sq * (p.S + 1)) *(u16x8*)(p.Cb + (sp - sq - 1) * KP + pc * 8) = *(const u16x8*)(smem + r * OROW + pc * 16);
      }
    }
    return;
  }
  // ---- results: dword stores, the 32 lanes of a half-wave write 128 contiguous bytes of one output row ---------------------------------
  float* cbase = p.C + (TN ? (size_t)part * p.M * p.ldc : 0);
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int j = 0; j < TN_; ++j) {
      const int n = n0 + (wc * TN_ + j) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + (wr * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M && n < p.N) cbase[m * p.ldc + n] = acc[a][j][r];
      }
    }
}

// dst[c][r] = src[r][c] for bf16 matrices (weights re-packed once per optimiser step so that the NN products run in the NT kernel):
// src [R][lds], dst [C][ldd]; columns r >= R of dst are left untouched (the caller zero-fills the padding once)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const u16* __restrict__ src, int R, int C, int64_t lds, u16* __restrict__ dst, int64_t ldd) {
  NR_SMEM_DECL(smem);                                            // u16 tile[32][33]
  u16* tile = (u16*)smem;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;           // bx: source column block, by: source row block
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    tile[i * 33 + tx] = (by + i < R && bx + tx < C) ? src[(int64_t)(by + i) * lds + bx + tx] : (u16)0;
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (bx + i < C && by + tx < R) dst[(int64_t)(bx + i) * ldd + by + tx] = tile[tx * 33 + i];
}

// A operand of the NT3 form: Wd2[d][tap * KP + f] = W[f][2 - tap][d] (Conv2d weight f32 [F][1][3][D]; flipped taps, transposed filters), bf16
// row-major [KP][3 * KP], zero padded
__global__ __launch_bounds__(256) void pack_conv_dgrad_kernel(const float* __restrict__ W, int F_, int D_, u16* __restrict__ Wd2) {
  const int total = KP * 3 * KP;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int d = i / (3 * KP), rem = i - d * 3 * KP, tap = rem / KP, f = rem - tap * KP;
    Wd2[i] = f2bf((d < D_ && f < F_) ? W[((size_t)f * 3 + (2 - tap)) * D_ + d] : 0.0f);
  }
}

// out[m][n] (+)= sum_p parts[p][m][n]: the split-K partials of the TN kernel summed in a FIXED order (deterministic).  A workgroup = 64 float4
// columns x 4 partition groups: group g sums partitions g, g + 4, ... with four loads in flight, the groups are combined through LDS in group
// order.  (One thread per float4 walking all P partitions -- the first version -- ran at 1.4 TB/s: 257 us per NAML step for 360 MB of partials.)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ parts, int P, int64_t n4, float* __restrict__ out, int accumulate) {
  NR_SMEM_DECL(smem);
  f32x4* red = (f32x4*)smem;                                      // [4][64]
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < n4; base += (int64_t)gridDim.x * 64) {
    const int64_t i = base + c;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
      int q = g;
      for (; q + 12 < P; q += 16) {
        const f32x4 a0 = *(const f32x4*)(parts + ((int64_t)q * n4 + i) * 4), a1 = *(const f32x4*)(parts + ((int64_t)(q + 4) * n4 + i) * 4);
        const f32x4 a2 = *(const f32x4*)(parts + ((int64_t)(q + 8) * n4 + i) * 4), a3 = *(const f32x4*)(parts + ((int64_t)(q + 12) * n4 + i) * 4);
        s += a0; s += a1; s += a2; s += a3;
      }
      for (; q < P; q += 4) s += *(const f32x4*)(parts + ((int64_t)q * n4 + i) * 4);
    }
    red[g * 64 + c] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
      f32x4 t = accumulate ? *(const f32x4*)(out + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      t += red[c]; t += red[64 + c]; t += red[128 + c]; t += red[192 + c];
      *(f32x4*)(out + i * 4) = t;
    }
    __syncthreads();
  }
}

}  // namespace nr
