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

  // Fragment set of one k-step (16 contraction indices): TM fragments of A, TN_ of B.  ONE set lives in registers: while the MFMAs of a row
  // group (one A fragment x the TN_ B fragments) run, the A fragment they just consumed is re-read for the NEXT k-step into the same registers;
  // the next B fragments arrive in a second small set.  (Two whole sets, as dx_gemm_ring_kernel keeps them, cost 2 x 28 registers beside the
  // 160 accumulator registers of the 320 x 256 tile: 98 spills, and spill reloads travel through vmcnt and drain the copy ring.)
  u16x8 af[TM], bf[TN_];
  const int prow = (l & 15) >> 2, pcol = 16 * ((l >> 4) & 1) + 4 * (l & 3);       // TN: geometry of the transposing reads (see lds_tr16_b64)
  auto read_a = [&](int c, int ks, int a) -> u16x8 {
    const unsigned char* abuf = smem + (c & (Gm::NB - 1)) * Gm::BUF;
    if (!TN) {
      const int row = (wr * TM + a) * 32 + li;
      return *(const u16x8*)(abuf + row * 64 + (((ks * 2 + h) ^ ((row >> 2) & 3)) * 16));
    }
    u16x4 v[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 16 * ks + 8 * h + 4 * t + prow, col = 32 * (wr * TM + a) + pcol;
      v[t] = lds_tr16_b64((const u16*)(abuf + (row * Gm::SLA + ((col >> 3) ^ tn_swz<Gm::SLA>(row))) * 16 + (col & 7) * 2));
    }
    return cat8(v[0], v[1]);
  };
  auto read_b = [&](int c, int ks, int j) -> u16x8 {
    const unsigned char* bbuf = smem + (c & (Gm::NB - 1)) * Gm::BUF + Gm::A_BYTES;
    if (!TN) {
      const int row = (wc * TN_ + j) * 32 + li;
      return *(const u16x8*)(bbuf + row * 64 + (((ks * 2 + h) ^ ((row >> 2) & 3)) * 16));
    }
    u16x4 v[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 16 * ks + 8 * h + 4 * t + prow, col = 32 * (wc * TN_ + j) + pcol;
      v[t] = lds_tr16_b64((const u16*)(bbuf + (row * Gm::SLB + ((col >> 3) ^ tn_swz<Gm::SLB>(row))) * 16 + (col & 7) * 2));
    }
    return cat8(v[0], v[1]);
  };
  static_assert(TM * TN_ >= Gm::CP, "one copy per MFMA at most");
  // the MFMAs of the k-step in registers; NEXT: its fragments are replaced by those of k-step (nc, nks) on the way; cnext >= 0: the CP copies of
  // chunk cnext are issued between the MFMAs, not as a burst behind the barrier (a copy holds the issue slot: see dx_gemm_ring_kernel)
  auto multiply = [&](auto next_tag, int nc, int nks, int cnext) {
    constexpr bool next = decltype(next_tag)::v;                  // compile time: no control flow around the accumulators
    constexpr bool AMAJ = TM >= TN_;                              // the LONGER operand side is refilled in place, the shorter one through a second set
    constexpr int NO = AMAJ ? TM : TN_, NI = AMAJ ? TN_ : TM;
    u16x8 nin[NI];
    if (next) {
#pragma unroll
      for (int i = 0; i < NI; ++i) nin[i] = AMAJ ? read_b(nc, nks, i) : read_a(nc, nks, i);
    }
    NR_SCHED_BARRIER();
#pragma unroll
    for (int o = 0; o < NO; ++o) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int a = AMAJ ? o : i, j = AMAJ ? i : o;
        acc[a][j] = mfma_32x32x16_bf16(af[a], bf[j], acc[a][j]);                // C[m][n]: the lane holds column n = l & 31, rows 8 q + 4 h + e
        if (o * NI + i < Gm::CP && cnext >= 0 && cnext < nchunk) piece(cnext, o * NI + i);           // (compile-time copy index)
      }
      if (next) {
        if (AMAJ) af[o] = read_a(nc, nks, o);
        else bf[o] = read_b(nc, nks, o);
      }
      NR_SCHED_BARRIER();
    }
    if (next) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (AMAJ) bf[i] = nin[i];
        else af[i] = nin[i];
      }
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
    for (int a = 0; a < TM; ++a) af[a] = read_a(0, 0, a);
#pragma unroll
    for (int j = 0; j < TN_; ++j) bf[j] = read_b(0, 0, j);
    int c = 0;
    for (; c + 1 < nchunk; ++c) {
      // registers hold k-step 0 of chunk c; the slot of chunk c - 1 is free (arrive(c) has passed): chunk c + 3 goes there
      multiply(GemmYes{}, c, 1, c >= 1 ? c + Gm::NB - 1 : -1);
      arrive(c + 1);
      multiply(GemmYes{}, c + 1, 0, -1);
    }
    multiply(GemmYes{}, c, 1, -1);                                // the last chunk: nothing left to copy
    multiply(GemmNo{}, 0, 0, -1);
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

// out[m][n] (+)= sum_p parts[p][m][n]: the split-K partials of the TN kernel summed in a fixed order (deterministic); float4 per lane
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ parts, int P, int64_t n4, float* __restrict__ out, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 s = accumulate ? *(const f32x4*)(out + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < P; ++q) s += *(const f32x4*)(parts + ((int64_t)q * n4 + i) * 4);
    *(f32x4*)(out + i * 4) = s;
  }
}

}  // namespace nr
