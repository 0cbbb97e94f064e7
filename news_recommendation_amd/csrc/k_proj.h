// Training form of the NRMS news-encoder front end as TWO kernels with a GEMM core (gfx950):
//
//   qkv_proj_kernel   x = dropout(table[ids])  ->  Q | K | V = x W^T + b   (src/model/NRMS/news_encoder.py:38-40,
//                     src/model/general/attention/multihead_self.py:53-58).  A gather-fused projection GEMM
//                     [tokens x 304] x [304 x 960] on v_mfma_f32_32x32x16_bf16: a wave owns 32 tokens whose operand fragments stay in
//                     registers (76 VGPRs) for all 30 column chunks; the weights stream through LDS in 32-column chunks
//                     (global_load_lds_dwordx4, fragment-major "tile32 order", shared by the four waves, double buffered).
//   attn_fwd_kernel   ScaledDotProductAttention (multihead_self.py:15-23) per (title, head) from the saved Q, K, V: two waves per title,
//                     writes the ctx rows (second dropout of news_encoder.py:43-45 applied) the pooling kernels read.
//
// Both kernels move every byte to and from global memory as whole cache lines.  The first version of this file loaded the table rows
// fragment-shaped (32 rows x 16 B per instruction) and stored Q / K / V^T / X / ctx as 8-byte pieces of 16 - 32 different rows per
// instruction: its phase decomposition (profiles/r03b_split_forward_phases.txt) put 483 of the projection kernel's 710 us on those stores,
// 177 us on the loads and only 103 us on the MFMAs; 322 of the attention kernel's 406 us were loads and stores.  Now:
//   * table rows are read as contiguous float4 pieces (a wave instruction = 1 KiB of one row), masked / rounded, written to a wave-private
//     LDS tile [4 rows][320] -- from which the x_save rows leave as one contiguous 2.5 KiB run and the lanes pick up their operand fragments;
//   * the projection accumulators of a 64-column group (= 3 heads + 4 padding columns: the packed weight rows are ordered that way) are
//     staged in the same wave-private tile as [head][token][20] and leave as contiguous runs of the head-major layout below;
//   * the attention kernel fetches a pair's 2,400 operand bytes as 16-byte pieces, assembles the title's ctx rows [20][320] in LDS and
//     writes them as one contiguous 12.8 KB run.
//
// The register-resident single-wave kernel (k_mhsa_fwd2.h) stays the inference form: it never writes Q / K / V.
//
// Saved-activation layout shared by both kernels and nr_attn_bwd_hm ("head-major"): qkv bf16 [n_seq][15][3][20][20]:
//   per (title, head) the blocks Q, K, V, each [token][d] row-major, back to back: 2,400 contiguous bytes per pair, 36,000 per title.
#pragma once
#include "nr_common.h"
#include "k_additive_fwd.h"
#include <type_traits>

namespace nr {

constexpr int K16 = (D + 15) / 16;          // 19 k-steps of 16 cover the 300 features (304)
constexpr int NT32 = NP / 32;               // 10 column tiles of 32 per projection
constexpr int HM_BLK = 20 * DK;             // 400 elements: one 20 x 20 block
constexpr int HM_PAIR = 3 * HM_BLK;         // 1,200 elements per (title, head)
constexpr int PG_COLS = 64;                 // packed column group: 3 heads (60 columns) + 4 zero columns
constexpr int PG_HEADS = 3;

// "tile32 order" of the packed projection matrix W[3 * NP][304] (operand of v_mfma_f32_32x32x16_bf16): block (32-row tile, 16-wide
// k-step) = 1 KiB = the 64 lanes' 16-byte fragments back to back; lane l holds W[32 T + (l & 31)][16 ks + 8 (l >> 5) + 0..7].
__device__ __host__ __forceinline__ size_t tile32_off(int r, int k) {
  return ((size_t)(r >> 5) * K16 + (k >> 4)) * 512 + ((((k & 15) >> 3) * 32) + (r & 31)) * 8 + (k & 7);
}
// packed row c (0 .. NP-1) of a projection -> output feature (head * 20 + d), or -1 for the 4 padding rows of each 64-row group
__device__ __host__ __forceinline__ int packed_row_feature(int c) {
  const int r = c % PG_COLS;
  return r < PG_HEADS * DK ? (c / PG_COLS) * (PG_HEADS * DK) + r : -1;
}

__device__ __forceinline__ void pack_qkv32_body(const float* __restrict__ Wq, const float* __restrict__ bq,
                                                const float* __restrict__ Wk, const float* __restrict__ bk,
                                                const float* __restrict__ Wv, const float* __restrict__ bv,
                                                u16* __restrict__ Wp32, float* __restrict__ bp, int bid, int nb) {
  constexpr int KW = K16 * 16;
  const int total = 3 * NP * KW;
  for (int i = bid * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
    const int row = i / KW, k = i - row * KW;
    const int which = row / NP, n = packed_row_feature(row - which * NP);
    const float* W = which == 0 ? Wq : (which == 1 ? Wk : Wv);
    const float* b = which == 0 ? bq : (which == 1 ? bk : bv);
    // columns D, D + 1 carry the bias as two bf16 numbers (hi + lo: 16 mantissa bits) against the two 1.0 columns of the token rows: the bias
    // rides the contraction -- no bias loads between a chunk's weight copies and its stores, nothing in the chunk loop waits on vmcnt for data
    float v = (n >= 0 && k < D) ? W[n * D + k] : 0.0f;
    if (n >= 0 && k == D) v = bf2f(f2bf(b[n]));
    if (n >= 0 && k == D + 1) v = b[n] - bf2f(f2bf(b[n]));
    Wp32[tile32_off(row, k)] = f2bf(v);
    if (k == 0) bp[row] = n >= 0 ? b[n] : 0.0f;
  }
}
__global__ __launch_bounds__(256) void pack_qkv32_kernel(const float* __restrict__ Wq, const float* __restrict__ bq,
                                                         const float* __restrict__ Wk, const float* __restrict__ bk,
                                                         const float* __restrict__ Wv, const float* __restrict__ bv,
                                                         u16* __restrict__ Wp32, float* __restrict__ bp) {
  pack_qkv32_body(Wq, bq, Wk, bk, Wv, bv, Wp32, bp, blockIdx.x, gridDim.x);
}

#ifndef NR_PROJ_NWAVE
#define NR_PROJ_NWAVE 8    // waves per workgroup (tuning knob of the build): 8 = 256 tokens, two workgroups per CU, four waves per SIMD (default: 502 vs
                           // 572 us for 4 = 128 tokens, three per CU, A/B on one MI355X: half the weight-chunk copies and barriers per token); 16 = 512 tokens, one per CU
#endif
struct ProjGeom {
  static constexpr int S = 20;
  static constexpr int NWAVE = NR_PROJ_NWAVE;
  static constexpr int TOKW = 32;                    // tokens per wave: one 32-row MFMA tile
  static constexpr int TOK_WG = NWAVE * TOKW;        // 128
  static constexpr int CH_BYTES = K16 * 1024;        // 19,456 B: one 32-column chunk of W = 19 fragment blocks
  static constexpr int RP = 4;                       // token rows per gather pass
  static constexpr int NPASS = TOKW / RP;            // 8
  static constexpr int XROW = 656;                   // bytes per staged token row: 320 bf16 + 16 (rows shift by 4 banks, 16-byte aligned)
  static constexpr int STAGE_BYTES = PG_HEADS * TOKW * DK * 2;      // 3,840 B per wave: the projection epilogue's [3 heads][32 tokens][20]; the gather
                                                     // passes use the first RP * XROW = 2,624 B
  static constexpr int SMEM = 2 * CH_BYTES + NWAVE * STAGE_BYTES;   // 69,632 B (8 waves): two workgroups per CU (163,840 B)
  static constexpr int NCHUNK = 3 * NT32;            // 30
  static constexpr int QPR = D / 4;                  // 75 float4 quads per table row
  static constexpr int LD_IT = (RP * QPR + 63) / 64; // 10 row pieces per lane and pass
  static constexpr int WO_PC = TOKW * DK * 2 / 16;   // 80 sixteen-byte pieces per head of a column group
  static constexpr int WO_IT = (PG_HEADS * WO_PC + 63) / 64;            // 4 sixteen-byte pieces per lane and column group
  static_assert(STAGE_BYTES >= RP * XROW && (NWAVE == 4 ? 3 : (NWAVE == 8 ? 2 : 1)) * SMEM <= 163840, "the gather tile fits the epilogue staging; three / two / one workgroups per CU");
};

struct ProjParams {
  const int64_t* ids;      // [n_tok]
  const float* table;      // [num_rows][D]
  int64_t num_rows;
  const u16* Wp32;         // [3*NP][304] in tile32 order, rows in packed order (packed_row_feature)
  const float* bp;         // [3*NP] in the same row order
  u16* qkv;                // [n_tok / 20][H][3][20][20] head-major Q, K, V
  u16* x_save;             // optional [n_tok][KP]: the dropout-masked bf16 token matrix (col D = 1.0) for the weight-gradient GEMM
  int64_t n_tok;           // multiple of 20
  DropCfg dc;              // dropout site 1
  int debug;               // profiling only (NR_PROJ_DEBUG, DBG instantiation): 1 skip the table loads, 2 skip the MFMAs, 4 skip the Q / K / V
                           // stores, 8 skip the x_save stores, 16 skip the weight-chunk copies
};

#ifndef NR_PROJ_OCC
#define NR_PROJ_OCC (NR_PROJ_NWAVE == 4 ? 3 : 4)      // waves per SIMD the register allocation must allow
#endif
// One accumulator chain per chunk (a second one for the odd k-steps spills 13 registers under the 128-register cap of four waves per SIMD:
// 648 vs 489 us, profiles/r05_ab_notes.txt).
//
// Round 5: nothing in the kernel waits for a STORE any more.  Its phase decomposition (profiles/r05_proj_phases.txt) had every phase additive --
// table loads 119 us, MFMAs 125, Q / K / V stores 175, x_save stores 85, weight copies 111 of 618 -- because each phase ended in a wait that
// covered the stores issued before it (one in-order memory counter per wave): __syncthreads() drains vmcnt to 0 while a global -> LDS copy is in
// flight, i.e. also the Q / K / V stores issued a moment earlier; the bias loads at the top of a chunk sat behind the previous chunk's stores; a
// gather pass waited for its table rows behind the x_save stores of the pass before.  Now: the bias rides the contraction (two bf16 columns
// against the token rows' two 1.0 columns: no loads in the chunk loop), the chunk barrier is a raw s_barrier behind a COUNTED wait that leaves
// the chunk's own stores in flight, and the table rows of pass p + 1 are requested before the x_save stores of pass p are issued.
template <bool DBG>
__global__ __launch_bounds__(ProjGeom::NWAVE * 64, NR_PROJ_OCC) void qkv_proj_kernel(ProjParams p) {
  using Gm = ProjGeom;
  constexpr int S = Gm::S;
  const int dbg = DBG ? p.debug : 0;
  p.dc = drop_resolve(p.dc);
  NR_SMEM_DECL(smem);
  int l = lane_id(), h = l >> 5, li = l & 31;                   // (re-derived after the gather phase, see below)
  const int w = wave_id();
  const int64_t tile_tok0 = ((int64_t)blockIdx.x * Gm::NWAVE + w) * Gm::TOKW;
  unsigned char* const stage = smem + 2 * Gm::CH_BYTES + w * Gm::STAGE_BYTES;      // wave-private

  // chunk c = 32 consecutive rows of the packed matrix: global -> LDS directly; the image is fragment-major already (tile32 order), so
  // block ks is read back as ONE conflict-free ds_read_b128 at block + 16 lane.  Blocks w, w + 4, ...: waves 0 - 2 issue five copies, wave 3 four
  auto chunk_fetch = [&](int c, int buf) {
    if (dbg & 16) return;
    const u16* src = p.Wp32 + (size_t)c * K16 * 512 + l * 8;
    unsigned char* dst = smem + buf * Gm::CH_BYTES;
    for (int blk = w; blk < K16; blk += Gm::NWAVE) NR_GLDS16(src + blk * 512, dst + blk * 1024);
  };
  chunk_fetch(0, 0);

  // ---- gather: 8 passes of 4 token rows.  Row pieces (float4 = one dropout quad) are read lane-linear along the rows, masked, rounded and
  // written to the wave's LDS tile; the tile leaves as 4 x 640 contiguous bytes of x_save, and the 8 lanes that own these 4 tokens take
  // their 19 operand fragments (features 16 ks + 8 h .. + 7 of token li) from it.  The rows of pass p + 1 are requested as soon as pass p's rows
  // are in LDS, BEFORE its x_save stores are issued: their wait then never covers those stores (the memory counter returns in issue order) ----
  u16x8 xf[K16];
  int myrow = 0;                                                  // table row of token li (lanes li and li + 32 hold the same)
  if (tile_tok0 + li < p.n_tok) {
    int64_t id = p.ids[tile_tok0 + li];
    id = id < 0 ? 0 : (id >= p.num_rows ? p.num_rows - 1 : id);
    myrow = (int)id;
  }
  if (l < Gm::RP * 5) {                                           // K padding of the staged rows: cols D, D + 1 = 1.0 (bias hi / lo parts), D + 2 .. KP - 1 = 0 (not rewritten per pass)
    const int r = l / 5, cq = l - r * 5;
    *(u16x4*)(stage + r * Gm::XROW + (D + cq * 4) * 2) = u16x4{(u16)(cq == 0 ? BF16_ONE : 0), (u16)(cq == 0 ? BF16_ONE : 0), 0, 0};
  }
  const BufRsrc r_xs = make_buf(p.x_save != nullptr ? p.x_save + tile_tok0 * KP : nullptr,
                                (p.x_save != nullptr && !(dbg & 8) && tile_tok0 < p.n_tok)
                                    ? (uint32_t)((p.n_tok - tile_tok0 < Gm::TOKW ? p.n_tok - tile_tok0 : Gm::TOKW) * KP * 2) : 0u);      // rows past the end: stores vanish
  // table rows through a buffer resource: one 32-bit byte offset per piece (row * 1,200 + quad * 16) instead of a 64-bit address pair -- the
  // address pairs of a pass were what the register allocator spilled; a piece with nothing to read (past the pass / the last token) gets an
  // offset behind the resource and reads zeros
  const BufRsrc r_tab = make_buf(p.table, (dbg & 1) ? 0u : (uint32_t)((uint64_t)p.num_rows * (D * 4) < 0xFFFFFFFFull ? (uint64_t)p.num_rows * (D * 4) : 0xFFFFFFFFull));
  auto load_pass = [&](int ps, f32x4 (&raw)[Gm::LD_IT]) {
    int lq = l;
    NR_OPAQUE(lq);
#pragma unroll
    for (int i = 0; i < Gm::LD_IT; ++i) {
      const int idx = lq + 64 * i;
      const int r = idx / Gm::QPR, qd = idx - r * Gm::QPR;
      const int rid = __builtin_bit_cast(int, shfl(__builtin_bit_cast(float, myrow), ps * Gm::RP + (r < Gm::RP ? r : 0)));
      const bool ok = idx < Gm::RP * Gm::QPR && tile_tok0 + ps * Gm::RP + r < p.n_tok;
      raw[i] = buf_load16f<0>(r_tab, ok ? (uint32_t)rid * (uint32_t)(D * 4) + (uint32_t)(qd * 16) : 0xFFFFFFF0u);
    }
  };
  {
    const uint64_t qbase = (uint64_t)tile_tok0 * D4;              // dropout quad index of the wave's first element: scalar; the lane adds 32 bits
    f32x4 raw[Gm::LD_IT];
    load_pass(0, raw);
#pragma unroll 1
    for (int ps = 0; ps < Gm::NPASS; ++ps) {
#pragma unroll
      for (int i = 0; i < Gm::LD_IT; ++i) {
        const int idx = l + 64 * i;
        const int r = idx / Gm::QPR, qd = idx - r * Gm::QPR;
        if (idx < Gm::RP * Gm::QPR) {
          f32x4 a = raw[i];
          if (p.dc.enabled) a = a * drop_mul4(p.dc, 1u, qbase + (uint32_t)((ps * Gm::RP + r) * D4 + qd));
          *(u16x4*)(stage + r * Gm::XROW + qd * 8) = pack4(a);
        }
      }
      NR_SCHED_BARRIER();
      if (ps + 1 < Gm::NPASS) load_pass(ps + 1, raw);             // requested BEFORE this pass's x_save stores are issued (same registers: the rows above are in LDS)
      NR_SCHED_BARRIER();
      wave_barrier();
      {                                                           // the pass's rows are RP x 640 contiguous bytes of x_save
        int lq = l;
        NR_OPAQUE(lq);
#pragma unroll
        for (int i = 0; i < (Gm::RP * (KP / 8) + 63) / 64; ++i) {
          const int idx = lq + 64 * i;
          const int r = idx / (KP / 8), pc = idx - r * (KP / 8);
          if (idx < Gm::RP * (KP / 8)) buf_store16<0>(r_xs, (uint32_t)(idx * 16), *(const u16x8*)(stage + r * Gm::XROW + pc * 16), (uint32_t)(ps * Gm::RP * KP * 2));
        }
      }
      if (li / Gm::RP == ps) {
        const unsigned char* src = stage + (li % Gm::RP) * Gm::XROW + h * 16;
#pragma unroll
        for (int ks = 0; ks < K16; ++ks) xf[ks] = *(const u16x8*)(src + ks * 32);
      }
      wave_barrier();
    }
  }
  __syncthreads();
  // The lane coordinates of the product phase are taken from the hardware lane counter: as values derived from the work-item id they were live
  // across the gather phase's register peak, where the allocator (128 registers for four waves per SIMD) spilled two of them -- 12 bytes of
  // scratch per lane.  Without the spill: 473 -> 467 us per launch on one box, step unchanged within noise (profiles/r06_ab_qkv_scratch.txt;
  // the ~35 us dispatch gap a rocprofv3 trace shows in front of this kernel is there with and without scratch: an artefact of the tracer)
  l = lane_id_fresh(); h = l >> 5; li = l & 31;

  // write-out geometry of a column group (fixed per lane): 16-byte piece idx = l + 64 i of [3 heads][32 tokens x 40 B].  A wave's 32 tokens
  // start at an even position of their title (32 k mod 20 is even) and title fragments hold an even number of tokens, so every fragment
  // starts on a 16-byte boundary of its head-major block and is a whole number of pieces: a piece may span two token rows, never two titles.
  // Through a buffer resource on the wave's titles: a lane with nothing to write (the tail of the last tile, idx >= 240) gets an offset past
  // the resource -- the store instruction is ALWAYS issued (its count is what the chunk barrier's wait relies on) and writes nothing
  const int64_t seq_base = tile_tok0 / S;
  const int64_t seq_end = (tile_tok0 + Gm::TOKW - 1) / S + 1;          // titles the wave's tokens touch
  const int64_t nseq_all = p.n_tok / S;
  const uint32_t wave_bytes = (dbg & 4) || tile_tok0 >= p.n_tok ? 0u : (uint32_t)(((seq_end < nseq_all ? seq_end : nseq_all) - seq_base) * (H * HM_PAIR * 2));
  const BufRsrc r_qkv = make_buf(p.qkv + seq_base * (H * HM_PAIR), wave_bytes);
  uint32_t wo_off[Gm::WO_IT];      // byte offset inside the wave's titles, 0xFFFFFFFF: nothing to write
#pragma unroll
  for (int i = 0; i < Gm::WO_IT; ++i) {
    const int idx = l + 64 * i;
    const int hh = idx / Gm::WO_PC, pc = idx - hh * Gm::WO_PC;
    const int t = (2 * pc) / 5, e = 8 * pc - DK * t;            // first token the piece touches, element offset inside its row
    const int64_t tok = tile_tok0 + t;
    const int sq = (int)(tok / S - seq_base), tis = (int)(tok % S);
    wo_off[i] = (idx < PG_HEADS * Gm::WO_PC && tok < p.n_tok) ? (uint32_t)(((sq * H + hh) * HM_PAIR + tis * DK + e) * 2) : 0xFFFFFFF0u;
  }

  // one chunk: prefetch the next, 19 MFMAs of the transposed product (A = weights: the lane ends up with 4 x 4 consecutive features of ITS
  // token), accumulators -> staging tile; after the second chunk of a 64-column group the three heads leave as contiguous runs
  auto run_chunk = [&](int which, int j, int s, int c) {
    if (c + 1 < Gm::NCHUNK) chunk_fetch(c + 1, (c + 1) & 1);
    const u16* wp = (const u16*)(smem + (c & 1) * Gm::CH_BYTES) + l * 8;
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
    // weight fragments are requested PF k-steps ahead of their MFMA (the compiler's own schedule: two reads, wait, two MFMAs)
    constexpr int PF = ProjGeom::NWAVE >= 8 ? 4 : 6;
    u16x8 wf[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) wf[i] = *(const u16x8*)(wp + i * 512);
    NR_SCHED_BARRIER();
#pragma unroll
    for (int ks = (dbg & 2) ? K16 : 0; ks < K16; ++ks) {
      acc[0] = mfma_32x32x16_bf16(wf[ks % PF], xf[ks], acc[0]);
      if (ks + PF < K16) wf[ks % PF] = *(const u16x8*)(wp + (ks + PF) * 512);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = s * 32 + 8 * q + 4 * h;                     // column inside the group: head r / 20, feature r % 20 (a quad never straddles heads)
      if (r < PG_HEADS * DK) {
        const int hh = r / DK, d = r - hh * DK;
        *(u16x4*)(stage + ((hh * Gm::TOKW + li) * DK + d) * 2) = pack4(f32x4{acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]});
      }
    }
    if (s == 1) {
      wave_barrier();
      const uint32_t soff = (uint32_t)((j * PG_HEADS * 3 + which) * HM_BLK * 2);
#pragma unroll
      for (int i = 0; i < Gm::WO_IT; ++i) buf_store16<0>(r_qkv, wo_off[i], *(const u16x8*)(stage + (l + 64 * i) * 16), soff);
    }
    // this wave's copies of chunk c + 1 have landed once at most the stores issued after them are outstanding (exactly WO_IT when s == 1);
    // the LDS reads of chunk c's buffer and of the staging tile have returned; then everybody meets: chunk c + 1 is complete and visible,
    // chunk c's buffer may take the copies of chunk c + 2.  No fence: the stores stay in flight
    if (s == 1) NR_WAIT_VMCNT(Gm::WO_IT); else NR_WAIT_VMCNT(0);
    NR_WAIT_LGKMCNT(0);
    NR_BARRIER_RAW();
  };
  int c = 0;
  for (int which = 0; which < 3; ++which)
    for (int j = 0; j < NT32 / 2; ++j) {
      run_chunk(which, j, 0, c++);
      run_chunk(which, j, 1, c++);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
struct AttnFwdParams {
  const u16* qkv;          // [n_seq][H][3][20][20] head-major Q, K, V (qkv_proj_kernel)
  u16* ctx;                // [n_seq*20][KP]: cols < D = attention output (x dropout 2), col D = 1.0, the rest of the K padding 0
  const int32_t* key_len;  // optional [n_seq]: keys >= key_len[seq] get zero weight (MultiHeadSelfAttention's `length`); null: 20
  int64_t n_seq;
  DropCfg dc;              // dropout site 2
  int debug;               // profiling only (NR_ATTNF_DEBUG, DBG instantiation): 1 skip the operand loads, 2 skip the exp / normalisation, 4 skip the ctx stores
  AdditiveParams pool;     // POOL instantiation: the additive pooling of the titles (k_additive_fwd.h) runs on the workgroup's ctx tiles while
                           // they are still in LDS (pool.ctx / pool.n_seq unused: the tile and n_seq above are taken)
};

struct AttnFwdGeom {
  static constexpr int S = 20;
  static constexpr int WPB = 8;                       // waves per workgroup: two per title (heads 0 .. 7 and 8 .. 14)
  static constexpr int TPB = WPB / 2;                 // titles per workgroup
  static constexpr int HSPLIT = 8;
  static constexpr int CROW = 656;                    // bytes per staged ctx row (320 bf16 + 16)
  static constexpr int TILE_BYTES = S * CROW;         // 13,120 B per title
  static constexpr int OPER_BYTES = HM_PAIR * 2;      // 2,400 B per wave: Q, K, V of one pair
  static constexpr int ZERO_BYTES = 2 * 800 + 16;     // lanes without a k-slot read zeros at block offsets 0 and 800 (Q, K row fragments)
  static constexpr int SMEM = TPB * TILE_BYTES + WPB * OPER_BYTES + 1664;   // 73,344 B: two workgroups per CU = 4 waves per SIMD
  static constexpr int OP_IT = (OPER_BYTES / 16 + 63) / 64;      // 3 sixteen-byte pieces per lane and pair
  static constexpr int WO_ROWS = S / 2;               // rows each of the title's two waves writes out
  static constexpr int WO_IT = (WO_ROWS * (KP / 8) + 63) / 64;   // 7 sixteen-byte pieces per lane
  // pooled form: the TPB title tiles back to back ARE the [80][XS] token tile of AddGeom<S, TPB, WPB>; its scratch reuses the operand buffers
  using Pool = AddGeom<S, TPB, WPB>;
  static_assert(CROW == XS * 2 && TPB * TILE_BYTES == Pool::X_BYTES, "title tiles form the pooling kernel's token tile");
  static_assert(Pool::SC_BYTES + Pool::W_BYTES <= WPB * OPER_BYTES, "pooling scratch fits the operand buffers");
};

template <bool DBG, bool POOL>
__global__ __launch_bounds__(AttnFwdGeom::WPB * 64, 4) void attn_fwd_kernel(AttnFwdParams p) {
  using Gm = AttnFwdGeom;
  constexpr int S = Gm::S;
  const int dbg = DBG ? p.debug : 0;
  p.dc = drop_resolve(p.dc);
  NR_SMEM_DECL(smem);
  const int l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  const int half = w & 1;                                         // this wave's share of the title's heads
  const int64_t seq = (int64_t)blockIdx.x * Gm::TPB + (w >> 1);
  const bool have = seq < p.n_seq;
  if (!POOL && !have) return;                                     // (both waves of a title leave together; exited waves do not hold up the barrier)
  unsigned char* const tile = smem + (w >> 1) * Gm::TILE_BYTES;   // ctx rows of the title, shared by its two waves (disjoint columns)
  unsigned char* const oper = smem + Gm::TPB * Gm::TILE_BYTES + w * Gm::OPER_BYTES;      // operands of this wave's current pair
  unsigned char* const zero = smem + Gm::TPB * Gm::TILE_BYTES + Gm::WPB * Gm::OPER_BYTES;      // lanes without a k-slot read zeros here
  const int hd0 = half * Gm::HSPLIT, hd1 = half ? H : Gm::HSPLIT;

  for (int i = l; i * 8 < Gm::ZERO_BYTES; i += 64) *(u16x4*)(zero + i * 8) = u16x4{0, 0, 0, 0};      // (every wave writes the same zeros: no barrier needed before its own reads)
  if (have) {                                                     // (a pooled workgroup keeps the waves of missing titles for the pooling)
  // K padding of the staged ctx rows: col D = 1.0 (bias-gradient column), cols D + 1 .. KP - 1 = 0
  if (half == 0)
    for (int i = l; i < S * 5; i += 64) {
      const int r = i / 5, cq = i - r * 5;
      *(u16x4*)(tile + r * Gm::CROW + (D + cq * 4) * 2) = u16x4{(u16)(cq == 0 ? BF16_ONE : 0), 0, 0, 0};
    }
  // a pair's Q | K | V block (2,400 contiguous bytes) goes global -> LDS directly (round 5, as in k_attn_bwd2.h): three copies with a scalar
  // base, issued as soon as the previous pair's fragments are in registers -- no staging registers, no LDS stores
  const unsigned char* const qkv_seq = (const unsigned char*)(p.qkv + seq * (H * HM_PAIR));
  auto fetch = [&](int hd) {
    if (dbg & 1) return;
    const unsigned char* base = qkv_seq + hd * (HM_PAIR * 2);
    const unsigned lo = (unsigned)l * 16u;
    NR_GLDS16_S(base, lo, oper);
    NR_GLDS16_S(base, lo + 1024u, oper + 1024);
    if (l < Gm::OPER_BYTES / 16 - 128) NR_GLDS16_S(base, lo + 2048u, oper + 2048);
  };
  fetch(hd0);

  const int klen = p.key_len != nullptr ? uniform(clamp_len(p.key_len[seq], S)) : S;
  const float c2 = LOG2E / sqrtf((float)DK), clamp2 = EXP_CLAMP * LOG2E;
  // Fragment shapes (v_mfma_f32_16x16x32_bf16, lane = (li, g)):
  //   K / Q tile t as A / B operand of S^T = K Q^T: row 16 t + li (clamped to 19: duplicates that are masked as keys / never stored as
  //   queries), k-slots d = 8 g .. 8 g + 7; halves with d >= 20 read the zero block; slot d = 20 carries the key mask (q = 1, k = 0 for a
  //   live key, -29952 for a padded one: exp2 of it underflows to exactly 0, no select per element);
  //   the packed P^T tiles are the B operand of ctx^T = V^T P^T with k-slot (g, j < 4) = key 4 g + j and (g, j >= 4) = key 16 + 4 g + j - 4:
  //   V^T as the A operand in exactly those slots comes from the TRANSPOSING LDS read of the row-major V block (rows = keys): lane (li, g)
  //   supplies the piece &V[16 t + 4 g + (li >> 2)][16 dt + 4 (li & 3)] and receives V[16 t + 4 g + 0 .. 3][16 dt + li] (round 4: four
  //   identity MFMAs + eight conversions per pair)
  const int zoff = (int)(zero - oper);
  int lo_off[2], hi_off[2], tr_off[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = 16 * t + li < S ? 16 * t + li : S - 1;
    lo_off[t] = g < 3 ? row * 40 + 16 * g : zoff;
    hi_off[t] = g < 2 ? row * 40 + 16 * g + 8 : zoff;
    const int trow = 16 * t + 4 * g + (li >> 2) < S ? 16 * t + 4 * g + (li >> 2) : S - 1;
    tr_off[t] = trow * 40 + 8 * (li & 3);
  }
  const uint32_t qone = g == 2 ? (uint32_t)BF16_ONE : 0u;
  const uint32_t kmask[2] = {(g == 2 && li >= klen) ? (uint32_t)BF16_NEG_BIG : 0u, (g == 2 && 16 + li >= klen) ? (uint32_t)BF16_NEG_BIG : 0u};
  constexpr int BLK = HM_BLK * 2;

  for (int hd = hd0; hd < hd1; ++hd) {
    NR_WAIT_VMCNT(0);                                // this pair's copy has landed (the copies are asm: nothing else waits for them)
    wave_barrier();
    u16x8 kf[2], qf[2];
    u16x4 vt[2][2];                                  // V^T fragments [key tile][dv tile]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      qf[t] = cat8(*(const u16x4*)(oper + lo_off[t]), *(const u16x4*)(oper + hi_off[t]));
      kf[t] = cat8(*(const u16x4*)(oper + BLK + lo_off[t]), *(const u16x4*)(oper + BLK + hi_off[t]));
      const u16* trv = (const u16*)(oper + tr_off[t]);
      vt[t][0] = lds_tr16_b64_async<2 * BLK>(trv);
      vt[t][1] = lds_tr16_b64_async<2 * BLK + 32>(trv);
      qf[t] = or_dword2(qf[t], qone);
      kf[t] = or_dword2(kf[t], kmask[t]);
    }
    NR_WAIT_LGKMCNT(0);                              // the fragments are in registers: the operand buffer may take the next pair
    wave_barrier();
    NR_SCHED_BARRIER();
    if (hd + 1 < hd1) fetch(hd + 1);
    u16x4 pt[2][2];                                  // P^T [key tile][query tile]
#pragma unroll
    for (int qj = 0; qj < 2; ++qj) {
      f32x4 e[2];
      float sum = 0.0f;
#pragma unroll
      for (int ki = 0; ki < 2; ++ki) {
        e[ki] = mfma_16x16x32_bf16(kf[ki], qf[qj], f32x4{0.f, 0.f, 0.f, 0.f});
        if (!(dbg & 2)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            e[ki][r] = fast_exp2(fminf(e[ki][r] * c2, clamp2));
            sum += e[ki][r];
          }
        }
      }
      if (!(dbg & 2)) sum = sum_rows4(sum);
      const float rden = fast_rcp(sum + 1e-8f);      // exp / (sum + 1e-8): multihead_self.py:16-20 verbatim
      pt[0][qj] = pack4(e[0] * rden);
      pt[1][qj] = pack4(e[1] * rden);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int qj = 0; qj < 2; ++qj) {
        f32x4 acc = mfma_16x16x32_bf16(cat8(vt[0][t], vt[1][t]), cat8(pt[0][qj], pt[1][qj]), f32x4{0.f, 0.f, 0.f, 0.f});
        const int tokl = 16 * qj + li, dv = 16 * t + 4 * g;
        if (tokl < S && dv < DK) {
          const int col = hd * DK + dv;
          if (p.dc.enabled) acc = acc * drop_mul4(p.dc, 2u, (uint64_t)(seq * S + tokl) * D4 + (col >> 2));
          *(u16x4*)(tile + tokl * Gm::CROW + col * 2) = pack4(acc);
        }
      }
    }
  }
  }
  __syncthreads();                                   // both waves of the title have written their heads' columns
  if (have && !(dbg & 4)) {
    u16* dst = p.ctx + (seq * S + half * Gm::WO_ROWS) * KP;       // this wave's 10 rows are 6,400 contiguous bytes
#pragma unroll
    for (int i = 0; i < Gm::WO_IT; ++i) {
      const int idx = l + 64 * i;
      const int r = idx / (KP / 8), pc = idx - r * (KP / 8);
      if (idx < Gm::WO_ROWS * (KP / 8)) *(u16x8*)(dst + idx * 8) = *(const u16x8*)(tile + (half * Gm::WO_ROWS + r) * Gm::CROW + pc * 16);
    }
  }
  if (POOL) {
    // the operand buffers are free (every wave is past the barrier above): scores / softmax weights of the pooling live there
    float* sc = (float*)(smem + Gm::TPB * Gm::TILE_BYTES);
    float* wl = (float*)(smem + Gm::TPB * Gm::TILE_BYTES + Gm::Pool::SC_BYTES);
    AdditiveParams ap = p.pool;
    ap.n_seq = p.n_seq;
    additive_pool_tile<S, Gm::TPB, Gm::WPB>(ap, (const u16*)smem, sc, wl, (int64_t)blockIdx.x * Gm::TPB);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Input gradient of the three projections as a hand-written GEMM: dX[tok][n] = sum_k dqkv[tok][k] Wall[k][n], k = (which, feature) over the
// 960 columns of the dQ | dK | dV rows, n < 320 (autograd of multihead_self.py:53-55 w.r.t. its input; the reference leaves it to
// torch.autograd, src/train.py:231).  Accumulator-stationary: the contraction runs in 30 chunks of 32 columns: the token rows' 64-byte pieces
// and the 20 weight fragments of a chunk are copied global -> LDS directly (global_load_lds_dwordx4); the token pieces are placed with a 4-row
// XOR swizzle of their 16-byte slots so that the fragment reads are bank-conflict free; the result leaves through LDS as whole 640-byte rows.
// (The two-buffer form of round 3 -- 4 or 8 waves, one barrier per chunk -- lost to the ring below in every measurement and is gone:
// profiles/r03_ab_switches.txt.)
constexpr int DXK = 3 * KP;                 // 960 contraction columns
// packed operand of dx_gemm_ring_kernel: WdX bf16 [60 k-steps][10 n-tiles][64 lanes][8]: lane l of block (ks, nt) holds
// Wall[k = 16 ks + 8 (l >> 5) + j][n = 32 nt + (l & 31)], j = 0..7, Wall[which * KP + f][n] = W_which[f][n] (zero for f, n >= D)
// rowmajor != 0 (the process runs dX in the persistent stream kernel, conv_gemm_kernel<., PLAIN> of k_convgemm.h; same 307,200 elements): WdX bf16
// [KP n][960 k] row-major, WdX[n][which KP + f] = W_which[f][n]
__device__ __forceinline__ void pack_qkv_dx_body(const float* __restrict__ Wq, const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                 u16* __restrict__ WdX, int bid, int nb, int rowmajor) {
  const int total = DXK * KP;
  if (rowmajor) {
    for (int i = bid * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
      const int n = i / DXK, k = i - n * DXK;
      const int which = k / KP, f = k - which * KP;
      const float* W = which == 0 ? Wq : (which == 1 ? Wk : Wv);
      WdX[i] = f2bf((f < D && n < D) ? W[f * D + n] : 0.0f);
    }
    return;
  }
  for (int i = bid * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
    const int k = i / KP, n = i - k * KP;
    const int which = k / KP, f = k - which * KP;
    const float* W = which == 0 ? Wq : (which == 1 ? Wk : Wv);
    const float v = (f < D && n < D) ? W[f * D + n] : 0.0f;
    const int ks = k >> 4, h = (k & 15) >> 3, j = k & 7, nt = n >> 5;
    WdX[((size_t)(ks * NT32 + nt) * 64 + h * 32 + (n & 31)) * 8 + j] = f2bf(v);
  }
}
__global__ __launch_bounds__(256) void pack_qkv_dx_kernel(const float* __restrict__ Wq, const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                          u16* __restrict__ WdX, int rowmajor) {
  pack_qkv_dx_body(Wq, Wk, Wv, WdX, blockIdx.x, gridDim.x, rowmajor);
}

// Every operand packing of ONE encoder (projection operands of the inference / training forward and of the input-gradient GEMM, the pooling
// layer's operand and its transpose) in one launch: blockIdx.y picks the packing, null outputs are skipped.  The training step re-packs after
// every optimiser step: five launches of ~8 us each per encoder (launch + latency of a 300 K-element scatter) were 60 us of the NRMS step.
struct PackEncoderParams {
  const float *Wq, *bq, *Wk, *bk, *Wv, *bv, *Wa, *ba, *qv;
  int qdim;
  u16* Wp; float* bp;          // pack_qkv (register-resident forward, S = 50 encoder)
  u16* Wp32; float* bp32;      // pack_qkv32 (qkv_proj_kernel)
  u16* WdX;                    // pack_qkv_dx (dx_gemm_ring_kernel, or row-major for the persistent stream kernel)
  int dx_rowmajor;
  u16* Wap; float* bap; float* qvp;     // pack_additive
  u16* WaT;                    // pack_additive_t
};
__global__ __launch_bounds__(256) void pack_encoder_kernel(PackEncoderParams p) {
  const int bid = blockIdx.x, nb = gridDim.x;
  switch (blockIdx.y) {
    case 0: if (p.Wp) pack_qkv_body(p.Wq, p.bq, p.Wk, p.bk, p.Wv, p.bv, p.Wp, p.bp, bid, nb); break;
    case 1: if (p.Wp32) pack_qkv32_body(p.Wq, p.bq, p.Wk, p.bk, p.Wv, p.bv, p.Wp32, p.bp32, bid, nb); break;
    case 2: if (p.WdX) pack_qkv_dx_body(p.Wq, p.Wk, p.Wv, p.WdX, bid, nb, p.dx_rowmajor); break;
    case 3: if (p.Wap) pack_additive_body(p.Wa, p.ba, p.qv, p.qdim, p.Wap, p.bap, p.qvp, bid, nb); break;
    default: if (p.WaT) pack_additive_t_body(p.Wa, p.qdim, p.WaT, bid, nb); break;
  }
}

struct DxParams {
  const u16* dqkv;       // [n_tok][960] bf16 (padding columns zero)
  const u16* WdX;        // packed (pack_qkv_dx_kernel)
  u16* dX;               // [n_tok][KP] bf16: columns >= D come out as exact zeros
  int64_t n_tok;
};

// dx_gemm with a RING of chunk buffers.  With two buffers a chunk's copies are issued one chunk of compute (~0.5 us of MFMA work per wave)
// before they are needed -- less than the loaded memory latency, so every chunk waited for its data: 3,600 cycles per chunk and workgroup
// against 640 of MFMA work, the same with 1 x 4 strips or 2 x 2 blocks of the waves (profiles/r03_ab_switches.txt, pass k).  Here: one workgroup of 8 waves
// per CU (256 tokens: wave = 64 tokens x 160 columns, 2 x 5 accumulator tiles), FOUR buffers of 36 KB, copies issued THREE chunks ahead;
// per chunk one counted wait (s_waitcnt vmcnt: this wave's copies of the chunk have landed, two younger chunks stay in flight) and one raw
// s_barrier, which both publishes the chunk and frees the slot the next copies overwrite.  Same chunks, same k order per output: same bits.
struct DxRingGeom {
  static constexpr int NWAVE = 8;
  static constexpr int TOK_WG = 256;
  static constexpr int KC = 2;
  static constexpr int NCH = DXK / (16 * KC);          // 30
  static constexpr int A_BYTES = TOK_WG * KC * 32;     // 16,384
  static constexpr int W_BYTES = KC * NT32 * 1024;     // 20,480
  static constexpr int BUF_BYTES = A_BYTES + W_BYTES;  // 36,864
  static constexpr int NB = 4;                         // ring slots; copies run NB - 1 chunks ahead
  static constexpr int SMEM = NB * BUF_BYTES;          // 147,456 B
  static constexpr int CP = 5;                         // copy instructions per wave and chunk (2 of A, 3 of W)
  static constexpr int OROW = 656;
  static_assert(SMEM <= 163840 && 128 * OROW <= SMEM && CP * (NB - 1) < 64, "ring fits the LDS, the epilogue staging fits the ring, vmcnt range");
};

// DBG (compile time; profiling only, NR_DXR_DEBUG): 1 only the first three chunks are copied, 2 no MFMAs, 4 no dX stores
template <int DBG>
__global__ __launch_bounds__(512, 2) void dx_gemm_ring_kernel(DxParams p) {
  using Gm = DxRingGeom;
  constexpr int dbg = DBG;
  constexpr int NTW = NT32 / 2;                                   // 5 column tiles per wave
  NR_SMEM_DECL(smem);
  const int l = lane_id(), w = wave_id(), h = l >> 5, li = l & 31;
  const int wc = w & 1, wr = w >> 1;                              // column half, token quarter
  const int64_t wg_tok0 = (int64_t)blockIdx.x * Gm::TOK_WG;

  auto piece = [&](int c, int i) {                                // copy i (0 .. CP - 1) of chunk c: two of the token rows, three of the weight fragments
    unsigned char* abuf = smem + (c & (Gm::NB - 1)) * Gm::BUF_BYTES;
    if (i < 2) {
      const int r = w * 32 + i * 16 + (l >> 2), s = l & 3;
      int64_t tok = wg_tok0 + r;
      tok = tok < p.n_tok ? tok : p.n_tok - 1;
      const u16* src = p.dqkv + tok * DXK + c * (16 * Gm::KC) + ((s ^ ((r >> 2) & 3)) * 8);
      NR_GLDS16(src, abuf + (w * 32 + i * 16) * 64);
    } else {
      int f = w + 8 * (i - 2);
      f = f < Gm::KC * NT32 ? f : f - 8;                          // 20 fragments over 24 copies: waves 4 .. 7 repeat one (same bytes, same place)
      NR_GLDS16(p.WdX + (size_t)c * Gm::KC * NT32 * 512 + l * 8 + f * 512, abuf + Gm::A_BYTES + f * 1024);
    }
  };
  auto fetch = [&](int c) {                                       // exactly CP copy instructions per wave
#pragma unroll
    for (int i = 0; i < Gm::CP; ++i) piece(c, i);
  };
  fetch(0);
  fetch(1);
  fetch(2);
  f32x16 acc[2][NTW];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][j][r] = 0.0f;
  const int arow0 = wr * 64 + li, arow1 = arow0 + 32;
  const int sw0 = (arow0 >> 2) & 3, sw1 = (arow1 >> 2) & 3;
  // Fragment sets of one k-step: five weight fragments + two token fragments.  The loop is rotated so that a wave never arrives at a barrier
  // with nothing to issue behind it: k-step 1 of chunk c - 1 is multiplied AFTER the barrier of chunk c, behind the LDS reads of chunk c's
  // k-step 0 -- every block of ten MFMAs (320 cycles) hides the LDS latency of the fragments the next block needs.  (Unrotated, the first
  // reads of a chunk were issued after the barrier and waited for by all waves at once: profiles/r03m, the loop with neither copies nor
  // MFMAs took 189 of the kernel's 433 us, and MFMA time simply added to it.)
  struct Frag { u16x8 wf[NTW], af[2]; };
  auto read_frag = [&](int c, int ks) -> Frag {
    const unsigned char* abuf = smem + (c & (Gm::NB - 1)) * Gm::BUF_BYTES;
    const unsigned char* wbuf = abuf + Gm::A_BYTES + (ks * NT32 + wc * NTW) * 1024 + l * 16;
    Frag f;
    f.af[0] = *(const u16x8*)(abuf + arow0 * 64 + (((ks * 2 + h) ^ sw0) * 16));
    f.af[1] = *(const u16x8*)(abuf + arow1 * 64 + (((ks * 2 + h) ^ sw1) * 16));
#pragma unroll
    for (int j = 0; j < NTW; ++j) f.wf[j] = *(const u16x8*)(wbuf + j * 1024);
    return f;
  };
  // cnext >= 0: the CP copies of chunk cnext are issued one per MFMA pair, not as a burst behind the barrier (a copy instruction holds its
  // wave's issue slot for 60 - 185 cycles, longer the more copies are queued: 428 - 434 vs 440 - 449 us, profiles/r03_ab_switches.txt pass m)
  auto multiply = [&](const Frag& f, int cnext = -1) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      if (!(dbg & 2)) {
        acc[0][j] = mfma_32x32x16_bf16(f.wf[j], f.af[0], acc[0][j]);
        acc[1][j] = mfma_32x32x16_bf16(f.wf[j], f.af[1], acc[1][j]);
      } else {
        acc[0][j][0] += bf2f(f.wf[j][0]) + bf2f(f.af[0][0]);
        acc[1][j][0] += bf2f(f.wf[j][1]) + bf2f(f.af[1][0]);
      }
      if (!(dbg & 1) && cnext >= 0 && cnext < Gm::NCH) { piece(cnext, j); NR_SCHED_BARRIER(); }
    }
  };
  auto arrive = [&](int c) {
    // this wave's copies of chunk c have landed once at most the copies of the chunks after it are outstanding
    if (dbg & 1) NR_WAIT_VMCNT(0);
    else if (c + 2 < Gm::NCH) NR_WAIT_VMCNT(2 * Gm::CP);
    else if (c + 1 < Gm::NCH) NR_WAIT_VMCNT(Gm::CP);
    else NR_WAIT_VMCNT(0);
    NR_WAIT_LGKMCNT(0);                                           // this wave's reads of chunk c - 1 have RETURNED (requested a block of MFMAs ago)
    NR_BARRIER_RAW();                                             // chunk c is complete for everybody; everybody has read chunk c - 1: its slot
  };                                                              // takes the copies of chunk c + 3 (issued by multiply)
  arrive(0);
  if (!(dbg & 1)) fetch(3);
  Frag f0 = read_frag(0, 0);
  Frag f1 = read_frag(0, 1);
  NR_SCHED_BARRIER();
  multiply(f0);
  for (int c = 1; c < Gm::NCH; ++c) {
    // The waits sit BEFORE the next reads are issued: what is outstanding then was requested a whole block of MFMAs ago.  (Left to the
    // compiler, the wait lands in front of the first MFMA that needs the data -- behind the next set's reads, whose latency it then exposes.)
    arrive(c);                                                    // (f1 = k-step 1 of chunk c - 1 is in registers)
    f0 = read_frag(c, 0);
    NR_SCHED_BARRIER();
    multiply(f1, c + 3);
    NR_SCHED_BARRIER();
    NR_WAIT_LGKMCNT(0);                                           // f0 is in registers
    NR_SCHED_BARRIER();
    f1 = read_frag(c, 1);
    NR_SCHED_BARRIER();
    multiply(f0);
  }
  multiply(f1);
  // ---- epilogue: 128 token rows of the workgroup at a time (token tile a of the four token quarters) through LDS ------------------------
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    __syncthreads();                                               // the last chunk (a = 0) / the previous batch (a = 1) has been read
    unsigned char* row = smem + (wr * 32 + li) * Gm::OROW + wc * (NTW * 64);
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(u16x4*)(row + (j * 32 + 8 * q + 4 * h) * 2) = pack4(f32x4{acc[a][j][4 * q], acc[a][j][4 * q + 1], acc[a][j][4 * q + 2], acc[a][j][4 * q + 3]});
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 128 * (KP / 8) / 512; ++i) {               // 10 sixteen-byte pieces per thread
      const int idx = threadIdx.x + 512 * i;
      const int r = idx / (KP / 8), pc = idx - r * (KP / 8);
      const int64_t tok = wg_tok0 + (r >> 5) * 64 + a * 32 + (r & 31);
      if (tok < p.n_tok && !(dbg & 4)) *(u16x8*)(p.dX + tok * KP + pc * 8) = *(const u16x8*)(smem + r * Gm::OROW + pc * 16);
    }
  }
}

}  // namespace nr
