// Backward kernels of the NRMS encoders for gfx950.
//
//  attn_bwd_kernel      per (sequence, head): recomputes P from the saved bf16 Q,K, then
//                       dP = dC V^T, dS = P*(dP - rowsum(P*dP))/sqrt(dk), dQ = dS K, dK = dS^T Q, dV = P^T dC
//                       (autograd of ScaledDotProductAttention, src/model/general/attention/multihead_self.py:15-23).
//                       One wave per pair, all 20x20 / 50x50 products on MFMA; transposed operand copies are
//                       built in wave-private LDS, so there are no workgroup barriers.
//  additive_bwd_kernel  autograd of AdditiveAttention (src/model/general/attention/additive.py:35-52) up to the
//                       pre-activation gradient dpre; the two plain GEMMs that follow (dpre @ Wa, dpre^T @ ctx)
//                       are left to hipBLASLt on the host side.
//  gather_bf16_kernel   materialises the (dropout-masked) bf16 token matrix X for the weight-gradient GEMM.
//  embed_scatter_add    autograd of nn.Embedding(padding_idx=0): scatter-add of token gradients into the table
//                       gradient, skipping row 0 (src/model/NRMS/news_encoder.py:15-20).
//  score_dot_bwd        autograd of DotProductClickPredictor.
#pragma once
#include "nr_common.h"
#include "k_misc.h"
#include "k_mhsa_fwd.h"
#include "k_additive_fwd.h"

namespace nr {

constexpr int LDG = 3 * KP;        // 960: row length of the dQKV gradient matrix (Q | K | V blocks of KP columns)

// ---------------------------------------------------------------------------------------------------------------
template <int S, int WPB>
struct AttnBwdGeom {
  static constexpr int QT = (S + 15) / 16;
  static constexpr int R = QT * 16;                 // padded sequence length
  static constexpr int RS = R + 8;                  // row stride of the dv-major V block in LDS
  static constexpr int DS = 24;                     // row stride of [token][d] matrices (DK=20 padded to 24)
  static constexpr int SP4 = (S + 3) / 4 * 4;
  static constexpr int KS = (R + 31) / 32;          // k-steps over tokens
  static constexpr int KP2 = (QT + 1) / 2;          // pairs of token tiles (one 32-wide MFMA k-step each)
  static constexpr int DTL = (DK + 15) / 16;        // tiles over the head dim
  static constexpr int TD_ELEMS = S * DS;           // Qm, Km, dCm : S rows (tile reads clamp the row index)
  static constexpr int VT_ELEMS = DK * RS + 32;     // Vt [dv][token] (+slack for the 8-wide reads of the last row)
  static constexpr int WAVE_ELEMS = 3 * TD_ELEMS + VT_ELEMS;
  static constexpr int WAVE_BYTES = (WAVE_ELEMS * 2 + 15) / 16 * 16;
  static constexpr int SMEM = WPB * WAVE_BYTES;
  static constexpr int PCS = DK / 4;                // 8-B pieces per 20-wide row
  static constexpr int IT = (S * PCS + 63) / 64;    // row-piece iterations per lane
  static constexpr int VP = SP4 / 4;                // 8-B pieces per dv row of the saved V block
  static constexpr int ITV = (DK * VP + 63) / 64;
  // TILE form (one workgroup per sequence): the sequence's dqkv rows are assembled in LDS and leave as one contiguous run
  static constexpr int TROW = 3 * KP * 2 + 16;      // bytes per staged dqkv row (1,920 + 16: rows shift by 4 banks, 16-byte aligned)
  static constexpr int TILE_BYTES = S * TROW;
  static constexpr int DCT_BYTES = S * KP * 2;      // the sequence's dctx_gemm rows [S][KP], copied global -> LDS one sequence ahead
  static constexpr int SMEM_TILE = SMEM + TILE_BYTES + DCT_BYTES;
  static constexpr int ROUNDS = (H + WPB - 1) / WPB;
  static constexpr int WO_IT = (S * (3 * KP / 8) + WPB * 64 - 1) / (WPB * 64);    // 16-byte pieces per thread of the row write-out
};

struct AttnBwdParams {
  const u16* q_save;     // [n_seq*S][KP]
  const u16* k_save;     // [n_seq*S][KP]
  const u16* vt_save;    // [n_seq][H][DK][SP4]
  const u16* dctx_gemm;  // [n_seq*S][ldc] bf16: dpre @ Wa  (additive backward, GEMM part)
  int ldc;
  const float* attn_w;   // [n_seq][S]  additive attention weights (forward)
  const float* g_out;    // [n_seq][D]  gradient of the pooled vector
  u16* dqkv;             // [n_seq*S][LDG] bf16 (padding columns are never written; host zero-fills once)
  int64_t n_seq;
  const int32_t* key_len;  // optional [n_seq]: the forward's key lengths (MhsaParams::key_len); null: S
  DropCfg dc;            // dropout site 2 (applied to ctx in the forward)
  int xcd_major;         // workgroup -> pair order (xcd_major_block); 0 = plain blockIdx order (A/B knob NR_ATTN_XCD=0)
  int debug;             // profiling only (NR_ATTNB_DEBUG, DBG instantiation): 1 skip the global loads, 4 skip the dqkv stores, 16 skip the arithmetic (8: nothing off)
  unsigned long long* stamps;   // DBG instantiation only (nr_debug_attnb_stamps): [2 workgroups][4 waves][4 sequences][4 rounds][12] cycle-counter stamps
  int hm;                // 1: q_save is the head-major [n_seq][H][3][S][DK] buffer of qkv_proj_kernel (k_proj.h; S = 20): Q, K, V of a pair, each
                         // [token][d] row-major, are 2,400 contiguous bytes (k_save / vt_save unused)
};

__device__ __forceinline__ u16x8 ld8(const u16* p) { return cat8(*(const u16x4*)p, *(const u16x4*)(p + 4)); }

// Register image of one (sequence, head) pair's inputs: loaded one pair ahead so the global-memory latency of
// pair i+1 hides behind the MFMA work of pair i.
template <int IT, int ITV>
struct AttnBwdRegs {
  u16x4 q[IT], k[IT], dg[IT], v[ITV];
  f32x4 go[IT];
  float wt[IT];
};

// Layout algebra used below (16x16x32 MFMA, lane = (g, li)):
//   AL(M) : fragment read from a row-major LDS matrix M[row][k]: lane holds M[li][k-slots]  -> A or B operand
//   CL(X) : accumulator tile of X[a][b]: lane holds X[4g+r][li].  A CL tile packed to bf16 is, without moving any
//           data, (1) a B operand "X with k = a" and (2) an A operand "X^T with k = a" (slot (g, r) <-> a = 4g+r,
//           a second tile fills slots j >= 4).  So everything that contracts over the ROW index of a CL tile is free.
//   Transposes LDS->CL are done by the matrix core itself: CL(M) = AL(M) x Identity.
// No scattered LDS writes, one wave barrier per pair.
#ifndef NR_ATTN_OCC
#define NR_ATTN_OCC 1      // minimum workgroups per CU the register allocation must allow (tuning knob)
#endif
// TILE = true: one workgroup per sequence at a time, wave w takes heads w, w + WPB, ... (ROUNDS rounds; a wave without a head in the last
// round only keeps the barriers); the pairs' outputs go to an LDS tile [S][3 KP] and the sequence's rows are written as ONE contiguous
// run of S x 1,920 bytes; the sequence's dctx_gemm rows arrive the same way, as one contiguous 12.8 KB global -> LDS copy issued a
// sequence ahead (when ldc == KP).  The plain form stores every output tile as 8-byte pieces of 16 different 1,920-byte rows -- 12 such
// instructions per pair -- and fetches dctx as 40-byte pieces of 20 rows; its phase decomposition (profiles/r03d_attn_bwd_phases.txt:
// 940 us, 598 without the loads, 596 without the stores, 462 without both) shows it issuing memory instructions for half of its time.
template <int S, int WPB, bool DBG = false, bool TILE = false>
__global__ __launch_bounds__(WPB * 64, NR_ATTN_OCC) void attn_bwd_kernel(AttnBwdParams p) {
  using Gm = AttnBwdGeom<S, WPB>;
  const int dbg = DBG ? p.debug : 0;
  p.dc = drop_resolve(p.dc);
  NR_SMEM_DECL(smem);
  const int l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  u16* base = (u16*)(smem + w * Gm::WAVE_BYTES);
  u16* Qm = base;
  u16* Km = Qm + Gm::TD_ELEMS;
  u16* dCm = Km + Gm::TD_ELEMS;
  u16* Vt = dCm + Gm::TD_ELEMS;
  unsigned char* const tile = smem + Gm::SMEM;        // TILE form only
  unsigned char* const dctile = tile + Gm::TILE_BYTES;
  // dctx rows through LDS: needs whole 16-byte pieces of contiguous rows
  const bool staged = TILE && p.ldc == KP && (((uintptr_t)p.dctx_gemm) & 15) == 0;
  auto stage_dctx = [&](int64_t sq) {                 // async copy of sequence sq's S x KP block; visible after the next workgroup barrier
    const u16* src = p.dctx_gemm + sq * S * KP;
    for (int b = w; b * 64 < S * KP / 8; b += WPB)
      if (b * 64 + l < S * KP / 8 && !(dbg & 1)) NR_GLDS16(src + (b * 64 + l) * 8, dctile + b * 1024);
  };

  const int64_t n_pairs = p.n_seq * H;
  const int64_t stride = (int64_t)gridDim.x * WPB;
  int64_t pair = TILE ? (int64_t)blockIdx.x * H + w
                      : (int64_t)(p.xcd_major ? xcd_major_block(blockIdx.x, gridDim.x) : (int)blockIdx.x) * WPB + w;
  if (TILE) {             // K padding of the staged rows (cols D .. KP - 1 of the three blocks) is zero and never overwritten
    constexpr int PADQ = (KP - D) / 4;
    for (int i = threadIdx.x; i < S * 3 * PADQ; i += WPB * 64) {
      const int r = i / (3 * PADQ), c = i - r * (3 * PADQ);
      *(u16x4*)(tile + r * Gm::TROW + ((c / PADQ) * KP + D + (c % PADQ) * 4) * 2) = u16x4{0, 0, 0, 0};
    }
  }

  AttnBwdRegs<Gm::IT, Gm::ITV> rg;
  // (seq, hd) are passed in, not derived from the pair index: the TILE form knows both, and a 64-bit division by H per call -- two calls per
  // pair -- was most of the 258 scalar instructions per pair that profiles/r03_pmc_sq_attn_bwd_hm.txt shows
  auto load_regs = [&](int64_t seq, int hd) {
    if (dbg & 1) return;
    const int64_t pr = seq * H + hd;
    const int64_t tok0 = seq * S;
    // (wave-uniform 64-bit base) + (32-bit lane offset that does not depend on the pair): the per-lane 64-bit multiply-adds of
    // `(tok0 + r) * KP` were a quarter-rate instruction per address
    // head-major saves: row r of the pair's Q block at + r * DK, K one block further -- same piece index i, a different row stride
    const int rs = p.hm ? DK : KP;
    const u16* qb = p.hm ? p.q_save + pr * (3 * S * DK) : p.q_save + tok0 * KP + hd * DK;
    const u16* kb = p.hm ? qb + S * DK : p.k_save + tok0 * KP + hd * DK;
    const u16* gb = p.dctx_gemm + tok0 * p.ldc + hd * DK;
    const float* gob = p.g_out + seq * D + hd * DK;
    const float* wb = p.attn_w + tok0;
#pragma unroll
    for (int it = 0; it < Gm::IT; ++it) {
      const int i = it * 64 + l;
      if (i < S * Gm::PCS) {
        const int r = i / Gm::PCS, c = (i - r * Gm::PCS) * 4;
        rg.q[it] = *(const u16x4*)(qb + (r * rs + c));
        rg.k[it] = *(const u16x4*)(kb + (r * rs + c));
        if (!staged) rg.dg[it] = *(const u16x4*)(gb + (r * p.ldc + c));
        rg.go[it] = *(const f32x4*)(gob + c);
        rg.wt[it] = wb[r];
      }
    }
    if (p.hm) {                        // V [token][d] like Q and K: the same piece index
      const u16* vb = qb + 2 * S * DK;
#pragma unroll
      for (int it = 0; it < Gm::IT && it < Gm::ITV; ++it) {
        const int i = it * 64 + l;
        if (i < S * Gm::PCS) rg.v[it] = *(const u16x4*)(vb + i * 4);
      }
    } else {
      const u16* vblk = p.vt_save + (seq * H + hd) * DK * Gm::SP4;
#pragma unroll
      for (int it = 0; it < Gm::ITV; ++it) {
        const int i = it * 64 + l;
        if (i < DK * Gm::VP) rg.v[it] = *(const u16x4*)(vblk + i * 4);     // block is dense: [dv][SP4]
      }
    }
  };
  // registers -> wave-private LDS, all 8-B row-major stores; dC assembled with its direct term and dropout
  auto store_lds = [&](int64_t seq, int hd) {
    const int64_t tok0 = seq * S;
#pragma unroll
    for (int it = 0; it < Gm::IT; ++it) {
      const int i = it * 64 + l;
      if (i < S * Gm::PCS) {
        const int r = i / Gm::PCS, c = (i - r * Gm::PCS) * 4;
        *(u16x4*)(Qm + r * Gm::DS + c) = rg.q[it];
        *(u16x4*)(Km + r * Gm::DS + c) = rg.k[it];
        f32x4 dc4;
        const u16x4 dg = staged ? *(const u16x4*)(dctile + (r * KP + hd * DK + c) * 2) : rg.dg[it];
#pragma unroll
        for (int j = 0; j < 4; ++j) dc4[j] = bf2f(dg[j]) + rg.wt[it] * rg.go[it][j];
        if (p.dc.enabled) {
          dc4 = dc4 * drop_mul4(p.dc, 2u, (uint64_t)(tok0 + r) * D4 + ((hd * DK + c) >> 2));
        }
        *(u16x4*)(dCm + r * Gm::DS + c) = pack4(dc4);
        if (p.hm && it < Gm::ITV) *(u16x4*)(Vt + r * Gm::DS + c) = rg.v[it];        // row-major V [token][DS] in the V^T region
      }
    }
    if (p.hm) return;
#pragma unroll
    for (int it = 0; it < Gm::ITV; ++it) {
      const int i = it * 64 + l;
      if (i < DK * Gm::VP) {
        const int dv = i / Gm::VP, t = (i - dv * Gm::VP) * 4;
        u16x4 v = rg.v[it];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (t + j < S) ? v[j] : (u16)0;     // tokens >= S of the block hold bias junk
        *(u16x4*)(Vt + dv * Gm::RS + t) = v;
      }
    }
  };

  if (pair < n_pairs) load_regs(pair / H, (int)(pair % H));
  // zero the wave-private scratch ONCE: padding columns must be finite (zero); real positions are rewritten per pair
  for (int i = l; i < Gm::WAVE_BYTES / 16; i += 64) *(u16x8*)(base + i * 8) = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  wave_barrier();

  const float inv_sqrt_dk = 1.0f / sqrtf((float)DK);
  const float c2 = LOG2E * inv_sqrt_dk, clamp2 = EXP_CLAMP * LOG2E;       // exp(s / sqrt(dk)) = exp2(s * c2), as in the forward kernel
  const u16 ONE = 0x3F80;
  const u16x4 Z4 = u16x4{0, 0, 0, 0};
  // AL fragment over the head dim from a [token][DS] matrix: slots d = 8g + j (d >= DK are zero)
  auto frag_d = [&](const u16* M, int row) -> u16x8 {
    row = row < S ? row : S - 1;
    const u16* q_ = M + row * Gm::DS + 8 * g;
    u16x4 lo = (8 * g < DK) ? *(const u16x4*)q_ : Z4;
    u16x4 hi = (8 * g + 4 < DK) ? *(const u16x4*)(q_ + 4) : Z4;
    return cat8(lo, hi);
  };
  // AL fragment of dC with the CL-compatible slot order over dv: slots j<4 -> dv = 4g+j, j>=4 -> dv = 16+4g+(j-4)
  auto frag_dc_perm = [&](int row) -> u16x8 {
    row = row < S ? row : S - 1;
    const u16* q_ = dCm + row * Gm::DS;
    u16x4 lo = *(const u16x4*)(q_ + 4 * g);                                  // 4g+3 <= 15 < DK
    u16x4 hi = (16 + 4 * g < DK) ? *(const u16x4*)(q_ + 16 + 4 * g) : Z4;
    return cat8(lo, hi);
  };
  // identity B operands: ident[t][k][n] = (k == 16 t + n) with natural slots k = 8g + j
  u16x8 ident[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) ident[t][j] = (8 * g + j == 16 * t + li) ? ONE : (u16)0;
  // identity on the CL k-slots: a packed CL tile used as an A operand carries row a = 4g + r in k-slot (g, r), i.e. k = 8g + r
  u16x8 identc;
#pragma unroll
  for (int j = 0; j < 8; ++j) identc[j] = (j < 4 && 4 * g + j == li) ? ONE : (u16)0;

  // TILE iteration state: (sequence, round); every wave of the workgroup runs the same iterations and meets the same barriers
  int64_t seq_t = blockIdx.x;
  int rnd = 0;
  int it_t = 0;                             // (timeline of the first sequences of a few workgroups: tools/attnb_timeline.py)
  auto stamp = [&](int k) {
    if (DBG && TILE && p.stamps != nullptr && l == 0 && blockIdx.x < 2 && it_t < 4)
      p.stamps[((((size_t)blockIdx.x * WPB + w) * 4 + it_t) * Gm::ROUNDS + rnd) * 12 + k] = __builtin_readcyclecounter();
  };
  if (TILE) {
    if (staged && seq_t < p.n_seq) stage_dctx(seq_t);
    __syncthreads();
  }
  while (TILE ? seq_t < p.n_seq : pair < n_pairs) {
    const bool act = !TILE || w + rnd * WPB < H;
    if (TILE) pair = seq_t * H + w + rnd * WPB;
    const int64_t seq = TILE ? seq_t : pair / H;
    const int hd = TILE ? w + rnd * WPB : (int)(pair - seq * H);
    const int64_t tok0 = seq * S;
    const int klen = p.key_len != nullptr ? uniform(clamp_len(p.key_len[seq], S)) : S;
    stamp(0);
    if (act) store_lds(seq, hd);
    stamp(1);
    if (TILE && staged && rnd == Gm::ROUNDS - 1) {          // every wave has taken its last pieces of this sequence's dctx rows
      __syncthreads();
      if (seq_t + gridDim.x < p.n_seq) stage_dctx(seq_t + gridDim.x);
    }
    // TILE: the wave's next head of this sequence, then its first head of the workgroup's next sequence
    const int64_t next = !TILE ? pair + stride : (hd + WPB < H ? pair + WPB : (seq + gridDim.x) * H + w);
    if (act) {
    if (next < n_pairs) {                         // prefetch the next pair while this one is computed
      if (TILE) load_regs(hd + WPB < H ? seq : seq + gridDim.x, hd + WPB < H ? hd + WPB : w);
      else load_regs(next / H, (int)(next % H));
    }
    wave_barrier();
    stamp(2);

    if (!(dbg & 16)) {     // (debug bit 16: the I/O skeleton alone -- loads, LDS staging, barriers and the write-out, no arithmetic)
    // ---- operand preparation ----------------------------------------------------------------------------------------
    u16x8 kf[Gm::QT], qf[Gm::QT], cperm[Gm::QT];
    u16x4 kcl[Gm::QT][Gm::DTL], qcl[Gm::QT][Gm::DTL], ccl[Gm::QT][Gm::DTL];   // CL(K), CL(Q), CL(dC) tiles [token tile][d tile]
#pragma unroll
    for (int t = 0; t < Gm::QT; ++t) {
      kf[t] = frag_d(Km, t * 16 + li);
      qf[t] = frag_d(Qm, t * 16 + li);
      // free k-slot DK = key-padding mask (see the forward kernel): scores of key rows >= S come out at -29952 -> exp2 = 0.
      // The extra "feature" only reaches output rows d = DK, which are never stored.
      if (8 * g + 4 == DK) { qf[t][4] = ONE; kf[t][4] = (t * 16 + li < klen) ? (u16)0 : BF16_NEG_BIG; }
      u16x8 cf = frag_d(dCm, t * 16 + li);
      cperm[t] = frag_dc_perm(t * 16 + li);
#pragma unroll
      for (int dt = 0; dt < Gm::DTL; ++dt) {
        kcl[t][dt] = pack4(mfma_16x16x32_bf16(kf[t], ident[dt], f32x4{0.f, 0.f, 0.f, 0.f}));
        qcl[t][dt] = pack4(mfma_16x16x32_bf16(qf[t], ident[dt], f32x4{0.f, 0.f, 0.f, 0.f}));
        ccl[t][dt] = pack4(mfma_16x16x32_bf16(cf, ident[dt], f32x4{0.f, 0.f, 0.f, 0.f}));
      }
    }
    // V as an A operand "V[key][k = dv]" per key tile, k-slots in the CL-compatible order of cperm (j < 4: dv = 4 g + j, j >= 4: dv = 16 + 4 g + j - 4).
    // Row-major V (head-major saves): a plain fragment read.  dv-major V^T blocks (vt_save): CL(Vt)[dv][key] tiles for dv tiles 0, 1 built on
    // the matrix core.
    u16x8 va[Gm::QT];
    if (p.hm) {
#pragma unroll
      for (int kt = 0; kt < Gm::QT; ++kt) {
        int row = kt * 16 + li;
        row = row < S ? row : S - 1;           // clamped duplicates: these keys carry exact-zero probabilities
        const u16* v_ = Vt + row * Gm::DS;
        va[kt] = cat8(*(const u16x4*)(v_ + 4 * g), (16 + 4 * g < DK) ? *(const u16x4*)(v_ + 16 + 4 * g) : Z4);
      }
    } else
#pragma unroll
    for (int kt = 0; kt < Gm::QT; ++kt) {
      u16x4 part[Gm::DTL];
#pragma unroll
      for (int dt = 0; dt < Gm::DTL; ++dt) {
        int dvrow = dt * 16 + li;
        dvrow = dvrow < DK ? dvrow : DK - 1;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < Gm::KS; ++ks) {
          // B = identity selecting token kt*16 + n out of the 32 tokens of k-step ks
          u16x8 idt = (kt / 2 == ks) ? ident[kt & 1] : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
          acc = mfma_16x16x32_bf16(ld8(Vt + dvrow * Gm::RS + ks * 32 + 8 * g), idt, acc);
        }
        // rows dv >= DK of the second dv tile are clamped duplicates: zero them (they sit in k-slots of the dP products)
        if (dt * 16 + 4 * g >= DK) acc = f32x4{0.f, 0.f, 0.f, 0.f};
        part[dt] = pack4(acc);
      }
      va[kt] = cat8(part[0], Gm::DTL > 1 ? part[Gm::DTL - 1] : Z4);
    }

    stamp(3);
    u16x4 dsT[Gm::QT][Gm::QT];    // CL(dS^T)  [key tile][query tile]
    u16x4 dsN[Gm::QT][Gm::QT];    // CL(dS)    [query tile][key tile]
    u16x4 pN[Gm::QT][Gm::QT];     // CL(P)     [query tile][key tile]
#pragma unroll
    for (int qt = 0; qt < Gm::QT; ++qt) {
      // ---- transposed pass: P^T (col = query li, rows = keys), row sums, dP^T, row dots ---------------------------
      f32x4 pT[Gm::QT];
      float sum = 0.0f;
#pragma unroll
      for (int kt = 0; kt < Gm::QT; ++kt) {
        pT[kt] = mfma_16x16x32_bf16(kf[kt], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = fast_exp2(fminf(pT[kt][r] * c2, clamp2));     // same formula as the forward kernel
          pT[kt][r] = e;
          sum += e;
        }
      }
      sum = sum_rows4(sum);
      const float rden = fast_rcp(sum + 1e-8f);
      float dot = 0.0f;
      f32x4 dPT[Gm::QT];
#pragma unroll
      for (int kt = 0; kt < Gm::QT; ++kt) {
        pT[kt] = pT[kt] * rden;
        dPT[kt] = mfma_16x16x32_bf16(va[kt], cperm[qt], f32x4{0.f, 0.f, 0.f, 0.f});   // dP^T[key][q] = sum_dv V dC
#pragma unroll
        for (int r = 0; r < 4; ++r) dot += pT[kt][r] * dPT[kt][r];
      }
      dot = sum_rows4(dot);
#pragma unroll
      for (int kt = 0; kt < Gm::QT; ++kt) {
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = pT[kt][r] * (dPT[kt][r] - dot) * inv_sqrt_dk;
        dsT[kt][qt] = pack4(ds);
      }
      // ---- the same tiles with rows = queries, CL(P) and CL(dS), for the dK / dV products: TRANSPOSED BY THE MATRIX CORE from the packed
      // transposed tiles (A = the packed CL(X^T) tile read as "X with k = key", B = identity on the CL k-slots: one MFMA per tile, exact
      // for bf16 inputs) instead of recomputed in the other orientation -- that second pass cost 16 more v_exp_f32 and ~160 more vector
      // instructions per pair plus 8 cross-lane fetches of the row statistics (A/B on one MI355X: 963 / 973 -> 894 / 949 us per launch).  Padded queries are columns here: they are zeroed first
      // (they must not reach the contractions over q; padded keys are already exact zeros).
      const bool qok = qt * 16 + li < S;
#pragma unroll
      for (int kt = 0; kt < Gm::QT; ++kt) {
        pN[kt][qt] = qok ? pack4(pT[kt]) : Z4;       // still CL(P^T) [key tile][query tile] here, transposed in place below
        if (!qok) dsT[kt][qt] = Z4;
      }
      NR_SCHED_BARRIER();
    }
    stamp(4);
    {
      u16x4 tp[Gm::QT][Gm::QT];
#pragma unroll
      for (int qt = 0; qt < Gm::QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < Gm::QT; ++kt) {
          tp[qt][kt] = pack4(mfma_16x16x32_bf16(cat8(pN[kt][qt], Z4), identc, f32x4{0.f, 0.f, 0.f, 0.f}));
          dsN[qt][kt] = pack4(mfma_16x16x32_bf16(cat8(dsT[kt][qt], Z4), identc, f32x4{0.f, 0.f, 0.f, 0.f}));
        }
#pragma unroll
      for (int qt = 0; qt < Gm::QT; ++qt)
#pragma unroll
        for (int kt = 0; kt < Gm::QT; ++kt) pN[qt][kt] = tp[qt][kt];
      NR_SCHED_BARRIER();
    }

    stamp(5);
    // ---- output products: every A/B operand below is a packed CL tile already in registers --------------------------
    // dQ^T[d][q]   = sum_key K^T[d][key] dS^T[key][q] : A = CL(K)  (k = key), B = CL(dS^T) (k = key)
    // dK^T[d][key] = sum_q   Q^T[d][q]   dS[q][key]   : A = CL(Q)  (k = q),   B = CL(dS)   (k = q)
    // dV^T[dv][key]= sum_q   dC^T[dv][q] P[q][key]    : A = CL(dC) (k = q),   B = CL(P)    (k = q)
#pragma unroll
    for (int dt = 0; dt < Gm::DTL; ++dt) {
      const int d0 = dt * 16 + 4 * g;     // CL(K)[key][d]: as A operand its rows are d = li of tile dt
#pragma unroll
      for (int ot = 0; ot < Gm::QT; ++ot) {       // output token tile (queries for dQ, keys for dK / dV)
        f32x4 aq = f32x4{0.f, 0.f, 0.f, 0.f}, ak = aq, av = aq;
#pragma unroll
        for (int kp = 0; kp < Gm::KP2; ++kp) {
          const int t0 = 2 * kp, t1 = (2 * kp + 1 < Gm::QT) ? 2 * kp + 1 : 0;
          const bool has1 = 2 * kp + 1 < Gm::QT;
          aq = mfma_16x16x32_bf16(cat8(kcl[t0][dt], has1 ? kcl[t1][dt] : Z4), cat8(dsT[t0][ot], has1 ? dsT[t1][ot] : Z4), aq);
          ak = mfma_16x16x32_bf16(cat8(qcl[t0][dt], has1 ? qcl[t1][dt] : Z4), cat8(dsN[t0][ot], has1 ? dsN[t1][ot] : Z4), ak);
          av = mfma_16x16x32_bf16(cat8(ccl[t0][dt], has1 ? ccl[t1][dt] : Z4), cat8(pN[t0][ot], has1 ? pN[t1][ot] : Z4), av);
        }
        // A operand rows: i = li -> d = dt*16 + li; output CL: rows d = dt*16 + 4g + r, col = token ot*16 + li
        const int tok = ot * 16 + li;
        if (TILE) {
          if (d0 < DK && tok < S) {
            u16* dst = (u16*)(tile + tok * Gm::TROW) + hd * DK + d0;
            *(u16x4*)dst = pack4(aq);
            *(u16x4*)(dst + KP) = pack4(ak);
            *(u16x4*)(dst + 2 * KP) = pack4(av);
          }
        } else if (d0 < DK && tok < S && !(dbg & 4)) {
          u16* dst = (p.dqkv + tok0 * LDG + hd * DK) + (tok * LDG + d0);
          *(u16x4*)dst = pack4(aq);
          *(u16x4*)(dst + KP) = pack4(ak);
          *(u16x4*)(dst + 2 * KP) = pack4(av);
        }
      }
    }
    }                      // !(dbg & 16)
    wave_barrier();        // all LDS reads of this pair done before the next pair's stores
    stamp(6);
    }                      // act
    if (TILE && rnd == Gm::ROUNDS - 1) {         // the sequence is complete in every wave after this barrier: S x 1,920 contiguous bytes leave
      stamp(7);
      __syncthreads();
      stamp(8);
      if (!(dbg & 4)) {
        u16* dst = p.dqkv + tok0 * LDG;
#pragma unroll
        for (int i = 0; i < Gm::WO_IT; ++i) {
          const int idx = threadIdx.x + i * WPB * 64;
          const int r = idx / (LDG / 8), pc = idx - r * (LDG / 8);
          if (idx < S * (LDG / 8)) *(u16x8*)(dst + idx * 8) = *(const u16x8*)(tile + r * Gm::TROW + pc * 16);
        }
      }
      stamp(9);
      __syncthreads();
      stamp(10);
    }
    if (TILE) {
      if (++rnd == Gm::ROUNDS) { rnd = 0; seq_t += gridDim.x; ++it_t; }
    } else {
      pair = next;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------

template <int S, int NSEQ, int NW = 4>
__global__ __launch_bounds__(NW * 64) void additive_bwd_kernel(AdditiveBwdParams p) {
  using Gm = AddGeom<S, NSEQ, NW>;
  constexpr int WG = NW * 64;          // shadows nr::WG: 4 or 8 waves
  NR_SMEM_DECL(smem);
  u16* Xs = (u16*)smem;
  float* dqp = (float*)(smem + Gm::X_BYTES);                 // [NW][QP]
  float* dsv = (float*)(smem + Gm::X_BYTES + Gm::DQ_FLOATS * 4);    // [ROWS]
  float* gl = (float*)(smem + Gm::X_BYTES + Gm::DQ_FLOATS * 4 + Gm::ROWS * 4);     // [NSEQ][D] g_out rows of this workgroup
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  const int64_t seq0 = (int64_t)blockIdx.x * NSEQ;
  const int64_t tok0 = seq0 * S, tok_total = p.n_seq * S;

  // g_out rows -> LDS, coalesced, in the same round trip as the ctx tile: the dw pass below reads each row 25 times per token
  // (as global loads that was 5 dependent L2 round trips per lane at the head of every workgroup)
  for (int i = tid; i < NSEQ * D4; i += WG) {
    const int seq = i / D4, c = i - seq * D4;
    *(f32x4*)(gl + seq * D + c * 4) = seq0 + seq < p.n_seq ? *(const f32x4*)(p.g_out + (seq0 + seq) * D + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // forward attention weights of the sequences this wave reduces below: requested now, with the tile
  constexpr int SPW = (NSEQ + NW - 1) / NW;
  float wt_pre[SPW];
#pragma unroll
  for (int k = 0; k < SPW; ++k) {
    const int seq = w + k * NW;
    wt_pre[k] = (seq < NSEQ && l < S && seq0 + seq < p.n_seq) ? p.attn_w[(seq0 + seq) * S + l] : 0.0f;
  }
  constexpr int PCS = XS / 8;
  for (int i = tid; i < Gm::ROWS * PCS; i += WG) {
    int r = i / PCS, c = i - r * PCS;
    u16x8 v = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (c < KP / 8 && r < Gm::TOK && tok0 + r < tok_total) v = *(const u16x8*)(p.ctx + (tok0 + r) * KP + c * 8);
    *(u16x8*)(Xs + r * XS + c * 8) = v;
  }
  for (int i = tid; i < Gm::DQ_FLOATS; i += WG) dqp[i] = 0.0f;
  for (int i = tid; i < Gm::ROWS; i += WG) dsv[i] = 0.0f;
  __syncthreads();

  // ---- dw[tok] = g_out . x[tok];  ds = w * (dw - sum_s w dw)  (softmax backward) -------------------------------
  // three lanes per token, 25 quads (100 columns) each; partials combined through LDS (dqp is still free here)
  {
    float* dwp = dqp;                                   // [3][ROWS] scratch (DQ_FLOATS >= 3*ROWS)
    for (int idx = tid; idx < Gm::TOK * 3; idx += WG) {
      const int tok = idx / 3, part = idx - tok * 3;
      const int seq = tok / S;
      float a = 0.0f;
      if (seq0 + seq < p.n_seq) {
        const float* go = gl + seq * D + part * 100;
        const u16* xr = Xs + tok * XS + part * 100;
#pragma unroll 5
        for (int c = 0; c < 25; ++c) {
          f32x4 g4 = *(const f32x4*)(go + c * 4);
          u16x4 x4 = *(const u16x4*)(xr + c * 4);
          a += g4[0] * bf2f(x4[0]) + g4[1] * bf2f(x4[1]) + g4[2] * bf2f(x4[2]) + g4[3] * bf2f(x4[3]);
        }
      }
      dwp[part * Gm::ROWS + tok] = a;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
      const int seq = w + k * NW;
      if (seq >= NSEQ) break;
      const bool live = l < S && seq0 + seq < p.n_seq;
      const int r = seq * S + l;
      const float mydw = live ? dwp[r] + dwp[Gm::ROWS + r] + dwp[2 * Gm::ROWS + r] : 0.0f;
      const float wt = wt_pre[k];
      const float tot = wave_sum(wt * mydw);
      if (l < S) dsv[r] = wt * (mydw - tot);
    }
    __syncthreads();
    for (int i = tid; i < Gm::DQ_FLOATS; i += WG) dqp[i] = 0.0f;
  }
  __syncthreads();

  // ---- recompute t = tanh(x Wa^T + ba); dpre = ds * qv * (1 - t^2); dq += ds * t ---------------------------------
  const int w_eff = (w + (int)blockIdx.x) % NW;
  for (int cg = 0; cg < (Gm::NTQ + 1) / 2; ++cg) {
    int G, mb, me;
    unit_range(Gm::NTQ, Gm::MT, w_eff, NW, cg, G, mb, me);
    if (mb >= me) continue;
    f32x4 dqa[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // query-vector rows of this column pair, loaded once per pair with the weight fragments (see additive_fwd_kernel)
    const int wrq0 = (2 * cg) * 16, wrq1 = G == 2 ? wrq0 + 16 : wrq0;
    const f32x4 qv2[2] = {*(const f32x4*)(p.qvp + wrq0 + 4 * g), *(const f32x4*)(p.qvp + wrq1 + 4 * g)};
    auto epi = [&](int j, int wr, int m, f32x4 acc) {   // acc = x.Wa[n] + ba[n] (bias = accumulator init)
      const f32x4 q4 = qv2[j];
      const int r_ = m * 16 + li;
      const float ds = dsv[r_];
      f32x4 dp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = fast_tanh(acc[r]);
        dp[r] = ds * q4[r] * (1.0f - t * t);
        dqa[j][r] += ds * t;
      }
      if (r_ < Gm::TOK && tok0 + r_ < tok_total) *(u16x4*)(p.dpre + (tok0 + r_) * QP + wr + 4 * g) = pack4(dp);
    };
    int wr0 = (2 * cg) * 16, wr1 = (2 * cg + 1) * 16;
    if (G == 2) {
      int wrow[2] = {wr0, wr1};
      const f32x4 binit[2] = {*(const f32x4*)(p.bap + wr0 + 4 * g), *(const f32x4*)(p.bap + wr1 + 4 * g)};
      proj_block<2, true>(p.Wap, wrow, Xs, mb, me, binit, [&](int j, int m, f32x4 acc) { epi(j, wrow[j], m, acc); });
    } else {
      int wrow[1] = {wr0};
      const f32x4 binit[1] = {*(const f32x4*)(p.bap + wr0 + 4 * g)};
      proj_block<1, true>(p.Wap, wrow, Xs, mb, me, binit, [&](int j, int m, f32x4 acc) { epi(0, wrow[0], m, acc); });
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j < G) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = dqa[j][r];
          v += shfl_xor(v, 1); v += shfl_xor(v, 2); v += shfl_xor(v, 4); v += shfl_xor(v, 8);
          if (li == 0) dqp[w * QP + (j == 0 ? wr0 : wr1) + 4 * g + r] += v;
        }
      }
    }
  }
  __syncthreads();
  for (int n = tid; n < QP; n += WG)
  {
    float a = 0.0f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) a += dqp[ww * QP + n];
    p.dq_part[(int64_t)blockIdx.x * QP + n] = a;
  }

  // ---- fused input-gradient product dctx[tok][:] = dpre[tok][:] @ Wa (the GEMM part of d ctx; the direct term attn_w (x) g_out is
  //      added by the consumer).  The workgroup's dpre tile was just written to global memory by its own waves (L2-hot): it is
  //      re-read into the LDS region of the no longer needed ctx tile as the MFMA B operand; A = rows of Wa^T straight from L2.
  if (p.dctx != nullptr) {
    constexpr int PS2 = QKP + 8;            // 232: conflict-free b128 fragment reads
    constexpr int PC2 = PS2 / 8;            // 29 16-B pieces per LDS row
    constexpr int KS2 = QKP / 32;           // 7 k-steps
    u16* Ps = Xs;
    // (the barrier above is a workgroup-scope release/acquire; all waves of the workgroup share this CU's L1, so the re-read sees
    //  their stores -- an agent-scope fence here costs an L2 writeback per workgroup and made the kernel 5x slower)
    for (int i = tid; i < Gm::ROWS * PC2; i += WG) {
      const int r = i / PC2, c = i - r * PC2;
      u16x8 v = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (c < QP / 8 && r < Gm::TOK && tok0 + r < tok_total) v = *(const u16x8*)(p.dpre + (tok0 + r) * QP + c * 8);
      *(u16x8*)(Ps + r * PS2 + c * 8) = v;
    }
    __syncthreads();
    constexpr int NTD = (D + 15) / 16;      // 19 output column tiles
    for (int cg = 0; cg < (NTD + 1) / 2; ++cg) {
      int G, mb, me;
      unit_range(NTD, Gm::MT, w_eff, NW, cg, G, mb, me);
      if (mb >= me) continue;
      const int wr0 = (2 * cg) * 16, wr1 = G == 2 ? wr0 + 16 : wr0;
      u16x8 wf[2][KS2];
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        wf[0][ks] = *(const u16x8*)(p.WaT + (size_t)wr0 * QKP + ks * 512 + l * 8);      // tile order
        wf[1][ks] = *(const u16x8*)(p.WaT + (size_t)wr1 * QKP + ks * 512 + l * 8);
      }
      for (int m = mb; m < me; ++m) {
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
        const u16* xp = Ps + (m * 16 + li) * PS2 + g * 4;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
          // pair-permuted contraction index of WaT (pack_additive_t_kernel): query rows 4g..4g+3 of tiles 2 ks and 2 ks + 1
          const u16x8 xf = cat8(*(const u16x4*)(xp + ks * 32), *(const u16x4*)(xp + ks * 32 + 16));
          a0 = mfma_16x16x32_bf16(wf[0][ks], xf, a0);
          if (G == 2) a1 = mfma_16x16x32_bf16(wf[1][ks], xf, a1);
        }
        const int r_ = m * 16 + li;
        if (r_ < Gm::TOK && tok0 + r_ < tok_total) {
          if (wr0 + 4 * g < D) *(u16x4*)(p.dctx + (tok0 + r_) * KP + wr0 + 4 * g) = pack4(a0);
          if (G == 2 && wr1 + 4 * g < D) *(u16x4*)(p.dctx + (tok0 + r_) * KP + wr1 + 4 * g) = pack4(a1);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// X materialisation for the weight-gradient GEMM: Xb[tok][0:D] = dropout1(table[ids[tok]]) (or dense x), Xb[tok][D] = 1,
// Xb[tok][D+1:KP] = 0.  One 16-B piece (8 bf16) per lane.
__global__ __launch_bounds__(256) void gather_bf16_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                          int64_t num_rows, const float* __restrict__ x_dense,
                                                          u16* __restrict__ Xb, int64_t n_tokens, DropCfg dc) {
  dc = drop_resolve(dc);          // exactly once: applied twice, the step counter's XOR into k0 cancels and k1 advances by two steps
  constexpr int PC = KP / 4;   // 80 quads per row
  const int64_t total = n_tokens * PC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tok = i / PC;
    const int c = (int)(i - tok * PC);
    u16x4 o = u16x4{0, 0, 0, 0};
    if (c < D4) {
      const float* src;
      if (ids != nullptr) {
        int64_t id = ids[tok];
        id = id < 0 ? 0 : (id >= num_rows ? num_rows - 1 : id);
        src = table + (id * D4 + c) * 4;
      } else {
        src = x_dense + (tok * D4 + c) * 4;
      }
      f32x4 x = *(const f32x4*)src;
      if (dc.enabled) {
        x = x * drop_mul4(dc, 1u, (uint64_t)tok * D4 + c);
      }
      o = pack4(x);
    } else if (c == D4) {
      o[0] = 0x3F80;   // 1.0 -> the GEMM's extra column accumulates the bias gradient
    }
    *(u16x4*)(Xb + tok * KP + c * 4) = o;
  }
}

// grad_table[ids[tok]][:] += dropout1(dx[tok][:])   for ids[tok] != 0 (padding_idx).  dx bf16 [n_tokens][ldx].
__global__ __launch_bounds__(256) void embed_scatter_add_kernel(const int64_t* __restrict__ ids, const u16* __restrict__ dx,
                                                                int ldx, float* __restrict__ grad_table, int64_t num_rows,
                                                                int64_t n_tokens, DropCfg dc) {
  dc = drop_resolve(dc);          // same key as the forward's site-1 mask, also under a device step counter (HIP-graph replays)
  const int64_t total = n_tokens * D4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tok = i / D4;
    const int c = (int)(i - tok * D4);
    const int64_t id = ids[tok];
    if (id <= 0 || id >= num_rows) continue;
    u16x4 v = *(const u16x4*)(dx + tok * ldx + c * 4);
    f32x4 x = f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
    uint32_t keep = 0xF;
    float sc = 1.0f;
    if (dc.enabled) { keep = drop_keep4(dc, 1u, (uint64_t)tok * D4 + c); sc = dc.scale; }
    float* dst = grad_table + (id * D4 + c) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((keep >> j) & 1u) atomic_add(dst + j, x[j] * sc);
  }
}

// Sorted variant: positions are sorted by id (ids_sorted ascending, perm = original token index), so all
// contributions to one table row are adjacent.  One wave reduces SPAN consecutive positions in registers (lane =
// one 4-column quad, lanes 0..10 a second quad) and writes each finished row once; only rows whose run touches
// the span boundary (and may continue in a neighbour wave) are combined per workgroup and flushed with atomics.  The result is ADDED to grad_table (a row that only this
// wave touches is read, added and written back), so the destination may be the live .grad buffer of the parameter.
//
// Round 6: token ids are Zipf-distributed -- in a NAML batch (1.9 M tokens) the most frequent word occurs 98,831 times, the top ten hold 26 % of
// the positions and 77 % of the positions sit in runs longer than 32.  A run that continues in the neighbouring wave must be flushed with
// ATOMIC adds (300 per flush), and with one flush per wave the hot rows took thousands of them (3,088 waves on the top row).  A measured
// dead end (profiles/r06_scatter_ab.txt): longer spans per wave cut the flushes but also the number of waves -- each walks its rows with four
// loads in flight, so the gather starves (span 256: NRMS 164 -> 266 us).  What is kept: spans of 64 positions (all waves resident at once) and
// the flushes of a WORKGROUP's eight waves combined -- a wave deposits the partial sums of its first / last run in LDS when the run is shared
// with a neighbour, and wave 0 adds up the consecutive deposits of equal id and flushes each merged run once: an eighth of the atomic traffic
// on the hot rows at unchanged parallelism.
constexpr int SC_SUB = 64;       // positions whose (id, token) pairs the lanes hold at a time
constexpr int SC_SPAN = 64;      // positions per wave (default instantiation)
constexpr int SC_WAVES = 8;      // waves per workgroup
constexpr int SC_EROW = 304;     // floats per deposited partial: 64 quads + 11 quads (+ padding)
constexpr int SC_SMEM = SC_WAVES * 2 * SC_EROW * 4 + SC_WAVES * 2 * 4;
// a row quad as it sits in memory (bf16: 8 bytes, two registers; f32: 16 bytes) and its f32 value: the prefetched rows are HELD in the memory
// format -- twice as many bf16 rows in flight per register as converted ones (the walk is bound by the depth of its row prefetch)
template <typename SRC> struct RowQuad;
template <> struct RowQuad<u16> {
  typedef u16x4 raw;
  static constexpr int GR = 8;
  static __device__ __forceinline__ raw ld(const u16* p) { return *(const u16x4*)p; }
  static __device__ __forceinline__ raw zero() { return u16x4{0, 0, 0, 0}; }
  static __device__ __forceinline__ f32x4 val(raw v) { return f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])}; }
};
template <> struct RowQuad<float> {
  typedef f32x4 raw;
  static constexpr int GR = 4;
  static __device__ __forceinline__ raw ld(const float* p) { return *(const f32x4*)p; }
  static __device__ __forceinline__ raw zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ f32x4 val(raw v) { return v; }
};

// SRC = u16 (bf16 rows) or float.  Rows with id <= pad_row are skipped (pad_row = 0: nn.Embedding(padding_idx=0); -1: none).
template <typename SRC, int SPAN>
__global__ __launch_bounds__(SC_WAVES * 64) void embed_scatter_sorted_kernel(const int64_t* __restrict__ ids_sorted,
                                                                             const int64_t* __restrict__ perm, const SRC* __restrict__ dx,
                                                                             int64_t ldx, float* __restrict__ grad_table, int64_t num_rows,
                                                                             int64_t n_tokens, DropCfg dc, int pad_row) {
  static_assert(SPAN % SC_SUB == 0, "whole sub-chunks");
  NR_SMEM_DECL(smem);
  float* edge = (float*)smem;                                // [SC_WAVES][2][SC_EROW]
  int* edge_id = (int*)(smem + SC_WAVES * 2 * SC_EROW * 4);  // [SC_WAVES][2]: id of the deposit, -1 = none
  dc = drop_resolve(dc);
  const int l = lane_id(), w = wave_id();
  const bool two = l < (D4 - 64);
  if (l < 2) edge_id[w * 2 + l] = -1;
  const int64_t s0 = ((int64_t)blockIdx.x * SC_WAVES + w) * SPAN;
  const int span = s0 >= n_tokens ? 0 : (int)((n_tokens - s0) < SPAN ? (n_tokens - s0) : SPAN);
  const bool active = span > 0 && (int)ids_sorted[s0 + (span > 0 ? span - 1 : 0)] > pad_row;      // (sorted: ids <= pad_row come first -- an all-padding span has nothing to do)
  if (active) {
    const int id_before = s0 > 0 ? (int)ids_sorted[s0 - 1] : -2;
    const int id_after = s0 + span < n_tokens ? (int)ids_sorted[s0 + span] : -2;
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur = (int)ids_sorted[s0];
    // current contents of the destination row of the run in progress, requested when the run starts so that the read-add-write of
    // flush() does not wait on a round trip per finished row
    f32x4 pd0 = f32x4{0.f, 0.f, 0.f, 0.f}, pd1 = pd0;
    auto peek = [&](int id) {
      if (id <= pad_row || id >= num_rows) return;
      const float* src = grad_table + ((int64_t)id * D4 + l) * 4;
      pd0 = *(const f32x4*)src;
      if (two) pd1 = *(const f32x4*)(src + 256);
    };
    peek(cur);
    // a finished run: not shared with a neighbouring wave -> the row is this wave's alone: read (peeked) + add + write; shared -> deposit the partial
    // in LDS slot `which` (0: the span's first run, 1: its last) for the workgroup's merge below
    auto flush = [&](int id, bool shared, int which) {
      if (id <= pad_row || id >= num_rows) return;
      if (shared) {
        float* e = edge + (w * 2 + which) * SC_EROW;
        *(f32x4*)(e + l * 4) = a0;
        if (two) *(f32x4*)(e + 256 + l * 4) = a1;
        if (l == 0) edge_id[w * 2 + which] = id;
      } else {
        float* dst = grad_table + ((int64_t)id * D4 + l) * 4;
        *(f32x4*)dst = pd0 + a0;              // the value peeked at run start is current: no other wave touches this row
        if (two) *(f32x4*)(dst + 256) = pd1 + a1;
      }
    };
    bool first = true;
    for (int sub = 0; sub < span; sub += SC_SUB) {
      const int cnt = span - sub < SC_SUB ? span - sub : SC_SUB;
      // each lane keeps one (id, token) pair of the sub-chunk; broadcast by shuffle while walking it
      int id_lo = -1, tok_lo = 0;                              // ids and token indices fit in 31 bits
      if (l < cnt) { id_lo = (int)ids_sorted[s0 + sub + l]; tok_lo = (int)perm[s0 + sub + l]; }
      if (__builtin_bit_cast(int, shfl(__builtin_bit_cast(float, id_lo), cnt - 1)) <= pad_row) continue;      // (leading padding of the span)
      // The rows are fetched in groups of GR (bf16 rows: eight, held as loaded), one group ahead of the accumulation (two register sets): a plain
      // loop walks dependent load -> add round trips.  Same additions in the same order as the position-by-position walk.
      using RQ = RowQuad<SRC>;
      constexpr int GR = RQ::GR;
      typename RQ::raw c0[GR], c1[GR], n0[GR], n1[GR];
      auto fetch_group = [&](int base, typename RQ::raw (&r0)[GR], typename RQ::raw (&r1)[GR]) {
#pragma unroll
        for (int u = 0; u < GR; ++u) {
          const int i = base + u < cnt ? base + u : cnt - 1;
          const int tok = __builtin_bit_cast(int, shfl(__builtin_bit_cast(float, tok_lo), i));
          const SRC* row = dx + (int64_t)tok * ldx;
          r0[u] = RQ::ld(row + l * 4);
          r1[u] = two ? RQ::ld(row + 256 + l * 4) : RQ::zero();
        }
      };
      fetch_group(0, c0, c1);
      for (int base = 0; base < cnt; base += GR) {
        if (base + GR < cnt) fetch_group(base + GR, n0, n1);
#pragma unroll
        for (int u = 0; u < GR; ++u) {
          const int i = base + u;
          if (i < cnt) {
            const int id = __builtin_bit_cast(int, shfl(__builtin_bit_cast(float, id_lo), i));
            const int tok = __builtin_bit_cast(int, shfl(__builtin_bit_cast(float, tok_lo), i));
            if (id != cur) {
              flush(cur, first && cur == id_before, 0);
              first = false;
              cur = id;
              peek(cur);
              a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0;
            }
            if (id > pad_row) {
              const f32x4 v0 = RQ::val(c0[u]), v1 = RQ::val(c1[u]);
              uint32_t k0 = 0xF, k1 = 0xF;
              float sc = 1.0f;
              if (dc.enabled) {
                k0 = drop_keep4(dc, 1u, (uint64_t)tok * D4 + l);
                k1 = two ? drop_keep4(dc, 1u, (uint64_t)tok * D4 + 64 + l) : 0u;
                sc = dc.scale;
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if ((k0 >> j) & 1u) a0[j] += v0[j] * sc;
                if ((k1 >> j) & 1u) a1[j] += v1[j] * sc;
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < GR; ++u) { c0[u] = n0[u]; c1[u] = n1[u]; }
      }
    }
    const bool sh_prev = first && cur == id_before;           // the last run is also the span's first run and began in the previous wave
    flush(cur, sh_prev || cur == id_after, sh_prev ? 0 : 1);
  }
  __syncthreads();
  // ---- the workgroup's deposits in position order (wave 0 slot 0, slot 1, wave 1 slot 0, ...): consecutive deposits of one id are one run -- summed
  // and flushed ONCE, atomically (the run may go on in the neighbouring workgroups, and nobody peeked at its row) ------------------------------------
  if (w == 0) {
    int cur = -1;
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
    auto flush_atomic = [&](int id) {
      if (id < 0) return;
      float* dst = grad_table + ((int64_t)id * D4 + l) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomic_add(dst + j, a0[j]);
      if (two) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomic_add(dst + 256 + j, a1[j]);
      }
    };
    for (int e = 0; e < SC_WAVES * 2; ++e) {
      const int id = edge_id[e];
      if (id < 0) continue;
      if (id != cur) {
        flush_atomic(cur);
        cur = id;
        a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0;
      }
      const float* er = edge + e * SC_EROW;
      a0 += *(const f32x4*)(er + l * 4);
      if (two) a1 += *(const f32x4*)(er + 256 + l * 4);
    }
    flush_atomic(cur);
  }
}

// d_cand[b,c,:] = dl[b,c] * user[b,:];  d_user[b,:] = sum_c dl[b,c] * cand[b,c,:]
__global__ __launch_bounds__(256) void score_dot_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ cand,
                                                            const float* __restrict__ user, float* __restrict__ d_cand,
                                                            float* __restrict__ d_user, int64_t B, int C, int d4) {
  const int64_t total = B * d4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / d4;
    const int c4 = (int)(i - b * d4);
    f32x4 u = *(const f32x4*)(user + (b * d4 + c4) * 4);
    f32x4 du = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      const float g_ = dl[b * C + c];
      f32x4 cv = *(const f32x4*)(cand + ((b * C + c) * d4 + c4) * 4);
      du += cv * g_;
      *(f32x4*)(d_cand + ((b * C + c) * d4 + c4) * 4) = u * g_;
    }
    *(f32x4*)(d_user + (b * d4 + c4) * 4) = du;
  }
}

}  // namespace nr
