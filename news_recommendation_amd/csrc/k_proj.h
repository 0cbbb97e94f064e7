// Training form of the NRMS news-encoder front end as TWO kernels with a GEMM core (gfx950):
//
//   qkv_proj_kernel   x = dropout(table[ids])  ->  Q | K | V = x W^T + b   (src/model/NRMS/news_encoder.py:38-40,
//                     src/model/general/attention/multihead_self.py:53-58).  A gather-fused projection GEMM
//                     [tokens x 304] x [304 x 960] on v_mfma_f32_32x32x16_bf16: a wave owns 32 tokens whose rows go from the fp32
//                     table straight into A/B operand fragments (76 VGPRs, dropout and bf16 rounding in the loader); the weights
//                     stream through LDS in 32-column chunks (global_load_lds_dwordx4, fragment-major "tile32 order", shared by
//                     the four waves of a workgroup, double buffered); three workgroups share a CU (3 waves per SIMD) so that one
//                     wave's gather / epilogue stores run beside the MFMAs of the others.
//   attn_fwd_kernel   ScaledDotProductAttention (multihead_self.py:15-23) per (title, head) from the saved Q, K, V^T: one wave per
//                     pair, operands straight from memory into fragments, no LDS, no workgroup barrier, high occupancy; writes the
//                     ctx rows (second dropout of news_encoder.py:43-45 applied) the pooling kernels read.
//
// The register-resident single-wave kernel (k_mhsa_fwd2.h) stays the inference form: it never writes Q / K / V.  In training every
// value it keeps in registers has to be written for the backward anyway, and holding them cost it occupancy (1 wave per SIMD).
//
// Saved-activation layout shared by both kernels and nr_attn_bwd_hm ("head-major"): qkv bf16 [n_seq][15][3][400]:
//   block 0 = Q [token][d], block 1 = K [token][d], block 2 = V^T [d][token]  (20 x 20 each, 2,400 B contiguous per (title, head)),
// so that the attention kernels fetch a pair's operands as whole cache lines instead of 40-byte pieces of 640-byte token rows.
#pragma once
#include "nr_common.h"
#include <type_traits>

namespace nr {

constexpr int K16 = (D + 15) / 16;          // 19 k-steps of 16 cover the 300 features (304)
constexpr int NT32 = NP / 32;               // 10 column tiles of 32 per projection
constexpr int HM_BLK = 20 * DK;             // 400 elements: one 20 x 20 block
constexpr int HM_PAIR = 3 * HM_BLK;         // 1,200 elements per (title, head)

// "tile32 order" of the packed projection matrix W[3 * NP][304] (operand of v_mfma_f32_32x32x16_bf16): block (32-row tile, 16-wide
// k-step) = 1 KiB = the 64 lanes' 16-byte fragments back to back; lane l holds W[32 T + (l & 31)][16 ks + 8 (l >> 5) + 0..7].
__device__ __host__ __forceinline__ size_t tile32_off(int r, int k) {
  return ((size_t)(r >> 5) * K16 + (k >> 4)) * 512 + ((((k & 15) >> 3) * 32) + (r & 31)) * 8 + (k & 7);
}

__global__ __launch_bounds__(256) void pack_qkv32_kernel(const float* __restrict__ Wq, const float* __restrict__ bq,
                                                         const float* __restrict__ Wk, const float* __restrict__ bk,
                                                         const float* __restrict__ Wv, const float* __restrict__ bv,
                                                         u16* __restrict__ Wp32, float* __restrict__ bp) {
  constexpr int KW = K16 * 16;
  const int total = 3 * NP * KW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int row = i / KW, k = i - row * KW;
    const int which = row / NP, n = row - which * NP;
    const float* W = which == 0 ? Wq : (which == 1 ? Wk : Wv);
    const float v = (n < D && k < D) ? W[n * D + k] : 0.0f;
    Wp32[tile32_off(row, k)] = f2bf(v);
    if (k == 0) {
      const float* b = which == 0 ? bq : (which == 1 ? bk : bv);
      bp[row] = n < D ? b[n] : 0.0f;
    }
  }
}

struct ProjGeom {
  static constexpr int S = 20;
  static constexpr int NWAVE = 4;
  static constexpr int TOKW = 32;                    // tokens per wave: one 32-row MFMA tile
  static constexpr int TOK_WG = NWAVE * TOKW;        // 128
  static constexpr int CH_BYTES = K16 * 1024;        // 19,456 B: one 32-column chunk of W = 19 fragment blocks
  static constexpr int B_BYTES = 3 * NP * 4;         // packed bias vector
  static constexpr int SMEM = 2 * CH_BYTES + B_BYTES;   // 42,752 B: three workgroups per CU
  static constexpr int NCHUNK = 3 * NT32;            // 30
  static constexpr int KB = 5;                       // k-steps of raw fp32 rows in flight per gather batch
};

struct ProjParams {
  const int64_t* ids;      // [n_tok]
  const float* table;      // [num_rows][D]
  int64_t num_rows;
  const u16* Wp32;         // [3*NP][304] in tile32 order
  const float* bp;         // [3*NP]
  u16* qkv;                // [n_tok / 20][H][3][400] head-major Q, K, V^T
  u16* x_save;             // optional [n_tok][KP]: the dropout-masked bf16 token matrix (col D = 1.0) for the weight-gradient GEMM
  int64_t n_tok;           // multiple of 20
  DropCfg dc;              // dropout site 1
  int debug;               // profiling only (NR_PROJ_DEBUG, DBG instantiation): 1 skip the table loads, 2 skip the MFMAs, 4 skip the Q / K / V^T
                           // stores, 8 skip the x_save stores, 16 skip the weight-chunk copies
};

#ifndef NR_PROJ_OCC
#define NR_PROJ_OCC 3      // waves per SIMD the register allocation must allow
#endif
// KSPLIT = 2: even / odd k-steps accumulate into two independent accumulators (no MFMA waits on the previous one's result)
template <int KSPLIT, bool DBG>
__global__ __launch_bounds__(256, NR_PROJ_OCC) void qkv_proj_kernel(ProjParams p) {
  using Gm = ProjGeom;
  constexpr int S = Gm::S;
  const int dbg = DBG ? p.debug : 0;
  NR_SMEM_DECL(smem);
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), h = l >> 5, li = l & 31;
  const int64_t tile_tok0 = ((int64_t)blockIdx.x * Gm::NWAVE + w) * Gm::TOKW;
  const int64_t tok = tile_tok0 + li;
  const bool live = tok < p.n_tok;

  // chunk c = 32 consecutive rows of the packed matrix (projection c / 10, columns 32 (c % 10) ..): global -> LDS directly; the image is
  // fragment-major already (tile32 order), so block ks is read back as ONE conflict-free ds_read_b128 at block + 16 lane
  auto chunk_fetch = [&](int c, int buf) {
    const u16* src = p.Wp32 + (size_t)c * K16 * 512 + l * 8;
    unsigned char* dst = smem + buf * Gm::CH_BYTES;
    if (dbg & 16) return;
    for (int blk = w; blk < K16; blk += Gm::NWAVE) NR_GLDS16(src + blk * 512, dst + blk * 1024);
  };
  chunk_fetch(0, 0);
  float* bl = (float*)(smem + 2 * Gm::CH_BYTES);
  for (int i = tid; i < 3 * NP / 4; i += 256) *(f32x4*)(bl + i * 4) = *(const f32x4*)(p.bp + i * 4);

  // ---- gather the wave's 32 token rows into operand fragments: lane (token li, half h) holds features 16 ks + 8 h .. + 7 -------------
  u16x8 xf[K16];
  const float* row = p.table;
  if (live) {
    int64_t id = p.ids[tok];
    id = id < 0 ? 0 : (id >= p.num_rows ? p.num_rows - 1 : id);
    row = p.table + (size_t)id * D;
  }
#pragma unroll
  for (int kb = 0; kb < K16; kb += Gm::KB) {
    f32x4 lo[Gm::KB], hi[Gm::KB];
#pragma unroll
    for (int j = 0; j < Gm::KB; ++j) {
      const int ks = kb + j;
      if (ks < K16) {
        const int c = ks * 16 + h * 8;                  // c + 3 < D for every (ks, h); c + 4 .. c + 7 leave the row at (18, 1)
        lo[j] = (live && !(dbg & 1)) ? *(const f32x4*)(row + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        hi[j] = (live && c + 4 < D && !(dbg & 1)) ? *(const f32x4*)(row + c + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int j = 0; j < Gm::KB; ++j) {
      const int ks = kb + j;
      if (ks < K16) {
        const int c = ks * 16 + h * 8;
        f32x4 a = lo[j], b = hi[j];
        if (p.dc.enabled) {
          a = a * drop_mul4(p.dc, 1u, (uint64_t)tok * D4 + (c >> 2));
          if (c + 4 < D) b = b * drop_mul4(p.dc, 1u, (uint64_t)tok * D4 + (c >> 2) + 1);
        }
        xf[ks] = cat8(pack4(a), pack4(b));
        if (p.x_save != nullptr && live && !(dbg & 8)) {
          u16x8 o = xf[ks];
          if (ks == D / 16 && h == (D % 16) / 8) o[D % 8] = BF16_ONE;      // column D = 1.0: the weight-gradient GEMM then also yields the bias gradient
          *(u16x8*)(p.x_save + tok * KP + c) = o;
        }
      }
    }
  }
  if (p.x_save != nullptr && live) *(u16x8*)(p.x_save + tok * KP + K16 * 16 + h * 8) = u16x8{0, 0, 0, 0, 0, 0, 0, 0};   // cols 304 .. 319
  __syncthreads();

  const int64_t seq = tok / S;
  const int tis = (int)(tok - seq * S);
  u16* const qrow = p.qkv + seq * (H * HM_PAIR) + tis * DK;       // + (head * 3 + which) * 400 + d

  // one chunk: prefetch the next, 19 MFMAs, epilogue, barrier.  WHICH (0 Q, 1 K, 2 V) is a compile-time constant: Q and K come out of
  // the transposed product (A = weights: the lane ends up with 4 x 4 consecutive features of ITS token -> [token][d] rows), V out of the
  // plain product (A = tokens: 4 x 4 consecutive tokens of ITS feature -> [d][token] rows).
  auto run_chunk = [&](auto WHICH, int nt, int c) {
    constexpr int which = decltype(WHICH)::value;
    if (c + 1 < Gm::NCHUNK) chunk_fetch(c + 1, (c + 1) & 1);
    const u16* wp = (const u16*)(smem + (c & 1) * Gm::CH_BYTES) + l * 8;
    const int c0 = nt * 32;
    f32x16 acc[KSPLIT];
    if (which < 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *(const f32x4*)(bl + which * NP + c0 + 8 * q + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][4 * q + j] = b4[j];
      }
    } else {
      const float b1 = bl[2 * NP + c0 + li];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = b1;
    }
    if (KSPLIT > 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[KSPLIT - 1][r] = 0.0f;
    }
#pragma unroll
    for (int ks = (dbg & 2) ? K16 : 0; ks < K16; ++ks) {
      const u16x8 wf = *(const u16x8*)(wp + ks * 512);
      f32x16& a = acc[ks % KSPLIT];
      a = which < 2 ? mfma_32x32x16_bf16(wf, xf[ks], a) : mfma_32x32x16_bf16(xf[ks], wf, a);
    }
    if (KSPLIT > 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] += acc[KSPLIT - 1][r];
    }
    if (dbg & 4) {
    } else if (which < 2) {
      if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = c0 + 8 * q + 4 * h;
          if (col < D) {
            const int hd = col / DK, d = col - hd * DK;
            *(u16x4*)(qrow + (hd * 3 + which) * HM_BLK + d) = pack4(f32x4{acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]});
          }
        }
      }
    } else {
      const int col = c0 + li;
      if (col < D) {
        const int hd = col / DK, d = col - hd * DK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t t4 = tile_tok0 + 8 * q + 4 * h;        // first of 4 consecutive tokens, all inside one title (20 % 4 == 0)
          if (t4 < p.n_tok) {
            const int64_t s4 = t4 / S;
            const int ti4 = (int)(t4 - s4 * S);
            *(u16x4*)(p.qkv + (s4 * H + hd) * HM_PAIR + 2 * HM_BLK + d * S + ti4) =
                pack4(f32x4{acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]});
          }
        }
      }
    }
    __syncthreads();       // drains the in-flight global->LDS copies of chunk c + 1; everybody is done reading chunk c's buffer
  };
  using std::integral_constant;
  for (int nt = 0; nt < NT32; ++nt) run_chunk(integral_constant<int, 0>{}, nt, nt);
  for (int nt = 0; nt < NT32; ++nt) run_chunk(integral_constant<int, 1>{}, nt, NT32 + nt);
  for (int nt = 0; nt < NT32; ++nt) run_chunk(integral_constant<int, 2>{}, nt, 2 * NT32 + nt);
}

// ---------------------------------------------------------------------------------------------------------------------------------
struct AttnFwdParams {
  const u16* qkv;          // [n_seq][H][3][400] head-major Q, K, V^T (qkv_proj_kernel)
  u16* ctx;                // [n_seq*20][KP]: cols < D = attention output (x dropout 2), col D = 1.0, the rest of the K padding 0
  const int32_t* key_len;  // optional [n_seq]: keys >= key_len[seq] get zero weight (MultiHeadSelfAttention's `length`); null: 20
  int64_t n_seq;
  DropCfg dc;              // dropout site 2
  int debug;               // profiling only (NR_ATTNF_DEBUG, DBG instantiation): 1 skip the operand loads, 2 skip the exp / normalisation, 4 skip the ctx stores
};

struct AttnFwdGeom {
  static constexpr int S = 20;
  static constexpr int WPB = 4;       // waves per workgroup
  static constexpr int TPB = 4;       // titles per workgroup: the 15 heads of a title are computed by the same workgroup within a few
                                      // microseconds, so their 40-byte pieces of a ctx row merge in one L2 before the row is written back
};

// raw operand pieces of one (title, head) pair, loaded one pair ahead
struct AttnFwdRaw {
  u16x4 klo[2], khi[2], qlo[2], qhi[2], vlo[2], vhi[2];
};

template <bool DBG>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnFwdParams p) {
  using Gm = AttnFwdGeom;
  constexpr int S = Gm::S;
  const int dbg = DBG ? p.debug : 0;
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  const int64_t seq0 = (int64_t)blockIdx.x * Gm::TPB;
  const int nseq_blk = (int)((p.n_seq - seq0) < Gm::TPB ? (p.n_seq - seq0) : Gm::TPB);
  const int npairs = nseq_blk * H;
  const u16x4 Z4 = u16x4{0, 0, 0, 0};

  // K padding of the ctx rows of this workgroup: col D = 1.0 (bias-gradient column), cols D+1 .. KP-1 = 0
  constexpr int PADQ = (KP - D) / 4;
  for (int i = tid; i < nseq_blk * S * PADQ; i += Gm::WPB * 64) {
    const int r = i / PADQ, cq = i - r * PADQ;
    *(u16x4*)(p.ctx + (seq0 * S + r) * KP + D + cq * 4) = u16x4{(u16)(cq == 0 ? BF16_ONE : 0), 0, 0, 0};
  }

  // Fragment shapes (v_mfma_f32_16x16x32_bf16, lane = (li, g)):
  //   K / Q tile t as A / B operand of S^T = K Q^T: row 16 t + li, k-slots d = 8 g .. 8 g + 7 (d < 20); slot d = 20 carries the key mask
  //   (q = 1, k = 0 for a live key, -29952 for a padded one: exp2 of it underflows to exactly 0, no select per element);
  //   V^T tile t as A operand of ctx^T = V^T P^T: row dv = 16 t + li; the packed P^T tiles are the B operand with k-slot (g, j < 4) = key
  //   4 g + j and (g, j >= 4) = key 16 + 4 g + j - 4, so the V^T fragment takes keys 4 g .. 4 g + 3 and (g == 0) 16 .. 19.
  auto load = [&](int pi, AttnFwdRaw& r) {
    const u16* base = p.qkv + (seq0 * H + pi) * HM_PAIR;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int rw = 16 * t + li;
      const bool rok = rw < S && !(dbg & 1);
      const u16* q_ = base + rw * DK + 8 * g;
      r.qlo[t] = (rok && g < 3) ? *(const u16x4*)q_ : Z4;
      r.qhi[t] = (rok && g < 2) ? *(const u16x4*)(q_ + 4) : Z4;
      r.klo[t] = (rok && g < 3) ? *(const u16x4*)(q_ + HM_BLK) : Z4;
      r.khi[t] = (rok && g < 2) ? *(const u16x4*)(q_ + HM_BLK + 4) : Z4;
      const u16* v_ = base + 2 * HM_BLK + rw * S;
      r.vlo[t] = rok ? *(const u16x4*)(v_ + 4 * g) : Z4;
      r.vhi[t] = (rok && g == 0) ? *(const u16x4*)(v_ + 16) : Z4;
    }
  };

  const float c2 = LOG2E / sqrtf((float)DK), clamp2 = EXP_CLAMP * LOG2E;
  AttnFwdRaw cur, nxt;
  int pi = w;
  if (pi < npairs) load(pi, cur);
  for (; pi < npairs; pi += Gm::WPB) {
    const int pn = pi + Gm::WPB;
    if (pn < npairs) load(pn, nxt);
    const int sq = pi / H, hd = pi - sq * H;
    const int64_t seq = seq0 + sq;
    const int klen = p.key_len != nullptr ? uniform(clamp_len(p.key_len[seq], S)) : S;
    u16x8 kf[2], qf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      kf[t] = cat8(cur.klo[t], cur.khi[t]);
      qf[t] = cat8(cur.qlo[t], cur.qhi[t]);
      if (g == 2) { kf[t][4] = (16 * t + li < klen) ? (u16)0 : BF16_NEG_BIG; qf[t][4] = BF16_ONE; }
    }
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
      const u16x8 va = cat8(cur.vlo[t], cur.vhi[t]);
#pragma unroll
      for (int qj = 0; qj < 2; ++qj) {
        f32x4 acc = mfma_16x16x32_bf16(va, cat8(pt[0][qj], pt[1][qj]), f32x4{0.f, 0.f, 0.f, 0.f});
        const int tokl = 16 * qj + li, dv = 16 * t + 4 * g;
        if (tokl < S && dv < DK && !(dbg & 4)) {
          const int64_t tok = seq * S + tokl;
          const int col = hd * DK + dv;
          if (p.dc.enabled) acc = acc * drop_mul4(p.dc, 2u, (uint64_t)tok * D4 + (col >> 2));
          *(u16x4*)(p.ctx + tok * KP + col) = pack4(acc);
        }
      }
    }
    cur = nxt;
  }
}

}  // namespace nr
