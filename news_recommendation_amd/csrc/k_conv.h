// Three-tap token convolution for gfx950: the title / abstract CNN of NAML and LSTUR,
//   y[s][f] = b[f] + sum_{w=0..2} sum_d x[s + w - 1][d] * W[f][w][d]     (zero rows outside the sequence)
// = nn.Conv2d(1, F, (3, D), padding=(1, 0)) on [B,1,S,D], squeezed and transposed
// (src/model/NAML/news_encoder.py:15-17,27-36; src/model/LSTUR/news_encoder.py:24-28,62-69),
// fused with its input stage (embedding gather + F.dropout, NAML :23-25 / LSTUR :58-60) and its
// output stage (F.relu + F.dropout, NAML :30-32 / LSTUR :64-67).
//
// The same kernel is the data gradient of the convolution: dX[s][d] = sum_w' sum_f dY[s + w' - 1][f] Wd[d][w'][f]
// with Wd[d][w'][f] = W[f][2 - w'][d] (flipped taps, transposed filters) -- dense bf16 input, plain epilogue.
//
// "seqpad" layout (LDS tile and the HBM buffers x_pad / x_save): row(seq, s) = seq*(S+1) + 1 + s, rows seq*(S+1)
// are all-zero separators shared by neighbouring sequences, so the tap shift is a plain row offset and needs no
// boundary tests.  In HBM the buffer has n_seq*(S+1)+1 rows of KP bf16.
//
// Workgroup = 4 waves, NSEQ sequences.  GEMM per workgroup: [TOK] x [3*KP] x [304]; filters are the MFMA A operand
// (rows straight from L2 into registers, one tap at a time), tokens the B operand (fragments from the LDS tile at
// row + tap); a wave keeps the accumulators of its (column group, token tiles) block across the three taps.  The lane
// ends up with 4 consecutive filters of one token -> one 8-B store.
#pragma once
#include "nr_common.h"

namespace nr {

constexpr int NPC = KP;          // rows per tap block of the packed conv weight [3][NPC][KP]
// LDS row stride (elements) of THIS kernel's token tile: 672 B.  The chip services a b128 fragment read in four non-contiguous 16-lane groups on 64
// banks (MI355X_MICROARCH.md, LDS): with nr_common.h's 656-byte rows (laid out for 8-lane groups on 32 banks) a fragment read of 16 consecutive
// token rows costs 8.8 LDS cycles instead of 4 (tools/lds_bank_model.py; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.51,
// profiles/r06_pmc_sq_conv_abs.txt).  672 B: 4.9 cycles (what is left are the tiles that straddle a separator row); measured: conflicts 0.51 -> 0.18
// of the LDS cycles, kernel time unchanged within noise (profiles/r06_ab_conv_lds_stride.txt) -- the LDS does not bound this kernel either.
constexpr int XSC = KP + 16;
constexpr int NTF = (D + 15) / 16;   // 19 filter tiles

template <int S, int NSEQ>
struct ConvGeom {
  static constexpr int TOK = S * NSEQ;
  static constexpr int MT = (TOK + 15) / 16;
  static constexpr int PR = NSEQ * (S + 1) + 1;       // seqpad rows of the tile
  static constexpr int X_BYTES = (PR * XSC * 2 + 15) / 16 * 16;
  static constexpr int IDS_BYTES = (TOK * 4 + 15) / 16 * 16;
  static constexpr int SMEM = X_BYTES + IDS_BYTES;
};

struct ConvParams {
  const int64_t* ids;      // gather form: [n_seq*S] token ids, or null
  const float* table;      // [num_rows][D]
  int64_t num_rows;
  const u16* x_pad;        // dense form: bf16 seqpad [n_seq*(S+1)+1][KP] (columns >= D are ignored)
  const u16* Wc;           // bf16 [3][NPC][KP]  (tap, output row, k), each tap matrix in tile order
  const float* bc;         // f32 [NPC], or null (data-gradient form: no bias)
  u16* out;                // bf16 [n_seq*S][KP], plain token layout
  u16* x_save;             // gather form, training: bf16 seqpad copy of the dropout-masked tokens (col D = 1.0), or null
  int relu_drop;           // 1: out = dropout2(relu(y)), col D = 1.0, cols > D zero;  0: out = y (cols >= D untouched)
  int64_t n_seq;
  int64_t tok_offset;      // added to the token index in the dropout counters (keeps title / abstract streams apart)
  int valid;               // gather form: positions s >= valid of every sequence are ZERO vectors (texts shorter than S are zero-padded by
                           // the host: the convolution then sees the reference's zero padding right after the last real token); 0: all S
  DropCfg dc;
  int debug;               // profiling only (NR_CONV_DEBUG): 1 = skip the token gather, 2 = skip the GEMM, 4 = skip the output stores
};

template <int S, int NSEQ, int NW>
__global__ __launch_bounds__(NW * 64) void conv3_kernel(ConvParams p) {
  p.dc = drop_resolve(p.dc);
  using Gm = ConvGeom<S, NSEQ>;
  constexpr int WG = NW * 64;          // shadows nr::WG: this kernel runs with 4 or 8 waves
  NR_SMEM_DECL(smem);
  u16* Xs = (u16*)smem;
  int* ids_s = (int*)(smem + Gm::X_BYTES);
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  const int64_t seq0 = (int64_t)blockIdx.x * NSEQ;
  const int64_t tok0 = seq0 * S, tok_total = p.n_seq * S;

  // ---- stage the seqpad tile -------------------------------------------------------------------------------------
  if (p.ids != nullptr) {
    for (int r = tid; r < Gm::TOK; r += WG) {
      int v = -1;
      if (tok0 + r < tok_total && (p.valid <= 0 || (r % S) < p.valid)) {
        int64_t id = p.ids[tok0 + r];
        id = id < 0 ? 0 : (id >= p.num_rows ? p.num_rows - 1 : id);
        v = (int)id;
      }
      ids_s[r] = v;
    }
    for (int i = tid; i < Gm::X_BYTES / 16; i += WG) *(u16x8*)(Xs + i * 8) = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
    constexpr int TOTAL = Gm::TOK * D4;
    constexpr int U = 8;
    for (int base = 0; base < TOTAL; base += WG * U) {
      f32x4 v[U];
      int rr[U], cc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * WG + tid;
        const int r = i / D4, c = i - r * D4;
        rr[u] = r; cc[u] = c;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < TOTAL) {
          const int id = ids_s[r];
          if (id >= 0 && !(p.debug & 1)) v[u] = *(const f32x4*)(p.table + ((size_t)id * D4 + c) * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * WG + tid;
        if (i < TOTAL) {
          f32x4 x = v[u];
          if (p.dc.enabled && !(p.debug & 16)) {
            x = x * drop_mul4(p.dc, 1u, (uint64_t)(p.tok_offset + tok0 + rr[u]) * D4 + cc[u]);
          }
          const int row = rr[u] + rr[u] / S + 1;
          *(u16x4*)(Xs + row * XSC + cc[u] * 4) = pack4(x);
        }
      }
    }
    __syncthreads();
    if (p.x_save != nullptr && !(p.debug & 8)) {          // keep the masked bf16 tokens for the weight-gradient GEMMs (col D = 1.0 -> bias gradient)
      // every seqpad row of this workgroup incl. the zero separators (the shared ones are written twice, with zeros)
      constexpr int PCS = KP / 8;
      const int64_t rows_total = p.n_seq * (S + 1) + 1;
      for (int i = tid; i < Gm::PR * PCS; i += WG) {
        const int row = i / PCS, c = i - row * PCS;
        const int64_t gr = seq0 * (S + 1) + row;
        if (gr >= rows_total) continue;
        u16x8 v = *(const u16x8*)(Xs + row * XSC + c * 8);
        if (c == D / 8 && (row % (S + 1)) != 0 && gr < rows_total - 1) v[D % 8] = 0x3F80;
        *(u16x8*)(p.x_save + gr * KP + c * 8) = v;
      }
    }
  } else {
    constexpr int PCS = XSC / 8;        // 42 pieces per LDS row (incl. stride padding)
    const int64_t rows_total = p.n_seq * (S + 1) + 1;
    for (int i = tid; i < Gm::PR * PCS; i += WG) {
      const int r = i / PCS, c = i - r * PCS;
      const int64_t gr = seq0 * (S + 1) + r;
      u16x8 v = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (c * 8 < D && gr < rows_total) {
        v = *(const u16x8*)(p.x_pad + gr * KP + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c * 8 + j < D) ? v[j] : (u16)0;
      }
      *(u16x8*)(Xs + r * XSC + c * 8) = v;
    }
    __syncthreads();
  }

  // ---- GEMM over the three taps ----------------------------------------------------------------------------------
  // A wave owns a contiguous range of (filter-tile pair, token tile) units (unit_range).  Per pair it walks its token tiles in
  // chunks of MC: the chunk's accumulators stay in registers across the three taps, the filter fragments of one tap (G x KSTEPS,
  // tile order: coalesced 1 KB loads) are held for the whole chunk, and the X fragments of one token tile are fetched from LDS as
  // ONE batch of KSTEPS reads ahead of its MFMAs (sched_barrier).  G is a compile-time constant here: with a runtime G the
  // compiler put a branch between the two MFMAs of every k-step and waited for each ds_read right before its MFMA
  // (lgkmcnt(0) per k-step): the matrix pipe idled ~60 % of the GEMM phase.
  const int w_eff = (w + (int)blockIdx.x) % NW;
  auto gemm = [&](auto GC, int cg, int mb, int me) {
    constexpr int G = decltype(GC)::value;
    constexpr int MC = (Gm::MT + 1) / 2;
    const int wr[2] = {(2 * cg) * 16, (2 * cg) * 16 + 16};
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bias[G];
#pragma unroll
    for (int j = 0; j < G; ++j) bias[j] = p.bc ? *(const f32x4*)(p.bc + wr[j] + 4 * g) : zero4;
    for (int c0 = mb; c0 < me; c0 += MC) {
      f32x4 acc[MC][G];
      int rowoff[MC];
#pragma unroll
      for (int mi = 0; mi < MC; ++mi) {
#pragma unroll
        for (int j = 0; j < G; ++j) acc[mi][j] = bias[j];
        int t = (c0 + mi) * 16 + li;
        t = t < Gm::TOK ? t : Gm::TOK - 1;
        rowoff[mi] = (t + t / S) * XSC + g * 8;          // seqpad row of tap 0 (= row(t) - 1)
      }
#pragma unroll 1
      for (int tap = (p.debug & 2) ? 3 : 0; tap < 3; ++tap) {
        u16x8 wf[G][KSTEPS];
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const u16* wp = p.Wc + ((size_t)tap * NPC + wr[j]) * KP + l * 8;      // tile order: k-step ks of a row tile = + ks * 512
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) wf[j][ks] = *(const u16x8*)(wp + ks * 512);
        }
#pragma unroll
        for (int mi = 0; mi < MC; ++mi) {
          if (c0 + mi < me) {
            const u16* xp = Xs + rowoff[mi] + tap * XSC;
            u16x8 xf[KSTEPS];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) xf[ks] = *(const u16x8*)(xp + ks * 32);
            NR_SCHED_BARRIER();
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
              for (int j = 0; j < G; ++j) acc[mi][j] = mfma_16x16x32_bf16(wf[j][ks], xf[ks], acc[mi][j]);
            }
            NR_SCHED_BARRIER();
          }
        }
      }
      // epilogue: lane holds filters col .. col+3 of token m*16 + li
#pragma unroll
      for (int mi = 0; mi < MC; ++mi) {
        if (c0 + mi < me) {
#pragma unroll
          for (int j = 0; j < G; ++j) {
            const int t = (c0 + mi) * 16 + li;
            const int col = wr[j] + 4 * g;
            const int64_t tok = tok0 + t;
            if (t < Gm::TOK && tok < tok_total && col < D && !(p.debug & 4)) {
              f32x4 y = acc[mi][j];
              if (p.relu_drop) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.0f);
                if (p.dc.enabled) {
                  y = y * drop_mul4(p.dc, 2u, (uint64_t)(p.tok_offset + tok) * D4 + (col >> 2));
                }
              }
              *(u16x4*)(p.out + tok * KP + col) = pack4(y);
            }
          }
        }
      }
    }
  };
  for (int cg = 0; cg < (NTF + 1) / 2; ++cg) {
    int G, mb, me;
    unit_range(NTF, Gm::MT, w_eff, NW, cg, G, mb, me);
    if (mb >= me) continue;
    if (G == 2) gemm(std::integral_constant<int, 2>{}, cg, mb, me);
    else gemm(std::integral_constant<int, 1>{}, cg, mb, me);
  }
  if (p.relu_drop && !(p.debug & 32)) {      // K padding of the activation rows: col D = 1.0 (bias-gradient column), cols D+1..KP = 0
    constexpr int PADQ = (KP - D) / 4;
    for (int i = tid; i < Gm::TOK * PADQ; i += WG) {
      const int r = i / PADQ, c = i - r * PADQ;
      const int64_t tok = tok0 + r;
      if (tok < tok_total) *(u16x4*)(p.out + tok * KP + D + c * 4) = u16x4{(u16)(c == 0 ? 0x3F80 : 0), 0, 0, 0};
    }
  }
}

// ---- weight packing: Conv2d weight f32 [F][1][3][D] + bias -> forward operand Wc[tap][f][d] and data-gradient operand
// Wd[tap'][d][f] = W[f][2 - tap'][d], both bf16 [3][NPC][KP] zero padded; bc f32 [NPC].
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ W, const float* __restrict__ b, int F_, int D_,
                                                        u16* __restrict__ Wc, u16* __restrict__ Wd, float* __restrict__ bc) {
  const int total = 3 * NPC * KP;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int tap = i / (NPC * KP), rem = i - tap * NPC * KP;
    const int row = rem / KP, k = rem - row * KP;
    float fwd = 0.0f, bwd = 0.0f;
    if (row < F_ && k < D_) fwd = W[((size_t)row * 3 + tap) * D_ + k];                 // Wc[tap][f=row][d=k]
    if (row < D_ && k < F_) bwd = W[((size_t)k * 3 + (2 - tap)) * D_ + row];           // Wd[tap][d=row][f=k]
    const size_t o = (size_t)tap * NPC * KP + tile_off(row, k, KP);      // each tap matrix [NPC][KP] in tile order
    Wc[o] = f2bf(fwd);
    if (Wd != nullptr) Wd[o] = f2bf(bwd);
    if (i < NPC) bc[i] = i < F_ ? b[i] : 0.0f;
  }
}

// ---- gradient of the activation stage: dY = (dact_gemm + attn_w (x) g_out) * [act != 0] * scale, written in seqpad
// layout for the weight-gradient GEMMs and the data-gradient convolution.  act is the saved forward output
// dropout2(relu(y)): act == 0 <=> dropped or y <= 0 (relu'(0) = 0 as in torch), in both cases the gradient is 0.
__global__ __launch_bounds__(256) void conv_act_bwd_kernel(const u16* __restrict__ act, const u16* __restrict__ dact_gemm, int ldc,
                                                           const float* __restrict__ attn_w, const float* __restrict__ g_out,
                                                           int64_t g_stride, u16* __restrict__ dy_pad, int64_t n_seq, int S, float scale) {
  const int64_t total = n_seq * S * D4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tok = i / D4;
    const int c = (int)(i - tok * D4);
    const int64_t seq = tok / S;
    const u16x4 a = *(const u16x4*)(act + tok * KP + c * 4);
    const u16x4 dg = *(const u16x4*)(dact_gemm + tok * ldc + c * 4);
    const f32x4 go = *(const f32x4*)(g_out + seq * g_stride + c * 4);
    const float wt = attn_w[tok];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (a[j] & 0x7FFF) ? (bf2f(dg[j]) + wt * go[j]) * scale : 0.0f;
    *(u16x4*)(dy_pad + (tok + seq + 1) * KP + c * 4) = pack4(o);
  }
}

// ---- gradient w.r.t. the input of an additive-attention pooling that is not followed by one of the fused backward
// kernels: dx[tok][:] = dgemm[tok][:] + attn_w[tok] * g_out[seq][:]  (additive.py:50-52 backward), fp32 output.
// view_major != 0 stores row tok at (tok % S) * n_seq + tok / S, i.e. S contiguous [n_seq][D] blocks.
__global__ __launch_bounds__(256) void additive_dx_kernel(const u16* __restrict__ dgemm, int ldc, const float* __restrict__ attn_w,
                                                          const float* __restrict__ g_out, float* __restrict__ dx, int64_t n_seq, int S,
                                                          int view_major) {
  const int64_t total = n_seq * S * D4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tok = i / D4;
    const int c = (int)(i - tok * D4);
    const int64_t seq = tok / S;
    const u16x4 dg = *(const u16x4*)(dgemm + tok * ldc + c * 4);
    const f32x4 go = *(const f32x4*)(g_out + (seq * D4 + c) * 4);
    const float wt = attn_w[tok];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = bf2f(dg[j]) + wt * go[j];
    const int64_t orow = view_major ? (tok - seq * S) * n_seq + seq : tok;
    *(f32x4*)(dx + (orow * D4 + c) * 4) = o;
  }
}

}  // namespace nr
