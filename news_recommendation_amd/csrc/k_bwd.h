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
  static constexpr int RS = R + 8;                  // row stride of [x][token] matrices
  static constexpr int DS = 24;                     // row stride of [token][d] matrices (DK=20 padded to 24)
  static constexpr int SP4 = (S + 3) / 4 * 4;
  static constexpr int KS = (R + 31) / 32;          // k-steps over tokens
  static constexpr int TD_ELEMS = R * DS;           // Qm, Km, Vm, dCm
  static constexpr int DT_ELEMS = DK * RS + 32;     // Qt, Kt, dCt (+slack for the 8-wide reads of the last row)
  static constexpr int TT_ELEMS = R * RS + 32;      // PmT, dSt
  static constexpr int WAVE_ELEMS = 4 * TD_ELEMS + 3 * DT_ELEMS + 2 * TT_ELEMS;
  static constexpr int WAVE_BYTES = (WAVE_ELEMS * 2 + 15) / 16 * 16;
  static constexpr int SMEM = WPB * WAVE_BYTES;
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
  DropCfg dc;            // dropout site 2 (applied to ctx in the forward)
};

__device__ __forceinline__ u16x8 ld8(const u16* p) { return cat8(*(const u16x4*)p, *(const u16x4*)(p + 4)); }

template <int S, int WPB>
__global__ __launch_bounds__(WPB * 64) void attn_bwd_kernel(AttnBwdParams p) {
  using Gm = AttnBwdGeom<S, WPB>;
  NR_SMEM_DECL(smem);
  const int l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  u16* base = (u16*)(smem + w * Gm::WAVE_BYTES);
  u16* Qm = base;
  u16* Km = Qm + Gm::TD_ELEMS;
  u16* Vm = Km + Gm::TD_ELEMS;
  u16* dCm = Vm + Gm::TD_ELEMS;
  u16* Qt = dCm + Gm::TD_ELEMS;
  u16* Kt = Qt + Gm::DT_ELEMS;
  u16* dCt = Kt + Gm::DT_ELEMS;
  u16* PmT = dCt + Gm::DT_ELEMS;
  u16* dSt = PmT + Gm::TT_ELEMS;

  const int64_t pair = (int64_t)blockIdx.x * WPB + w;
  const bool live = pair < p.n_seq * H;
  const int64_t seq = live ? pair / H : 0;
  const int hd = live ? (int)(pair - seq * H) : 0;
  const int64_t tok0 = seq * S;

  // zero the whole wave-private scratch once: every padding row / column must be finite (zero)
  for (int i = l; i < Gm::WAVE_BYTES / 16; i += 64) *(u16x8*)(base + i * 8) = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  wave_barrier();

  if (live) {
    // ---- load Q, K (row-major + transposed copies), V (from dv-major blocks), dC (with direct term + dropout) ----
    constexpr int PCS = DK / 4;   // 8-B pieces per 20-wide row
    for (int i = l; i < S * PCS; i += 64) {
      const int r = i / PCS, c = (i - r * PCS) * 4;
      u16x4 q = *(const u16x4*)(p.q_save + (tok0 + r) * KP + hd * DK + c);
      u16x4 k = *(const u16x4*)(p.k_save + (tok0 + r) * KP + hd * DK + c);
      *(u16x4*)(Qm + r * Gm::DS + c) = q;
      *(u16x4*)(Km + r * Gm::DS + c) = k;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Qt[(c + j) * Gm::RS + r] = q[j];
        Kt[(c + j) * Gm::RS + r] = k[j];
      }
      // dC[q][dv] = (dctx_gemm + w[q] * g_out[dv]) * dropout2
      u16x4 dg = *(const u16x4*)(p.dctx_gemm + (tok0 + r) * p.ldc + hd * DK + c);
      f32x4 go = *(const f32x4*)(p.g_out + seq * D + hd * DK + c);
      const float wt = p.attn_w[tok0 + r];
      f32x4 dc4;
#pragma unroll
      for (int j = 0; j < 4; ++j) dc4[j] = bf2f(dg[j]) + wt * go[j];
      if (p.dc.enabled) {
        uint32_t keep = drop_keep4(p.dc, 2u, (uint64_t)(tok0 + r) * D4 + ((hd * DK + c) >> 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) dc4[j] = ((keep >> j) & 1u) ? dc4[j] * p.dc.scale : 0.0f;
      }
      u16x4 dcb = pack4(dc4);
      *(u16x4*)(dCm + r * Gm::DS + c) = dcb;
#pragma unroll
      for (int j = 0; j < 4; ++j) dCt[(c + j) * Gm::RS + r] = dcb[j];
    }
    constexpr int VP = Gm::SP4 / 4;   // 8-B pieces per dv row of the saved block
    const u16* vblk = p.vt_save + (seq * H + hd) * DK * Gm::SP4;
    for (int i = l; i < DK * VP; i += 64) {
      const int dv = i / VP, t = (i - dv * VP) * 4;
      u16x4 v = *(const u16x4*)(vblk + dv * Gm::SP4 + t);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (t + j < S) Vm[(t + j) * Gm::DS + dv] = v[j];
    }
  }
  wave_barrier();

  if (live) {
    const float inv_sqrt_dk = 1.0f / sqrtf((float)DK);
    // fragments over the head dim (k-slots: d = 8g + j; g == 3 and the upper half of g == 2 are padding)
    auto frag_d = [&](const u16* M, int row) -> u16x8 {
      const u16* q_ = M + row * Gm::DS + 8 * g;
      u16x4 z = u16x4{0, 0, 0, 0};
      u16x4 lo = (8 * g < DK) ? *(const u16x4*)q_ : z;
      u16x4 hi = (8 * g + 4 < DK) ? *(const u16x4*)(q_ + 4) : z;
      return cat8(lo, hi);
    };
    u16x8 kf[Gm::QT], qf[Gm::QT], vf[Gm::QT], cf[Gm::QT];
#pragma unroll
    for (int t = 0; t < Gm::QT; ++t) {
      kf[t] = frag_d(Km, t * 16 + li);
      qf[t] = frag_d(Qm, t * 16 + li);
      vf[t] = frag_d(Vm, t * 16 + li);
      cf[t] = frag_d(dCm, t * 16 + li);
    }
    u16x4 dsb[Gm::QT][Gm::QT];     // dS^T packed bf16: [key tile][query tile]
#pragma unroll
    for (int qt = 0; qt < Gm::QT; ++qt) {
      // ---- recompute P^T (column = query li, rows = keys) --------------------------------------------------
      f32x4 pT[Gm::QT];
      float mx = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < Gm::QT; ++kt) {
        pT[kt] = mfma_16x16x32_bf16(kf[kt], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pT[kt][r] *= inv_sqrt_dk;
          if (kt * 16 + 4 * g + r < S) mx = fmaxf(mx, pT[kt][r]);
        }
      }
      mx = fmaxf(mx, shfl_xor(mx, 16));
      mx = fmaxf(mx, shfl_xor(mx, 32));
      float sum = 0.0f;
#pragma unroll
      for (int kt = 0; kt < Gm::QT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = (kt * 16 + 4 * g + r < S) ? fast_exp(pT[kt][r] - mx) : 0.0f;
          pT[kt][r] = e;
          sum += e;
        }
      sum += shfl_xor(sum, 16);
      sum += shfl_xor(sum, 32);
      const float rden = fast_rcp(sum + 1e-8f * fast_exp(-mx));
      // ---- dP^T = V dC^T, dS^T = P^T * (dP^T - sum_keys P^T dP^T) / sqrt(dk) ---------------------------------
      f32x4 dP[Gm::QT];
      float dot = 0.0f;
#pragma unroll
      for (int kt = 0; kt < Gm::QT; ++kt) {
        pT[kt] = pT[kt] * rden;
        dP[kt] = mfma_16x16x32_bf16(vf[kt], cf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) dot += pT[kt][r] * dP[kt][r];
      }
      dot += shfl_xor(dot, 16);
      dot += shfl_xor(dot, 32);
      const int q = qt * 16 + li;
#pragma unroll
      for (int kt = 0; kt < Gm::QT; ++kt) {
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = pT[kt][r] * (dP[kt][r] - dot) * inv_sqrt_dk;
        u16x4 pb = pack4(pT[kt]);
        dsb[kt][qt] = pack4(ds);
        // transposed copies [key][query] for the products that contract over queries
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * g + r;
          if (key < S && q < S) {
            PmT[key * Gm::RS + q] = pb[r];
            dSt[key * Gm::RS + q] = dsb[kt][qt][r];
          }
        }
      }
    }
    // ---- dQ^T[d][q] = sum_key Kt[d][key] dS^T[key][q]  (B operand straight from registers) -------------------
#pragma unroll
    for (int dt = 0; dt < (DK + 15) / 16; ++dt) {
      int drow = dt * 16 + li;
      drow = drow < DK ? drow : DK - 1;
      const int d0 = dt * 16 + 4 * g;
      u16x8 af[(Gm::QT + 1) / 2];
#pragma unroll
      for (int kp = 0; kp < (Gm::QT + 1) / 2; ++kp) {
        u16x4 z = u16x4{0, 0, 0, 0};
        u16x4 lo = *(const u16x4*)(Kt + drow * Gm::RS + (2 * kp) * 16 + 4 * g);
        u16x4 hi = (2 * kp + 1 < Gm::QT) ? *(const u16x4*)(Kt + drow * Gm::RS + (2 * kp + 1) * 16 + 4 * g) : z;
        af[kp] = cat8(lo, hi);
      }
#pragma unroll
      for (int qt = 0; qt < Gm::QT; ++qt) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < (Gm::QT + 1) / 2; ++kp) {
          u16x4 hi = (2 * kp + 1 < Gm::QT) ? dsb[(2 * kp + 1 < Gm::QT) ? 2 * kp + 1 : 0][qt] : u16x4{0, 0, 0, 0};
          acc = mfma_16x16x32_bf16(af[kp], cat8(dsb[2 * kp][qt], hi), acc);
        }
        const int q = qt * 16 + li;
        if (d0 < DK && q < S) *(u16x4*)(p.dqkv + (tok0 + q) * LDG + hd * DK + d0) = pack4(acc);
      }
    }
  }
  wave_barrier();
  if (live) {
    // ---- dK^T[d][key] = sum_q Qt[d][q] dSt[key][q] ;  dV^T[dv][key] = sum_q dCt[dv][q] PmT[key][q] --------------
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const u16* At = which == 0 ? Qt : dCt;
      const u16* Bt = which == 0 ? dSt : PmT;
#pragma unroll
      for (int dt = 0; dt < (DK + 15) / 16; ++dt) {
        int drow = dt * 16 + li;
        drow = drow < DK ? drow : DK - 1;
        const int d0 = dt * 16 + 4 * g;
        u16x8 af[Gm::KS];
#pragma unroll
        for (int ks = 0; ks < Gm::KS; ++ks) af[ks] = ld8(At + drow * Gm::RS + ks * 32 + 8 * g);
#pragma unroll
        for (int kt = 0; kt < Gm::QT; ++kt) {
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < Gm::KS; ++ks)
            acc = mfma_16x16x32_bf16(af[ks], ld8(Bt + (kt * 16 + li) * Gm::RS + ks * 32 + 8 * g), acc);
          const int key = kt * 16 + li;
          if (d0 < DK && key < S)
            *(u16x4*)(p.dqkv + (tok0 + key) * LDG + (which == 0 ? KP : 2 * KP) + hd * DK + d0) = pack4(acc);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
struct AdditiveBwdParams {
  const u16* ctx;        // [n_seq*S][KP]  (forward input of the additive layer)
  const u16* Wap;        // [QP][KP]
  const float* bap;      // [QP]
  const float* qvp;      // [QP]
  const float* attn_w;   // [n_seq][S]
  const float* g_out;    // [n_seq][D]
  u16* dpre;             // [n_seq*S][QP] bf16
  float* dq_part;        // [gridDim.x][QP]  per-workgroup partial gradient of the query vector
  int64_t n_seq;
};

template <int S, int NSEQ>
__global__ __launch_bounds__(WG, 2) void additive_bwd_kernel(AdditiveBwdParams p) {
  using Gm = AddGeom<S, NSEQ>;
  NR_SMEM_DECL(smem);
  u16* Xs = (u16*)smem;
  float* dqp = (float*)(smem + Gm::X_BYTES);                 // [4][QP]   
  float* dsv = (float*)(smem + Gm::X_BYTES + 4 * QP * 4);    // [ROWS]
  const int tid = threadIdx.x, l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  const int64_t seq0 = (int64_t)blockIdx.x * NSEQ;
  const int64_t tok0 = seq0 * S, tok_total = p.n_seq * S;

  constexpr int PCS = XS / 8;
  for (int i = tid; i < Gm::ROWS * PCS; i += WG) {
    int r = i / PCS, c = i - r * PCS;
    u16x8 v = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (c < KP / 8 && r < Gm::TOK && tok0 + r < tok_total) v = *(const u16x8*)(p.ctx + (tok0 + r) * KP + c * 8);
    *(u16x8*)(Xs + r * XS + c * 8) = v;
  }
  for (int i = tid; i < 4 * QP; i += WG) dqp[i] = 0.0f;
  for (int i = tid; i < Gm::ROWS; i += WG) dsv[i] = 0.0f;
  __syncthreads();

  // ---- dw[tok] = g_out . x[tok];  ds = w * (dw - sum_s w dw)  (softmax backward) -------------------------------
  for (int seq = w; seq < NSEQ; seq += 4) {
    if (seq0 + seq >= p.n_seq) continue;
    const float* go = p.g_out + (seq0 + seq) * D;
    float mydw = 0.0f;           // lane s keeps dw of token s
    for (int s = 0; s < S; ++s) {
      const u16* xr = Xs + (seq * S + s) * XS;
      float a = 0.0f;
      for (int c = l; c < D; c += 64) a += go[c] * bf2f(xr[c]);
      a = wave_sum(a);
      if (l == s) mydw = a;
    }
    const float wt = l < S ? p.attn_w[(seq0 + seq) * S + l] : 0.0f;
    const float tot = wave_sum(wt * mydw);
    if (l < S) dsv[seq * S + l] = wt * (mydw - tot);
  }
  __syncthreads();

  // ---- recompute t = tanh(x Wa^T + ba); dpre = ds * qv * (1 - t^2); dq += ds * t ---------------------------------
  const int w_eff = (w + (int)blockIdx.x) & 3;
  for (int cg = 0; cg < (Gm::NTQ + 1) / 2; ++cg) {
    int G, mb, me;
    unit_range(Gm::NTQ, Gm::MT, w_eff, cg, G, mb, me);
    if (mb >= me) continue;
    f32x4 dqa[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto epi = [&](int j, int wr, int m, f32x4 acc) {
      f32x4 b4 = *(const f32x4*)(p.bap + wr + 4 * g);
      f32x4 q4 = *(const f32x4*)(p.qvp + wr + 4 * g);
      const int r_ = m * 16 + li;
      const float ds = dsv[r_];
      f32x4 dp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = fast_tanh(acc[r] + b4[r]);
        dp[r] = ds * q4[r] * (1.0f - t * t);
        dqa[j][r] += ds * t;
      }
      if (r_ < Gm::TOK && tok0 + r_ < tok_total) *(u16x4*)(p.dpre + (tok0 + r_) * QP + wr + 4 * g) = pack4(dp);
    };
    int wr0 = (2 * cg) * 16, wr1 = (2 * cg + 1) * 16;
    if (G == 2) {
      int wrow[2] = {wr0, wr1};
      proj_block<2, true>(p.Wap, wrow, Xs, mb, me, [&](int j, int m, f32x4 acc) { epi(j, wrow[j], m, acc); });
    } else {
      int wrow[1] = {wr0};
      proj_block<1, true>(p.Wap, wrow, Xs, mb, me, [&](int j, int m, f32x4 acc) { epi(0, wrow[0], m, acc); });
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
    p.dq_part[(int64_t)blockIdx.x * QP + n] = dqp[n] + dqp[QP + n] + dqp[2 * QP + n] + dqp[3 * QP + n];
}

// ---------------------------------------------------------------------------------------------------------------
// X materialisation for the weight-gradient GEMM: Xb[tok][0:D] = dropout1(table[ids[tok]]) (or dense x), Xb[tok][D] = 1,
// Xb[tok][D+1:KP] = 0.  One 16-B piece (8 bf16) per lane.
__global__ __launch_bounds__(256) void gather_bf16_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                          int64_t num_rows, const float* __restrict__ x_dense,
                                                          u16* __restrict__ Xb, int64_t n_tokens, DropCfg dc) {
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
        uint32_t keep = drop_keep4(dc, 1u, (uint64_t)tok * D4 + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = ((keep >> j) & 1u) ? x[j] * dc.scale : 0.0f;
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
// contributions to one table row are adjacent.  One wave reduces CH consecutive positions in registers (lane =
// one 4-column quad, lanes 0..10 a second quad) and writes each finished row once; only rows whose run touches
// the chunk boundary (and may continue in a neighbour wave) use atomics.  grad_table must start zeroed.
constexpr int SC_CH = 32;
__global__ __launch_bounds__(256) void embed_scatter_sorted_kernel(const int64_t* __restrict__ ids_sorted,
                                                                   const int64_t* __restrict__ perm, const u16* __restrict__ dx,
                                                                   int ldx, float* __restrict__ grad_table, int64_t num_rows,
                                                                   int64_t n_tokens, DropCfg dc) {
  const int l = lane_id();
  const int64_t s0 = ((int64_t)blockIdx.x * 4 + wave_id()) * SC_CH;
  if (s0 >= n_tokens) return;
  const int cnt = (int)((n_tokens - s0) < SC_CH ? (n_tokens - s0) : SC_CH);
  // each lane keeps one (id, token) pair of the chunk; broadcast by shuffle while walking the chunk
  int64_t my_id = -1, my_tok = 0;
  if (l < cnt) { my_id = ids_sorted[s0 + l]; my_tok = perm[s0 + l]; }
  int id_lo = (int)my_id, tok_lo = (int)my_tok;            // ids and token indices fit in 31 bits
  const int id_last = __builtin_bit_cast(int, shfl(__builtin_bit_cast(float, id_lo), cnt - 1));
  if (id_last <= 0) return;                                // chunk is all padding (sorted: ids <= 0 come first)
  const int id_before = s0 > 0 ? (int)ids_sorted[s0 - 1] : -1;
  const int id_after = s0 + cnt < n_tokens ? (int)ids_sorted[s0 + cnt] : -1;
  const bool two = l < (D4 - 64);
  f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
  int cur = __builtin_bit_cast(int, shfl(__builtin_bit_cast(float, id_lo), 0));
  auto flush = [&](int id, bool shared) {
    if (id <= 0 || id >= num_rows) return;
    float* dst = grad_table + ((int64_t)id * D4 + l) * 4;
    if (shared) {
#pragma unroll
      for (int j = 0; j < 4; ++j) atomic_add(dst + j, a0[j]);
      if (two) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atomic_add(dst + 256 + j, a1[j]);
      }
    } else {
      *(f32x4*)dst = a0;
      if (two) *(f32x4*)(dst + 256) = a1;
    }
  };
  bool first = true;
  for (int i = 0; i < cnt; ++i) {
    const int id = __builtin_bit_cast(int, shfl(__builtin_bit_cast(float, id_lo), i));
    const int tok = __builtin_bit_cast(int, shfl(__builtin_bit_cast(float, tok_lo), i));
    if (id != cur) {
      flush(cur, first && cur == id_before);
      first = false;
      cur = id;
      a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0;
    }
    if (id <= 0) continue;
    const u16* row = dx + (int64_t)tok * ldx;
    u16x4 v0 = *(const u16x4*)(row + l * 4);
    u16x4 v1 = two ? *(const u16x4*)(row + 256 + l * 4) : u16x4{0, 0, 0, 0};
    uint32_t k0 = 0xF, k1 = 0xF;
    float sc = 1.0f;
    if (dc.enabled) {
      k0 = drop_keep4(dc, 1u, (uint64_t)tok * D4 + l);
      k1 = two ? drop_keep4(dc, 1u, (uint64_t)tok * D4 + 64 + l) : 0u;
      sc = dc.scale;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((k0 >> j) & 1u) a0[j] += bf2f(v0[j]) * sc;
      if ((k1 >> j) & 1u) a1[j] += bf2f(v1[j]) * sc;
    }
  }
  flush(cur, (first && cur == id_before) || cur == id_after);
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
