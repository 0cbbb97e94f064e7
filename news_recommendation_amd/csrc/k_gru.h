// LSTUR user encoder: one-layer GRU over the click history (src/model/LSTUR/user_encoder.py:11-14,27-45; nn.GRU gate order
// r, z, n), one kernel launch per time step (the steps are sequentially dependent; the launches are queued back to back on
// one stream).  The input projection x W_ih^T of ALL steps is hoisted into one plain GEMM by the caller (gi, fp32).
//
// Padded layouts (Hd = hidden size, I = input size):
//   Hg = Hd rounded up to 16        gate stride: gate q of unit j lives at column / row q*Hg + j
//   Hp = (Hd+1) rounded up to 32    K extent of h rows; column Hd of every bf16 h row holds 1.0 (bias-gradient column)
//   Kp = 3*Hg rounded up to 32      K extent of the dGh rows
// Forward step t:   Gh = h_{t-1} W_hh^T on MFMA (A = W_hh rows of the unit tile for the three gates, B = h^T of 16 samples):
//   a lane ends up with 4 consecutive hidden units of one sample for all three gates -> the gate math is in-lane.
//   r = s(gi_r + b_ir + Gh_r + b_hr), z likewise, q = Gh_n + b_hn, n = tanh(gi_n + b_in + r q), h = (1-z) n + z h_{t-1};
//   samples with t >= length keep their state (pack_padded_sequence semantics: the first `length` slots are consumed).
// Backward step t:  dh_t = carry_{t+1} + dGh_{t+1} W_hh (MFMA, A = W_hh^T rows of the unit tile, B = dGh^T), then the gate
//   derivatives of step t for the same (sample, units) in the epilogue; writes dGi_t, dGh_t and carry_t = dh_t z_t.
#pragma once
#include "nr_common.h"

namespace nr {

__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// Workgroup -> (unit tile, sample group) mapping shared by both step kernels.  Workgroups are dealt round-robin to the 8 XCDs
// (linear id % 8), each with a private 4 MB L2.  W_hh (5 MB at Hd = 900) is the big operand and is the same at every step: XCD c
// takes the unit tiles c, c + 8, c + 16, ... for ALL sample groups, so that it reads (and keeps L2-resident from one step launch to
// the next) only its eighth of the weight rows; the hidden state of the step (1 MB) is what every XCD re-reads.  The 1-D grid holds
// 8 * ceil(n_tiles / 8) * n_groups workgroups; the ones past the last unit tile exit at once.
__device__ __forceinline__ bool gru_tile_of_wg(int n_tiles, int n_groups, int& tile, int& group) {
  const int L = (int)blockIdx.x, c = L & 7, s = L >> 3;
  tile = c + 8 * (s / n_groups);
  group = s % n_groups;
  return tile < n_tiles;
}
__host__ inline int gru_grid(int n_tiles, int n_groups) { return 8 * ((n_tiles + 7) / 8) * n_groups; }

struct GruFwdParams {
  const float* gi;        // [B*N][3*Hg] f32, row b*N + t  (x_t W_ih^T, no bias)
  const int* gi_row;      // optional [B*N]: the row of gi to take for (b, t) instead of b*N + t (evaluation: histories index a table of
                          // per-news projections, nr_gru_fwd_seq_rows)
  const u16* Whh;         // bf16 [3*Hg][Hp], tile order
  const float* b_ih;      // [3*Hd]
  const float* b_hh;      // [3*Hd]
  const int* len;         // [B], >= 1
  const u16* h_in_t;      // bf16 [ceil16(B)][Hp], tile order
  u16* h_out_b;           // bf16 [B][Hp] row-major (operand of the W_hh gradient GEMM and of the backward sweep), or null
  u16* h_out_t;           // bf16 [ceil16(B)][Hp], tile order: the next step's operand
  const float* h_in_f;    // f32 [B][Hp]   NEVER __restrict__: the row-indexed evaluation sweep (nr_gru_fwd_seq_rows) passes the SAME buffer as
  float* h_out_f;         // f32 [B][Hp]   h_in_f and h_out_f (state updated in place; each element is read and written by one lane only)
  u16* gates;             // training: bf16 [B][4][Hg] of this step (r, z, n, q), or null
  int B, N, Hd, Hg, Hp, t;
};

#ifndef NR_GRU_DEPTH
#define NR_GRU_DEPTH 8
#endif
// The step is latency-bound (PMC: 87 % of wave cycles waiting on memory with ~2 waves per SIMD): with a rolled k-loop every wave
// walks a chain of Hp/32 L2 round trips.  KS_CT > 0: the k-step count is a compile-time constant (Hp / 32 for the reference's
// hidden sizes); the loop is fully unrolled into a register pipeline with NR_GRU_DEPTH k-steps (4 x 16-byte loads each) in
// flight, pinned by sched_barriers (left alone the scheduler sinks every load next to its MFMA to save registers).
// KS_CT = 0: generic rolled loop.
// WLDS (experimental, NR_GRU_LDS=1, KS_CT > 0 only; one timing run: 50 forward steps 1.36 -> 1.18 ms, GPU parity suite pending): the workgroup's W_hh tile (3 gates x 16 units x
// Hp: 87 KB at Hd = 900, contiguous 1 KB fragments in tile order) is copied global -> LDS once (global_load_lds) and shared by the
// four waves, which otherwise each fetch their own copy from L2: 136 -> ~59 MB of L2 -> L1 traffic per step at B = 512.
// The workgroup's W_hh tile (3 gates x 16 units x Hp, contiguous 1 KB fragments in tile order) global -> LDS; visible after a barrier.
template <int KS_CT>
__device__ __forceinline__ void gru_fwd_fetch_w(const GruFwdParams& p, int tile) {
  NR_SMEM_DECL(smem);
  const int l = lane_id(), w = wave_id();
  for (int blk = w; blk < 3 * KS_CT; blk += 4) {              // block (gate q, k-step ks) = 1 KB, lane-linear
    const int q = blk / KS_CT, ks = blk - q * KS_CT;
    NR_GLDS16(p.Whh + ((size_t)q * p.Hg + tile * 16) * p.Hp + ks * 512 + l * 8, smem + blk * 1024);
  }
}

// One step for the workgroup (unit tile, sample group).
template <int KS_CT, int NB, bool WLDS>
__device__ __forceinline__ void gru_fwd_step_impl(const GruFwdParams& p, int tile, int group) {
  // NB sample tiles per wave share every W_hh fragment (NB = 2: 5 loads per 6 MFMAs instead of 4 per 3; the step is bound by the
  // L2 -> L1 operand volume, ~15 TB/s measured)
  const int l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  const int s0r = (group * 4 + w) * NB * 16;
  const bool idle = s0r >= p.B;                                // no samples left for this wave
  const int s0 = idle ? 0 : s0r;                               // (an idle wave of a WLDS workgroup stays for the barrier and recomputes tile 0, discarded)
  if constexpr (WLDS && KS_CT > 0) gru_fwd_fetch_w<KS_CT>(p, tile);
  if (idle && !(WLDS && KS_CT > 0)) return;
  const int j0 = tile * 16, jb = j0 + 4 * g;
  // fragment pointers in tile order: k-step ks of an operand tile is the 1 KB at + ks * 512 elements
  const u16* w0 = p.Whh + (size_t)tile * p.Hp * 16 + l * 8;
  const u16* w1 = w0 + (size_t)p.Hg * p.Hp;
  const u16* w2 = w1 + (size_t)p.Hg * p.Hp;
  const u16* hp[NB];
  int sb[NB], len_s[NB];
  f32x4 gir[NB], giz[NB], gin[NB], ho[NB];
  f32x4 ar[NB], az[NB], an[NB];
  // epilogue operands (lane: sample s0 + 16 nb + li, units j0 + 4g + r) are requested before the k pipeline so that their round
  // trip overlaps it; raw values: nothing below may wait on them before the pipeline has been issued
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int st = s0 + nb * 16 < p.B ? s0 + nb * 16 : s0;          // a missing second tile recomputes the first (results discarded)
    hp[nb] = p.h_in_t + (size_t)(st >> 4) * p.Hp * 16 + l * 8;
    sb[nb] = st + li < p.B ? st + li : p.B - 1;
    len_s[nb] = p.len[sb[nb]];
    const size_t girow = p.gi_row != nullptr ? (size_t)p.gi_row[(size_t)sb[nb] * p.N + p.t] : (size_t)sb[nb] * p.N + p.t;
    const float* gi = p.gi + girow * 3 * p.Hg + jb;
    gir[nb] = *(const f32x4*)gi; giz[nb] = *(const f32x4*)(gi + p.Hg); gin[nb] = *(const f32x4*)(gi + 2 * p.Hg);
    ho[nb] = *(const f32x4*)(p.h_in_f + (size_t)sb[nb] * p.Hp + jb);
    ar[nb] = f32x4{0.f, 0.f, 0.f, 0.f}; az[nb] = ar[nb]; an[nb] = ar[nb];
  }
  f32x4 b_ir, b_hr, b_iz, b_hz, b_in, b_hn;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = jb + r < p.Hd ? jb + r : p.Hd - 1;
    b_ir[r] = p.b_ih[j]; b_hr[r] = p.b_hh[j];
    b_iz[r] = p.b_ih[p.Hd + j]; b_hz[r] = p.b_hh[p.Hd + j];
    b_in[r] = p.b_ih[2 * p.Hd + j]; b_hn[r] = p.b_hh[2 * p.Hd + j];
  }
  if (KS_CT > 0 && WLDS) {
    NR_SMEM_DECL(smem);
    const u16* wl = (const u16*)smem + l * 8;                  // fragment (q, ks) at + (q * KS_CT + ks) * 512
    constexpr int D = NR_GRU_DEPTH;
    u16x8 fh[NB][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < KS_CT) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) fh[nb][i] = *(const u16x8*)(hp[nb] + i * 512);
      }
    __syncthreads();                                           // drains the global -> LDS copies of all four waves
    u16x8 a0 = *(const u16x8*)wl, a1 = *(const u16x8*)(wl + KS_CT * 512), a2 = *(const u16x8*)(wl + 2 * KS_CT * 512);
    NR_SCHED_BARRIER();
#pragma unroll
    for (int ks = 0; ks < KS_CT; ++ks) {
      const int sl = ks % D, kn = ks + 1 < KS_CT ? ks + 1 : ks;
      const u16x8 n0 = *(const u16x8*)(wl + kn * 512), n1 = *(const u16x8*)(wl + (KS_CT + kn) * 512),
                  n2 = *(const u16x8*)(wl + (2 * KS_CT + kn) * 512);          // next k-step's fragments in flight during the MFMAs
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        ar[nb] = mfma_16x16x32_bf16(a0, fh[nb][sl], ar[nb]);
        az[nb] = mfma_16x16x32_bf16(a1, fh[nb][sl], az[nb]);
        an[nb] = mfma_16x16x32_bf16(a2, fh[nb][sl], an[nb]);
      }
      if (ks + D < KS_CT) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) fh[nb][sl] = *(const u16x8*)(hp[nb] + (ks + D) * 512);
      }
      a0 = n0; a1 = n1; a2 = n2;
      NR_SCHED_BARRIER();
    }
  } else if (KS_CT > 0) {
    constexpr int D = NB == 1 ? NR_GRU_DEPTH : (NR_GRU_DEPTH * 3) / 4;
    u16x8 fh[NB][D], f0[D], f1[D], f2[D];
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < KS_CT) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) fh[nb][i] = *(const u16x8*)(hp[nb] + i * 512);
        f0[i] = *(const u16x8*)(w0 + i * 512);
        f1[i] = *(const u16x8*)(w1 + i * 512);
        f2[i] = *(const u16x8*)(w2 + i * 512);
      }
    NR_SCHED_BARRIER();
#pragma unroll
    for (int ks = 0; ks < KS_CT; ++ks) {
      const int sl = ks % D;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        ar[nb] = mfma_16x16x32_bf16(f0[sl], fh[nb][sl], ar[nb]);
        az[nb] = mfma_16x16x32_bf16(f1[sl], fh[nb][sl], az[nb]);
        an[nb] = mfma_16x16x32_bf16(f2[sl], fh[nb][sl], an[nb]);
      }
      if (ks + D < KS_CT) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) fh[nb][sl] = *(const u16x8*)(hp[nb] + (ks + D) * 512);
        f0[sl] = *(const u16x8*)(w0 + (ks + D) * 512);
        f1[sl] = *(const u16x8*)(w1 + (ks + D) * 512);
        f2[sl] = *(const u16x8*)(w2 + (ks + D) * 512);
      }
      NR_SCHED_BARRIER();
    }
  } else {
    const int ksteps = p.Hp / 32;
#pragma unroll 2
    for (int ks = 0; ks < ksteps; ++ks) {
      const u16x8 a0 = *(const u16x8*)(w0 + ks * 512), a1 = *(const u16x8*)(w1 + ks * 512), a2 = *(const u16x8*)(w2 + ks * 512);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const u16x8 hf = *(const u16x8*)(hp[nb] + ks * 512);
        ar[nb] = mfma_16x16x32_bf16(a0, hf, ar[nb]);
        az[nb] = mfma_16x16x32_bf16(a1, hf, az[nb]);
        an[nb] = mfma_16x16x32_bf16(a2, hf, an[nb]);
      }
    }
  }
  if (idle) return;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int s = s0 + nb * 16 + li;
    if (s >= p.B) continue;
    const bool active = p.t < len_s[nb];
    f32x4 hn, rr, zz, nn, qq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = jb + r < p.Hd;                    // units >= Hd: zero weight rows and gi columns, their h stays 0
      rr[r] = fast_sigmoid(gir[nb][r] + b_ir[r] + ar[nb][r] + b_hr[r]);
      zz[r] = fast_sigmoid(giz[nb][r] + b_iz[r] + az[nb][r] + b_hz[r]);
      qq[r] = an[nb][r] + b_hn[r];
      nn[r] = fast_tanh(gin[nb][r] + b_in[r] + rr[r] * qq[r]);
      const float v = active ? (1.0f - zz[r]) * nn[r] + zz[r] * ho[nb][r] : ho[nb][r];
      hn[r] = ok ? v : 0.0f;
    }
    if (jb < p.Hd) {         // the 4 units straddle Hd only at the very end, where the row padding absorbs them
      *(f32x4*)(p.h_out_f + (size_t)s * p.Hp + jb) = hn;
      u16x4 hb = pack4(hn);
      if (jb <= p.Hd && p.Hd < jb + 4) hb[p.Hd - jb] = 0x3F80;        // column Hd = 1.0
      *(u16x4*)(p.h_out_t + tile_off(s, jb, p.Hp)) = hb;           // the tile-order state: the next step's MFMA operand of every unit tile
      if (p.h_out_b != nullptr) *(u16x4*)(p.h_out_b + (size_t)s * p.Hp + jb) = hb;
      if (jb + 4 == p.Hd) {                                          // Hd % 4 == 0: the lane owning the last units also sets column Hd
        p.h_out_t[tile_off(s, p.Hd, p.Hp)] = 0x3F80;
        if (p.h_out_b != nullptr) p.h_out_b[(size_t)s * p.Hp + p.Hd] = 0x3F80;
      }
    }
    if (p.gates != nullptr && jb < p.Hg) {
      u16* gp = p.gates + (size_t)s * 4 * p.Hg + jb;
      *(u16x4*)gp = pack4(rr);
      *(u16x4*)(gp + p.Hg) = pack4(zz);
      *(u16x4*)(gp + 2 * p.Hg) = pack4(nn);
      *(u16x4*)(gp + 3 * p.Hg) = pack4(qq);
    }
  }
}

template <int KS_CT, int NB, bool WLDS = false>
__global__ __launch_bounds__(WG) void gru_fwd_step_kernel(GruFwdParams p) {
  int tile, group;
  if (!gru_tile_of_wg(p.Hg / 16, (p.B + 64 * NB - 1) / (64 * NB), tile, group)) return;
  gru_fwd_step_impl<KS_CT, NB, WLDS>(p, tile, group);
}

// (A persistent whole-sequence form of both sweeps -- W_hh tile resident in LDS, a grid-wide barrier between the steps -- was built in round 2,
// bit-identical, and measured 0.3-0.4 ms per LSTUR step SLOWER in every form tried: back-to-back launches of a 15 us kernel cost ~3-4 us of gap
// each on this stack, a grid barrier across 228 workgroups with non-coherent L2s ~7 us.  Removed in round 4; the record is DESIGN.md 5.3.)

struct GruBwdParams {
  const float* g_last;     // [B][Hd] gradient of the returned hidden state (used when first != 0)
  const u16* dgh_next;     // bf16 [ceil16(B)][Kp], tile order: dGh of step t+1
  const float* carry_next; // f32 [B][Hp]: dh_{t+1} z_{t+1} (or dh_{t+1} for finished samples)
  const u16* WhhT;         // bf16 [Hp][Kp], tile order: WhhT[j][q*Hg + i] = W_hh[q*Hd + i][j]
  const u16* gates;        // bf16 [B][4][Hg] of step t; null for the final call (t = -1: only dh_0 is produced)
  const u16* h_prev_b;     // bf16 [B][Hp] = h_{t-1}
  const int* len;
  u16* dgi;                // bf16 [B*N][Kp]: row b*N + t receives [dr_pre | dz_pre | dn_pre]
  u16* dgh;                // bf16 [B][Kp] of step t: [dr_pre | dz_pre | dn_pre * r], row-major (operand of the W_hh gradient GEMM)
  u16* dgh_t;              // the same in tile order [ceil16(B)][Kp]: the next launch's operand
  float* carry;            // f32 [B][Hp] of step t; for t = -1 it receives dh_0
  int B, N, Hd, Hg, Hp, Kp, t, first;
};

// Gate derivatives of step t for lane (sample s, units jb .. jb+3) given dh_t; shared by the step kernels.
__device__ __forceinline__ void gru_bwd_finish(const GruBwdParams& p, int s, int jb, f32x4 dh, int len_s, u16x4 rb, u16x4 zb, u16x4 nb,
                                               u16x4 qb, u16x4 hb) {
  if (p.gates == nullptr) {                    // t = -1: dh_0
    *(f32x4*)(p.carry + (size_t)s * p.Hp + jb) = dh;
    return;
  }
  const bool active = p.t < len_s;
  f32x4 d_r = f32x4{0.f, 0.f, 0.f, 0.f}, d_z = d_r, d_n = d_r, d_nr = d_r, cy = dh;
  if (active) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = jb + r < p.Hd;
      const float rr = bf2f(rb[r]), zz = bf2f(zb[r]), nn = bf2f(nb[r]), qq = bf2f(qb[r]), hp = bf2f(hb[r]);
      const float dn = dh[r] * (1.0f - zz);
      const float dz = dh[r] * (hp - nn);
      const float dnp = dn * (1.0f - nn * nn);
      const float drp = dnp * qq * rr * (1.0f - rr);
      const float dzp = dz * zz * (1.0f - zz);
      d_r[r] = ok ? drp : 0.0f;
      d_z[r] = ok ? dzp : 0.0f;
      d_n[r] = ok ? dnp : 0.0f;
      d_nr[r] = ok ? dnp * rr : 0.0f;
      cy[r] = dh[r] * zz;
    }
  }
  *(f32x4*)(p.carry + (size_t)s * p.Hp + jb) = cy;
  u16* gi = p.dgi + ((size_t)s * p.N + p.t) * p.Kp + jb;
  *(u16x4*)gi = pack4(d_r);
  *(u16x4*)(gi + p.Hg) = pack4(d_z);
  *(u16x4*)(gi + 2 * p.Hg) = pack4(d_n);
  const u16x4 pr = pack4(d_r), pz = pack4(d_z), pn = pack4(d_nr);
  u16* gh = p.dgh + (size_t)s * p.Kp + jb;
  *(u16x4*)gh = pr;
  *(u16x4*)(gh + p.Hg) = pz;
  *(u16x4*)(gh + 2 * p.Hg) = pn;
  *(u16x4*)(p.dgh_t + tile_off(s, jb, p.Kp)) = pr;               // tile-order copy: the next launch's MFMA operand
  *(u16x4*)(p.dgh_t + tile_off(s, p.Hg + jb, p.Kp)) = pz;
  *(u16x4*)(p.dgh_t + tile_off(s, 2 * p.Hg + jb, p.Kp)) = pn;
}

template <int KS_CT>
__global__ __launch_bounds__(WG) void gru_bwd_step_kernel(GruBwdParams p) {
  const int l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  int tile, group;
  if (!gru_tile_of_wg(p.Hg / 16, (p.B + 63) / 64, tile, group)) return;
  const int s0 = (group * 4 + w) * 16;
  if (s0 >= p.B) return;
  const int j0 = tile * 16;
  const int sb = s0 + li < p.B ? s0 + li : p.B - 1;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  // epilogue operands of lane (sample s0 + li, units j0 + 4g + r), requested ahead of the k pipeline
  const int jb = j0 + 4 * g, jc = jb < p.Hg ? jb : 0;
  const int len_s = p.len[sb];
  f32x4 cn = acc;
  u16x4 rb = u16x4{0, 0, 0, 0}, zb = rb, nb = rb, qb = rb, hb = rb;
  if (!p.first) cn = *(const f32x4*)(p.carry_next + (size_t)sb * p.Hp + jc);
  if (p.gates != nullptr) {
    const u16* gp = p.gates + (size_t)sb * 4 * p.Hg + jc;
    rb = *(const u16x4*)gp; zb = *(const u16x4*)(gp + p.Hg); nb = *(const u16x4*)(gp + 2 * p.Hg); qb = *(const u16x4*)(gp + 3 * p.Hg);
    hb = *(const u16x4*)(p.h_prev_b + (size_t)sb * p.Hp + jc);
  }
  if (KS_CT > 0 && !p.first) {
    // same register pipeline as the forward step (see there): 2 x 16-byte loads per k-step, 2 * NR_GRU_DEPTH k-steps in flight
    constexpr int D = 2 * NR_GRU_DEPTH;
    const u16* dp = p.dgh_next + (size_t)(s0 >> 4) * p.Kp * 16 + l * 8;
    const u16* wp = p.WhhT + (size_t)tile * p.Kp * 16 + l * 8;
    u16x8 fd[D], fw[D];
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < KS_CT) {
        fw[i] = *(const u16x8*)(wp + i * 512);
        fd[i] = *(const u16x8*)(dp + i * 512);
      }
    NR_SCHED_BARRIER();
    f32x4 a1 = acc, a2 = acc, a3 = acc;
#pragma unroll
    for (int ks = 0; ks < KS_CT; ++ks) {
      const int sl = ks % D;
      if ((ks & 3) == 0) acc = mfma_16x16x32_bf16(fw[sl], fd[sl], acc);
      else if ((ks & 3) == 1) a1 = mfma_16x16x32_bf16(fw[sl], fd[sl], a1);
      else if ((ks & 3) == 2) a2 = mfma_16x16x32_bf16(fw[sl], fd[sl], a2);
      else a3 = mfma_16x16x32_bf16(fw[sl], fd[sl], a3);
      if (ks + D < KS_CT) {
        fw[sl] = *(const u16x8*)(wp + (ks + D) * 512);
        fd[sl] = *(const u16x8*)(dp + (ks + D) * 512);
      }
      NR_SCHED_BARRIER();
    }
    acc = (acc + a1) + (a2 + a3);
  } else if (!p.first) {
    const u16* dp = p.dgh_next + (size_t)(s0 >> 4) * p.Kp * 16 + l * 8;
    const u16* wp = p.WhhT + (size_t)tile * p.Kp * 16 + l * 8;
    const int ksteps = p.Kp / 32;
    // four independent accumulators: a single one would serialise the 86 MFMAs of the K loop on the accumulate dependency
    f32x4 a1 = acc, a2 = acc, a3 = acc;
    int ks = 0;
    for (; ks + 4 <= ksteps; ks += 4) {
      acc = mfma_16x16x32_bf16(*(const u16x8*)(wp + ks * 512), *(const u16x8*)(dp + ks * 512), acc);
      a1 = mfma_16x16x32_bf16(*(const u16x8*)(wp + (ks + 1) * 512), *(const u16x8*)(dp + (ks + 1) * 512), a1);
      a2 = mfma_16x16x32_bf16(*(const u16x8*)(wp + (ks + 2) * 512), *(const u16x8*)(dp + (ks + 2) * 512), a2);
      a3 = mfma_16x16x32_bf16(*(const u16x8*)(wp + (ks + 3) * 512), *(const u16x8*)(dp + (ks + 3) * 512), a3);
    }
    for (; ks < ksteps; ++ks) acc = mfma_16x16x32_bf16(*(const u16x8*)(wp + ks * 512), *(const u16x8*)(dp + ks * 512), acc);
    acc = (acc + a1) + (a2 + a3);
  }
  const int s = s0 + li;
  if (s >= p.B || jb >= p.Hg) return;
  f32x4 dh;
  if (p.first) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[r] = jb + r < p.Hd ? p.g_last[(size_t)s * p.Hd + jb + r] : 0.0f;
  } else {
    dh = acc + cn;
  }
  gru_bwd_finish(p, s, jb, dh, len_s, rb, zb, nb, qb, hb);
}

// Experimental (NR_GRU_LDS=1, Hd = 900 / 450; one timing run: 21.4 -> 18.7 us per step, GPU parity suite pending): backward step with the workgroup's W_hh^T tile (16 rows
// x Kp: 86 KB at Hd = 900) copied global -> LDS once and shared by the four waves, each of which takes TWO sample tiles (so that 57 x 4
// = 228 workgroups cover B = 512 in one round at one workgroup per CU).  L2 -> L1 traffic per step 321 -> ~165 MB.
template <int KS_CT>
__device__ __forceinline__ void gru_bwd_fetch_w(const GruBwdParams& p, int tile) {
  NR_SMEM_DECL(smem);
  const int l = lane_id(), w = wave_id();
  for (int blk = w; blk < KS_CT; blk += 4) NR_GLDS16(p.WhhT + (size_t)tile * p.Kp * 16 + blk * 512 + l * 8, smem + blk * 1024);
}

template <int KS_CT>
__device__ __forceinline__ void gru_bwd_step_lds_impl(const GruBwdParams& p, int tile, int group) {
  constexpr int NB = 2;
  NR_SMEM_DECL(smem);
  const int l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  const int s0r = (group * 4 + w) * NB * 16;
  const bool idle = s0r >= p.B;
  const int s0 = idle ? 0 : s0r;
  const int j0 = tile * 16, jb = j0 + 4 * g, jc = jb < p.Hg ? jb : 0;
  if (!p.first) gru_bwd_fetch_w<KS_CT>(p, tile);
  // epilogue operands of both tiles, requested ahead of the k pipeline
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const u16x4 z4 = u16x4{0, 0, 0, 0};
  int sb[NB], len_s[NB];
  f32x4 cn[NB], acc[NB][2];
  u16x4 rb[NB], zb[NB], nb_[NB], qb[NB], hb[NB];
  const u16* dp[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int st = s0 + nb * 16 < p.B ? s0 + nb * 16 : s0;
    sb[nb] = st + li < p.B ? st + li : p.B - 1;
    len_s[nb] = p.len[sb[nb]];
    cn[nb] = zero4; acc[nb][0] = zero4; acc[nb][1] = zero4;
    rb[nb] = z4; zb[nb] = z4; nb_[nb] = z4; qb[nb] = z4; hb[nb] = z4;
    if (!p.first) cn[nb] = *(const f32x4*)(p.carry_next + (size_t)sb[nb] * p.Hp + jc);
    if (p.gates != nullptr) {
      const u16* gp = p.gates + (size_t)sb[nb] * 4 * p.Hg + jc;
      rb[nb] = *(const u16x4*)gp; zb[nb] = *(const u16x4*)(gp + p.Hg); nb_[nb] = *(const u16x4*)(gp + 2 * p.Hg); qb[nb] = *(const u16x4*)(gp + 3 * p.Hg);
      hb[nb] = *(const u16x4*)(p.h_prev_b + (size_t)sb[nb] * p.Hp + jc);
    }
    dp[nb] = p.dgh_next + (size_t)(st >> 4) * p.Kp * 16 + l * 8;
  }
  if (!p.first) {
    constexpr int D = NR_GRU_DEPTH;
    u16x8 fd[NB][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i < KS_CT) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) fd[nb][i] = *(const u16x8*)(dp[nb] + i * 512);
      }
    __syncthreads();                                           // drains the global -> LDS copies of all four waves
    const u16* wl = (const u16*)smem + l * 8;
    u16x8 a = *(const u16x8*)wl;
    NR_SCHED_BARRIER();
#pragma unroll
    for (int ks = 0; ks < KS_CT; ++ks) {
      const int sl = ks % D;
      const u16x8 an = *(const u16x8*)(wl + (ks + 1 < KS_CT ? ks + 1 : ks) * 512);      // next fragment in flight during the MFMAs
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb][ks & 1] = mfma_16x16x32_bf16(a, fd[nb][sl], acc[nb][ks & 1]);   // 4 independent chains
      if (ks + D < KS_CT) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) fd[nb][sl] = *(const u16x8*)(dp[nb] + (ks + D) * 512);
      }
      a = an;
      NR_SCHED_BARRIER();
    }
  }
  if (idle || jb >= p.Hg) return;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int s = s0 + nb * 16 + li;
    if (s >= p.B) continue;
    f32x4 dh;
    if (p.first) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[r] = jb + r < p.Hd ? p.g_last[(size_t)s * p.Hd + jb + r] : 0.0f;
    } else {
      dh = (acc[nb][0] + acc[nb][1]) + cn[nb];
    }
    gru_bwd_finish(p, s, jb, dh, len_s[nb], rb[nb], zb[nb], nb_[nb], qb[nb], hb[nb]);
  }
}

template <int KS_CT>
__global__ __launch_bounds__(WG) void gru_bwd_step_lds_kernel(GruBwdParams p) {
  int tile, group;
  if (!gru_tile_of_wg(p.Hg / 16, (p.B + 127) / 128, tile, group)) return;
  gru_bwd_step_lds_impl<KS_CT>(p, tile, group);
}

// ---- operand packing --------------------------------------------------------------------------------------------------
// Gate stage of one inference step for LARGE batches (evaluation: tens of thousands of histories at once).  The fused step kernel above
// gives a workgroup 16 hidden units, so 57 workgroups per sample group each fetch the group's state rows: fine at B = 512, but at
// B = 73,000 that is 7.7 GB of operand traffic per step.  There the recurrent product Gh = h W_hh^T runs as a plain library GEMM (its tiles
// re-read the state 11 x) and this kernel does the rest: r, z, n, the state update -- the formulas and intrinsics of gru_fwd_step_impl -- on
// (gi via the row index, gh), writing the fp32 state in place and its bf16 copy (the next step's GEMM operand; column Hd = 1.0 as
// everywhere).  One lane = four consecutive units of one sample.
struct GruGateParams {
  const float* gi;        // [n_rows][3*Hg] f32
  const int* gi_row;      // [B][N]
  const float* gh;        // [B][3*Hg] f32 = bf16(h) @ bf16(W_hh)^T of this step
  const float* b_ih;      // [3*Hd]
  const float* b_hh;      // [3*Hd]
  const int* len;         // [B]
  float* h_f;             // [B][Hp] f32 state, in place
  u16* h_b;               // [B][Hp] bf16 copy of the new state
  int B, N, Hd, Hg, Hp, t;
};

__global__ __launch_bounds__(256) void gru_gate_rows_kernel(GruGateParams p) {
  const int qpr = p.Hg / 4;                                   // unit quads per sample
  const int64_t total = (int64_t)p.B * qpr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(i / qpr), jb = (int)(i - (int64_t)s * qpr) * 4;
    if (jb >= p.Hd) continue;                                 // (a quad may straddle Hd -- Hd = 450 --: its units >= Hd stay 0, like the step kernel's)
    const bool active = p.t < p.len[s];
    const float* gi = p.gi + (size_t)p.gi_row[(size_t)s * p.N + p.t] * 3 * p.Hg + jb;
    const float* gh = p.gh + (size_t)s * 3 * p.Hg + jb;
    const f32x4 gir = *(const f32x4*)gi, giz = *(const f32x4*)(gi + p.Hg), gin = *(const f32x4*)(gi + 2 * p.Hg);
    const f32x4 ar = *(const f32x4*)gh, az = *(const f32x4*)(gh + p.Hg), an = *(const f32x4*)(gh + 2 * p.Hg);
    const f32x4 ho = *(const f32x4*)(p.h_f + (size_t)s * p.Hp + jb);
    f32x4 hn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = jb + r < p.Hd;
      const int j = ok ? jb + r : p.Hd - 1;
      const float rr = fast_sigmoid(gir[r] + p.b_ih[j] + ar[r] + p.b_hh[j]);
      const float zz = fast_sigmoid(giz[r] + p.b_ih[p.Hd + j] + az[r] + p.b_hh[p.Hd + j]);
      const float qq = an[r] + p.b_hh[2 * p.Hd + j];
      const float nn = fast_tanh(gin[r] + p.b_ih[2 * p.Hd + j] + rr * qq);
      const float v = active ? (1.0f - zz) * nn + zz * ho[r] : ho[r];
      hn[r] = ok ? v : 0.0f;
    }
    *(f32x4*)(p.h_f + (size_t)s * p.Hp + jb) = hn;
    u16x4 hb = pack4(hn);
    if (jb <= p.Hd && p.Hd < jb + 4) hb[p.Hd - jb] = 0x3F80;           // column Hd = 1.0
    *(u16x4*)(p.h_b + (size_t)s * p.Hp + jb) = hb;
    if (jb + 4 == p.Hd) p.h_b[(size_t)s * p.Hp + p.Hd] = 0x3F80;
  }
}

// W f32 [3*Hd][K] (nn.GRU weight_ih_l0 / weight_hh_l0) -> dst bf16 [3*Hg][Kpad]: row q*Hg + j = W[q*Hd + j][:], zero padded.
// dstT (optional) bf16 [Kpad_rows = Hp][Kp]: dstT[k][q*Hg + j] = W[q*Hd + j][k]   (the data-gradient operand).
// tiled != 0: both in tile order (step-kernel operands); 0: row-major (GEMM operand).
__global__ __launch_bounds__(256) void pack_gru_kernel(const float* __restrict__ W, int Hd, int K, int Hg, int Kpad, u16* __restrict__ dst,
                                                       u16* __restrict__ dstT, int Trows, int Kp, int tiled) {
  const int n1 = 3 * Hg * Kpad;
  const int n2 = dstT ? Trows * Kp : 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += gridDim.x * blockDim.x) {
    if (i < n1) {
      const int row = i / Kpad, k = i - row * Kpad;
      const int q = row / Hg, j = row - q * Hg;
      dst[tiled ? tile_off(row, k, Kpad) : (size_t)i] = f2bf((j < Hd && k < K) ? W[((size_t)q * Hd + j) * K + k] : 0.0f);
    } else {
      const int i2 = i - n1;
      const int k = i2 / Kp, c = i2 - k * Kp;
      const int q = c / Hg, j = c - q * Hg;
      dstT[tiled ? tile_off(k, c, Kp) : (size_t)i2] = f2bf((q < 3 && j < Hd && k < K) ? W[((size_t)q * Hd + j) * K + k] : 0.0f);
    }
  }
}

// The tile-order case (tiled != 0: W_hh and its transpose, re-packed after every optimiser step) by OUTPUT piece: one lane produces one 16-byte
// piece of the tile-order image -- eight consecutive columns of one row -- and consecutive lanes take consecutive pieces, i.e. the 16 rows of a
// tile: stores are contiguous, and the reads of the transpose (element (k, c) = W[c][k]) touch 64 contiguous bytes per column across the 16 lanes
// of a row group.  The element-per-lane form above walks the transpose column-fastest: every read of a wave in a different row of W (46 us per
// LSTUR step for the two 2,736 x 928 images).
__global__ __launch_bounds__(256) void pack_gru_tiled_kernel(const float* __restrict__ W, int Hd, int K, int Hg, int Kpad, u16* __restrict__ dst,
                                                             u16* __restrict__ dstT, int Trows, int Kp) {
  const int n1 = 3 * Hg * Kpad / 8;                       // pieces of dst  [3 Hg][Kpad]
  const int n2 = dstT ? Trows * Kp / 8 : 0;               // pieces of dstT [Trows][Kp]
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += gridDim.x * blockDim.x) {
    const bool tr = i >= n1;
    const int pi = tr ? i - n1 : i;
    const int ncol = tr ? Kp : Kpad;
    const int blk = pi >> 6, lane = pi & 63;              // 64 pieces per 16 x 32 block
    const int kb = ncol >> 5;
    const int r = (blk / kb) * 16 + (lane & 15);          // row of the image
    const int c0 = (blk % kb) * 32 + (lane >> 4) * 8;     // its first column
    u16x8 o;
    if (!tr) {                                            // dst[row][k] = W[q Hd + j][k], row = q Hg + j
      const int q = r / Hg, j = r - q * Hg;
      const float* src = W + ((size_t)q * Hd + j) * K;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf((j < Hd && c0 + e < K) ? src[c0 + e] : 0.0f);
    } else {                                              // dstT[k][c] = W[q Hd + j][k], c = q Hg + j
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + e, q = c / Hg, j = c - q * Hg;
        o[e] = f2bf((q < 3 && j < Hd && r < K) ? W[((size_t)q * Hd + j) * K + r] : 0.0f);
      }
    }
    *(u16x8*)((tr ? dstT : dst) + (size_t)pi * 8) = o;
  }
}

// f32 rows [n][d] (row stride lds_) -> bf16 rows [n][dp]: cols < d converted, col d = 1.0 (when d < dp), rest 0.
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* __restrict__ src, int64_t lds_, int d, u16* __restrict__ dst, int dp,
                                                           int64_t n) {
  const int64_t total = n * dp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / dp;
    const int c = (int)(i - r * dp);
    dst[i] = c < d ? f2bf(src[r * lds_ + c]) : (u16)(c == d ? 0x3F80 : 0);
  }
}

// the same, four columns per lane (d, dp, the row stride and both base addresses multiples of 4 elements: every case of the training steps).  The
// element-per-lane form above spends a 64-bit division per element: 52 us for LSTUR's [25,600][900] history block (139 MB moved), this one is
// bound by the bytes.
__global__ __launch_bounds__(256) void rows_to_bf16_v4_kernel(const float* __restrict__ src, int64_t lds_, int d, u16* __restrict__ dst, int dp,
                                                              int64_t n) {
  const int q = dp >> 2;
  const int64_t total = n * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) * 4;
    u16x4 o = u16x4{0, 0, 0, 0};
    if (c < d) {
      const f32x4 v = *(const f32x4*)(src + r * lds_ + c);
      o = pack4(v);
    } else if (c == d) o[0] = 0x3F80;
    *(u16x4*)(dst + r * (int64_t)dp + c) = o;
  }
}

// bf16 rows [n][K] row-major -> tile order [ceil16(n)][K] (rows >= n zero).
__global__ __launch_bounds__(256) void tile_rows_kernel(const u16* __restrict__ src, int n, int K, u16* __restrict__ dst) {
  const int64_t total = (int64_t)((n + 15) & ~15) * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / K), k = (int)(i - (int64_t)r * K);
    dst[tile_off(r, k, K)] = r < n ? src[i] : (u16)0;
  }
}

}  // namespace nr
