// LSTUR user encoder, PERSISTENT form of the GRU sweeps for gfx950 (round 5): one launch per sweep instead of one per time step
// (src/model/LSTUR/user_encoder.py:30-37: nn.GRU over the packed click history; autograd for the backward sweep).
//
// The step kernels (k_gru.h) re-fetch their W_hh tile (87 KB per workgroup) from the L2 at every step and pay a kernel boundary per step:
// 12.9 us x 50 forward + 17.8 us x 51 backward = 28 % of the LSTUR step at 0.05 - 0.08 of the MFMA peak.  A chip-wide persistent form was built
// in round 2 and lost: a grid barrier has to publish the new state across eight non-coherent L2s (write-back + invalidate per step).
// This form never crosses an L2:
//   * the BATCH is split over the XCDs (64 samples each at B = 512); the state of those samples is produced and consumed by the 32 CUs of
//     one XCD, i.e. through ONE L2: plain stores + vmcnt drain on the producer, L1-bypassing global -> LDS copies on the consumer, one L2 atomic
//     + relaxed poll per workgroup and step (csrc/k_xcd.h) -- no cache maintenance;
//   * W_hh never moves: CU slot s of an XCD owns unit tiles s and s + 32 (of 57), and the three gates' rows of a tile live in the REGISTERS of
//     three waves as MFMA A fragments (29 k-steps x 4 registers = 116 per lane) for the whole sweep: 5.08 MB of weights = 6 x 116 registers x
//     64 lanes x 32 CUs, loaded once per sweep;
//   * per step a workgroup copies its XCD's 64 state rows (4 sample tiles x 29 KB, tile order, contiguous) into LDS, six waves multiply,
//     accumulators meet in LDS, all eight waves do the gate arithmetic of one (unit tile, sample tile) each and store.
// Placement is never assumed: workgroups join the team of the XCC they find themselves on (k_xcd.h); teams that do not come out at exactly 32
// raise the error word, every wait is bounded.  A failed sweep's outputs are garbage; nothing falls back by itself: the failure is recorded in
// the process's STICKY fault words, the optimiser kernels skip every step from then on (k_optim.h), and the trainer repeats those steps on the
// step kernels after its next look (news_recommendation_amd/train_fast.py; include/nr_engine.h "Fault words").
#pragma once
#include "nr_common.h"
#include "k_gru.h"
#include "k_xcd.h"

namespace nr {

struct GruSeqFwdParams {
  const float* gi;        // [B*N][3*Hg] f32, row b*N + t
  const u16* Whh;         // bf16 [3*Hg][Hp], tile order
  const float* b_ih;      // [3*Hd]
  const float* b_hh;      // [3*Hd]
  const int* len;         // [B], >= 1
  u16* h_t2;              // 2 x bf16 [ceil16(B)][Hp], tile order: step t reads buffer t & 1, writes (t + 1) & 1
  u16* H_all;             // bf16 [T + 1][B][Hp] row-major (step t writes block t + 1), or null
  float* h_f2;            // 2 x f32 [B][Hp]: step t reads buffer t & 1, writes (t + 1) & 1
  u16* gates;             // bf16 [T][B][4][Hg], or null
  int B, N, Hd, Hg, Hp, T;
  XcdSync sync;
  long long* stamps;      // debug (tools/gru_persist_ab.py --timeline): [workgroup][step][wave][8] stamps of the 100 MHz constant clock (s_memrealtime: comparable across CUs), or null
};

template <int KS>
struct GruPersistGeom {
  static constexpr int NWAVE = 8, NT = NWAVE * 64;
  static constexpr int STX = 4;                          // sample tiles (of 16) per XCD: B <= 512
  static constexpr int H_BYTES = STX * KS * 1024;        // the XCD's state rows in tile order: 118,784 B at Hd = 900
  static constexpr int EX_BYTES = 2 * 3 * STX * 1024;    // accumulator images [unit tile][gate][sample tile][lane] f32x4
  static constexpr int SMEM = H_BYTES + EX_BYTES + 16;
  static_assert(SMEM <= 163840, "LDS");
};

// L1-bypassing global -> LDS copy (nt): the source was written by OTHER CUs of this XCD during the previous step and must come from the L2
#ifdef NR_EMU
#define NR_GLDS16_S_NT(base, voff, lds_dst) NR_GLDS16_S(base, voff, lds_dst)
#else
__device__ __forceinline__ void glds16_sbase_nt(const void* base, uint32_t voff, void* lds_dst) {
  uint32_t keep;
  const uint64_t b64 = (uint64_t)(uintptr_t)base;           // (pinned to scalar registers: the allocator does not always see that the base is wave-uniform)
  const uint64_t sb = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64) |
                      ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b64 >> 32)) << 32);
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sb), "s"(__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds_dst)) : "memory");
}
#define NR_GLDS16_S_NT(base, voff, lds_dst) nr::glds16_sbase_nt((base), (voff), (lds_dst))
#endif

template <int KS>
__global__ __launch_bounds__(GruPersistGeom<KS>::NT, 2) void gru_fwd_persist_kernel(GruSeqFwdParams p) {
  using Gm = GruPersistGeom<KS>;
  NR_SMEM_DECL(smem);
  unsigned char* const hbuf = smem;
  unsigned char* const exch = smem + Gm::H_BYTES;
  int* const bcast = (int*)(smem + Gm::H_BYTES + Gm::EX_BYTES);
  const int l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15;
  const int xcd = uniform(xcc_id());
  const int slot = uniform(xcd_join(p.sync, xcd, bcast));
  if (slot < 0) return;
  const int n_tiles = p.Hg / 16, nst = (p.B + 15) / 16;
  const int tpx = (nst + NR_XCDS - 1) / NR_XCDS;                  // sample tiles per XCD (<= STX: checked by the launcher)
  const int st0 = xcd * tpx;                                      // this XCD's first sample tile
  const int nstx = st0 < nst ? (nst - st0 < tpx ? nst - st0 : tpx) : 0;      // ... and how many it has
  const int tile2[2] = {slot, slot + NR_XCD_TEAM};                // the unit tiles of this CU slot
  // ---- MFMA role (waves 0 .. 5): gate q of unit tile tile2[tw]; its W_hh rows stay in registers for the whole sweep -----------------------
  const int tw = w / 3, q = w - tw * 3;
  const bool mfma_role = w < 6 && tile2[tw < 2 ? tw : 0] < n_tiles && nstx > 0;
  u16x8 wa[KS];
  if (mfma_role) {
    const u16* wp = p.Whh + ((size_t)q * p.Hg + tile2[tw] * 16) * p.Hp + l * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wa[ks] = *(const u16x8*)(wp + ks * 512);
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wa[ks] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  // ---- gate role (all waves): unit tile tile2[gt], sample tile gs of the XCD: lane = sample li, units jb .. jb + 3 ------------------------------
  const int gt = w >> 2, gs = w & 3;
  const bool gate_role = tile2[gt] < n_tiles && gs < nstx;
  const int jb = tile2[gt] * 16 + 4 * g;
  const int s = (st0 + gs) * 16 + li;                             // the lane's sample
  const int sb = s < p.B ? s : p.B - 1;
  const bool live = gate_role && s < p.B && jb < p.Hg;          // (units Hd .. Hg - 1: zero weight rows; their gate records are written, their state is not)
  f32x4 b_ir, b_hr, b_iz, b_hz, b_in, b_hn;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = jb + r < p.Hd ? jb + r : p.Hd - 1;
    b_ir[r] = p.b_ih[j]; b_hr[r] = p.b_hh[j];
    b_iz[r] = p.b_ih[p.Hd + j]; b_hz[r] = p.b_hh[p.Hd + j];
    b_in[r] = p.b_ih[2 * p.Hd + j]; b_hn[r] = p.b_hh[2 * p.Hd + j];
  }
  const int len_s = p.len[sb];
  const size_t ht = (size_t)((p.B + 15) / 16) * 16 * p.Hp, hf = (size_t)p.B * p.Hp;
  const int jc = jb < p.Hg ? jb : 0;
  unsigned bar = 0;
  auto stamp = [&](int t, int k) {
    if (p.stamps != nullptr && l == 0) p.stamps[(((size_t)blockIdx.x * p.T + t) * Gm::NWAVE + w) * 8 + k] = (long long)__builtin_amdgcn_s_memrealtime();
  };
  // epilogue operands of a step (the lane's gi rows and its own fp32 state): requested BEFORE the inter-workgroup wait of the previous step
  // -- neither depends on another workgroup -- so that their HBM latency runs under it
  f32x4 gir = f32x4{0.f, 0.f, 0.f, 0.f}, giz = gir, gin = gir, ho = gir;
  auto fetch_epi = [&](int t) {
    if (!gate_role) return;
    const float* gi = p.gi + ((size_t)sb * p.N + t) * 3 * p.Hg + jc;
    gir = *(const f32x4*)gi; giz = *(const f32x4*)(gi + p.Hg); gin = *(const f32x4*)(gi + 2 * p.Hg);
    ho = ld_nt((const f32x4*)(p.h_f2 + (size_t)(t & 1) * hf + (size_t)sb * p.Hp + jc));     // (written by this very lane a step ago; nt: not through a stale L1 line)
  };
  fetch_epi(0);
  for (int t = 0; t < p.T; ++t) {
    stamp(t, 0);
    // this XCD's state rows of step t: nstx * KS contiguous 1 KB blocks (tile order), L1-bypassing copies spread over the eight waves.  The 32
    // workgroups of the XCD read the SAME 116 KB at the same moment: each starts at a different block (slot * 7 mod the count) so that they
    // do not queue on the same L2 channels in lock step.
    {
      const unsigned char* src = (const unsigned char*)(p.h_t2 + (size_t)(t & 1) * ht + (size_t)st0 * p.Hp * 16);
      const int nblk = nstx * KS;
      int blk = (w + slot * 7) % (nblk > 0 ? nblk : 1);
      for (int i = w; i < nblk; i += Gm::NWAVE) {
        NR_GLDS16_S_NT(src, (unsigned)(blk * 1024 + l * 16), hbuf + blk * 1024);
        blk += Gm::NWAVE;
        if (blk >= nblk) blk -= nblk;
      }
    }
    stamp(t, 1);
    NR_WAIT_VMCNT(0);
    stamp(t, 2);
    __syncthreads();
    stamp(t, 3);
    if (mfma_role) {
      // one accumulation chain per sample tile, k ascending: the summation order of the step kernel (bit-identical states).  All STX tiles
      // unconditionally (an XCD with fewer sample tiles multiplies whatever its LDS holds and nobody reads the result): no branch in the chain
      f32x4 acc[Gm::STX];
#pragma unroll
      for (int st = 0; st < Gm::STX; ++st) acc[st] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* hb = hbuf + l * 16;
      constexpr int DEP = 3;                                    // k-steps of state fragments in flight (LDS latency ~ 2 k-steps of MFMAs), pinned
      u16x8 fh[DEP][Gm::STX];
#pragma unroll
      for (int i = 0; i < DEP; ++i)
#pragma unroll
        for (int st = 0; st < Gm::STX; ++st) fh[i][st] = *(const u16x8*)(hb + (st * KS + i) * 1024);
      NR_SCHED_BARRIER();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int sl = ks % DEP;
#pragma unroll
        for (int st = 0; st < Gm::STX; ++st) acc[st] = mfma_16x16x32_bf16(wa[ks], fh[sl][st], acc[st]);
        if (ks + DEP < KS) {
#pragma unroll
          for (int st = 0; st < Gm::STX; ++st) fh[sl][st] = *(const u16x8*)(hb + (st * KS + ks + DEP) * 1024);
        }
        NR_SCHED_BARRIER();
      }
#pragma unroll
      for (int st = 0; st < Gm::STX; ++st)
        *(f32x4*)(exch + (((tw * 3 + q) * Gm::STX + st) * 64 + l) * 16) = acc[st];
    }
    stamp(t, 4);
    __syncthreads();
    stamp(t, 5);
    if (live) {
      const f32x4 ar = *(const f32x4*)(exch + (((gt * 3 + 0) * Gm::STX + gs) * 64 + l) * 16);
      const f32x4 az = *(const f32x4*)(exch + (((gt * 3 + 1) * Gm::STX + gs) * 64 + l) * 16);
      const f32x4 an = *(const f32x4*)(exch + (((gt * 3 + 2) * Gm::STX + gs) * 64 + l) * 16);
      const bool active = t < len_s;
      f32x4 hn, rr, zz, nn, qq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = jb + r < p.Hd;                    // units >= Hd: zero weight rows and gi columns, their h stays 0
        rr[r] = fast_sigmoid(gir[r] + b_ir[r] + ar[r] + b_hr[r]);
        zz[r] = fast_sigmoid(giz[r] + b_iz[r] + az[r] + b_hz[r]);
        qq[r] = an[r] + b_hn[r];
        nn[r] = fast_tanh(gin[r] + b_in[r] + rr[r] * qq[r]);
        const float v = active ? (1.0f - zz[r]) * nn[r] + zz[r] * ho[r] : ho[r];
        hn[r] = ok ? v : 0.0f;
      }
      if (jb < p.Hd) {
        *(f32x4*)(p.h_f2 + (size_t)((t + 1) & 1) * hf + (size_t)s * p.Hp + jb) = hn;
        u16x4 hb4 = pack4(hn);
        if (jb <= p.Hd && p.Hd < jb + 4) hb4[p.Hd - jb] = 0x3F80;        // column Hd = 1.0
        u16* ht_out = p.h_t2 + (size_t)((t + 1) & 1) * ht;
        *(u16x4*)(ht_out + tile_off(s, jb, p.Hp)) = hb4;
        u16* hall = p.H_all != nullptr ? p.H_all + (size_t)(t + 1) * hf + (size_t)s * p.Hp : nullptr;
        if (hall != nullptr) *(u16x4*)(hall + jb) = hb4;
        if (jb + 4 == p.Hd) {                                          // Hd % 4 == 0: the lane owning the last units also sets column Hd
          ht_out[tile_off(s, p.Hd, p.Hp)] = 0x3F80;
          if (hall != nullptr) hall[p.Hd] = 0x3F80;
        }
      }
      if (p.gates != nullptr) {
        u16* gp = p.gates + ((size_t)t * p.B + s) * 4 * p.Hg + jb;
        *(u16x4*)gp = pack4(rr);
        *(u16x4*)(gp + p.Hg) = pack4(zz);
        *(u16x4*)(gp + 2 * p.Hg) = pack4(nn);
        *(u16x4*)(gp + 3 * p.Hg) = pack4(qq);
      }
    }
    stamp(t, 6);
    if (t + 1 < p.T) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores of the step are in the XCD's L2
      stamp(t, 7);
      // waves 1 .. 7 request the next step's epilogue operands BEFORE the wait (the gi rows come from HBM: 535 -> 502 us per 50-step sweep);
      // wave 0 polls, and its poll loads would return behind them
      if (w != 0) fetch_epi(t + 1);
      xcd_barrier(p.sync, xcd, ++bar);
      if (w == 0) fetch_epi(t + 1);
    }
  }
}

// ---- backward sweep ---------------------------------------------------------------------------------------------------------------------------
// dh_t = carry_{t+1} + dGh_{t+1} W_hh for the XCD's 64 samples and the CU slot's 32 hidden units (tiles slot and slot + 32), then the gate
// derivatives of step t (gru_bwd_finish's arithmetic).  W_hh^T of the 32 units (32 x Kp: 176 KB) lives in the registers of all eight waves
// as 32x32x16 MFMA A fragments, each wave one EIGHTH of the contraction range (11 blocks of 32 at Hd = 900: 88 registers); the B fragments
// (dGh_{t+1} of the XCD's samples, tile order) are used once per workgroup and so come STRAIGHT from the L2 into registers (L1-bypassing
// 16-byte loads, a pinned ring) -- every byte of the XCD's 344 KB crosses into a CU exactly once per step, LDS is only the meeting point of
// the eight partial sums (64 KB).  Summation order differs from the step kernels (two chains over all of K there): equal to rounding, not
// bit for bit.
struct GruSeqBwdParams {
  const float* g_last;    // [B][Hd]
  const u16* WhhT;        // bf16 [Hp][Kp], tile order
  const u16* gates;       // bf16 [T][B][4][Hg]
  const u16* H_all;       // bf16 [T + 1][B][Hp]
  const int* len;
  u16* dgi;               // bf16 [B*N][Kp]
  u16* dgh;               // bf16 [T][B][Kp]
  u16* dgh_t2;            // 2 x bf16 [ceil16(B)][Kp], tile order: call i writes buffer i & 1, reads (i + 1) & 1
  float* carry2;          // 2 x f32 [B][Hp], same alternation; the last call (t = -1) leaves dh_0 in buffer T & 1
  int B, N, Hd, Hg, Hp, Kp, T;
  XcdSync sync;
  long long* stamps;      // debug: [workgroup][call][wave][8] constant-clock stamps, or null
};

template <int KB>        // Kp / 32
struct GruBwdPersistGeom {
  static constexpr int NWAVE = 8, NT = NWAVE * 64;
  static constexpr int NBW = (KB + NWAVE - 1) / NWAVE;       // contraction blocks per wave (waves beyond KB % 8 multiply a zero block at the end)
  static constexpr int EX_BYTES = NWAVE * 2 * 4 * 64 * 16;   // partial sums [wave][sample pair][unit group][lane] f32x4: 64 KB
  static constexpr int SMEM = EX_BYTES + 16;
  static constexpr int RING = 12;                            // B fragments in flight per wave (16 bytes per lane each)
};

template <int KB>
__global__ __launch_bounds__(GruBwdPersistGeom<KB>::NT, 2) void gru_bwd_persist_kernel(GruSeqBwdParams p) {
  using Gm = GruBwdPersistGeom<KB>;
  NR_SMEM_DECL(smem);
  unsigned char* const exch = smem;
  int* const bcast = (int*)(smem + Gm::EX_BYTES);
  const int l = lane_id(), w = wave_id();
  const int xcd = uniform(xcc_id());
  const int slot = uniform(xcd_join(p.sync, xcd, bcast));
  if (slot < 0) return;
  const int n_tiles = p.Hg / 16, nst = (p.B + 15) / 16;
  const int tpx = (nst + NR_XCDS - 1) / NR_XCDS;
  const int st0 = xcd * tpx;
  const int nstx = st0 < nst ? (nst - st0 < tpx ? nst - st0 : tpx) : 0;
  const int tile2[2] = {slot, slot + NR_XCD_TEAM};
  // ---- this wave's share of the contraction: blocks kb0 .. kb0 + nb - 1 (nb = NBW or NBW - 1) ------------------------------------------------------
  constexpr int BASE = KB / Gm::NWAVE, REM = KB % Gm::NWAVE;
  const int nb = BASE + (w < REM ? 1 : 0);
  const int kb0 = w * BASE + (w < REM ? w : REM);
  const int m = l & 31, kg = l >> 5;                       // MFMA row (A: unit, B: sample) and k group of the lane
  // piece of a 1 KB tile-order block (16 rows x 32 k) the lane takes for MFMA half h: k chunk 2 h + kg, row m & 15
  const int piece0 = (kg * 16 + (m & 15)) * 8, piece1 = ((2 + kg) * 16 + (m & 15)) * 8;
  u16x8 wa[Gm::NBW][2];
  {
    const int tl = tile2[m >> 4];
    const bool have = tl < n_tiles && nstx > 0;
    const u16* wp = p.WhhT + (size_t)(have ? tl : 0) * KB * 512;
#pragma unroll
    for (int j = 0; j < Gm::NBW; ++j) {
      const bool on = have && j < nb;
      const u16* bp = wp + (size_t)(kb0 + (j < nb ? j : 0)) * 512;
      const u16x8 z = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      wa[j][0] = on ? *(const u16x8*)(bp + piece0) : z;
      wa[j][1] = on ? *(const u16x8*)(bp + piece1) : z;
    }
  }
  // B side: sample pair sp = sample tiles st0 + 2 sp, + 1 (clamped into the XCD's tiles: products of a clamped tile are never read)
  int boff[2];
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) {
    int st = 2 * sp + (m >> 4);
    st = st < nstx ? st : (nstx > 0 ? nstx - 1 : 0);
    boff[sp] = ((nstx > 0 ? st0 + st : 0) * KB + kb0) * 512;              // elements; < 2^31 for B <= 512
  }
  // ---- epilogue role: wave e = (sample pair e >> 2, unit group G = e & 3): lane = sample m of the pair, units 8 G + 4 kg .. + 3 of the slot's 32 ------
  const int esp = w >> 2, eG = w & 3;
  const int m0 = 8 * eG + 4 * kg;                           // row of the 32-unit block
  const int etile = tile2[m0 >> 4];
  const int jb = etile * 16 + (m0 & 15);
  const int est = 2 * esp + (m >> 4);                       // sample tile of the XCD
  const int s = (st0 + est) * 16 + (m & 15);
  const bool live = etile < n_tiles && est < nstx && s < p.B;
  const int sb = live ? s : 0, jc = live ? jb : 0;
  const int len_s = p.len[sb];
  const size_t dt = (size_t)((p.B + 15) / 16) * 16 * p.Kp, cf = (size_t)p.B * p.Hp, gb = (size_t)p.B * 4 * p.Hg, db = (size_t)p.B * p.Kp;
  f32x4 cn = f32x4{0.f, 0.f, 0.f, 0.f};
  u16x4 rb = u16x4{0, 0, 0, 0}, zb = rb, nb_ = rb, qb = rb, hb = rb;
  auto fetch_epi = [&](int i) {                              // operands of call i (t = T - 1 - i): none depends on another workgroup
    const int t = p.T - 1 - i;
    if (!live) return;
    if (i > 0) cn = ld_nt((const f32x4*)(p.carry2 + (size_t)((i + 1) & 1) * cf + (size_t)sb * p.Hp + jc));      // written by this lane one call ago
    if (t >= 0) {
      const u16* gp = p.gates + (size_t)t * gb + (size_t)sb * 4 * p.Hg + jc;
      rb = *(const u16x4*)gp; zb = *(const u16x4*)(gp + p.Hg); nb_ = *(const u16x4*)(gp + 2 * p.Hg); qb = *(const u16x4*)(gp + 3 * p.Hg);
      hb = *(const u16x4*)(p.H_all + (size_t)t * cf + (size_t)sb * p.Hp + jc);
    }
  };
  fetch_epi(0);
  unsigned bar = 0;
  auto stamp = [&](int i, int k) {
    if (p.stamps != nullptr && l == 0) p.stamps[(((size_t)blockIdx.x * (p.T + 1) + i) * Gm::NWAVE + w) * 8 + k] = (long long)__builtin_amdgcn_s_memrealtime();
  };
  for (int i = 0; i <= p.T; ++i) {
    const int t = p.T - 1 - i;
    stamp(i, 0);
    if (i > 0 && nstx > 0) {               // (workgroup-uniform: an XCD without samples only keeps the team's step count)
      // ---- dGh_{t+1} W_hh: NBW blocks x 2 halves x 2 sample pairs per wave, operands through a pinned ring of L1-bypassing loads ---------------
      const u16* bsrc = p.dgh_t2 + (size_t)((i + 1) & 1) * dt;
      constexpr int NL = Gm::NBW * 4;                        // loads (= MFMAs) of the wave: index q = (j * 2 + h) * 2 + sp
      auto bptr = [&](int q) {
        const int sp = q & 1, h = (q >> 1) & 1, j = q >> 2;
        return (const u16x8*)(bsrc + boff[sp] + (j < nb ? j : 0) * 512 + (h ? piece1 : piece0));
      };
      u16x8 fr[Gm::RING];
#pragma unroll
      for (int q = 0; q < Gm::RING; ++q)
        if (q < NL) fr[q] = ld_nt(bptr(q));
      f32x16 acc[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[0][e] = 0.0f; acc[1][e] = 0.0f; }
      NR_SCHED_BARRIER();
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        const int sp = q & 1, h = (q >> 1) & 1, j = q >> 2;
        acc[sp] = mfma_32x32x16_bf16(wa[j][h], fr[q % Gm::RING], acc[sp]);
        if (q + Gm::RING < NL) fr[q % Gm::RING] = ld_nt(bptr(q + Gm::RING));
        NR_SCHED_BARRIER();
      }
      stamp(i, 1);
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int G = 0; G < 4; ++G)
          *(f32x4*)(exch + ((((w * 2 + sp) * 4 + G) * 64 + l) * 16)) = f32x4{acc[sp][4 * G], acc[sp][4 * G + 1], acc[sp][4 * G + 2], acc[sp][4 * G + 3]};
      stamp(i, 2);
      __syncthreads();
    }
    stamp(i, 3);
    if (live) {
      f32x4 dh;
      if (i == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[r] = jb + r < p.Hd ? p.g_last[(size_t)s * p.Hd + jb + r] : 0.0f;
      } else {
        dh = cn;
#pragma unroll
        for (int ww = 0; ww < Gm::NWAVE; ++ww) dh = dh + *(const f32x4*)(exch + ((((ww * 2 + esp) * 4 + eG) * 64 + l) * 16));
      }
      float* cy_out = p.carry2 + (size_t)(i & 1) * cf + (size_t)s * p.Hp + jb;
      if (t < 0) {
        *(f32x4*)cy_out = dh;                                 // dh_0
      } else {
        const bool active = t < len_s;
        f32x4 d_r = f32x4{0.f, 0.f, 0.f, 0.f}, d_z = d_r, d_n = d_r, d_nr = d_r, cy = dh;
        if (active) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = jb + r < p.Hd;
            const float rr = bf2f(rb[r]), zz = bf2f(zb[r]), nn = bf2f(nb_[r]), qq = bf2f(qb[r]), hp = bf2f(hb[r]);
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
        *(f32x4*)cy_out = cy;
        u16* gi = p.dgi + ((size_t)s * p.N + t) * p.Kp + jb;
        const u16x4 pr = pack4(d_r), pz = pack4(d_z), pn = pack4(d_n), pnr = pack4(d_nr);
        *(u16x4*)gi = pr;
        *(u16x4*)(gi + p.Hg) = pz;
        *(u16x4*)(gi + 2 * p.Hg) = pn;
        u16* gh = p.dgh + (size_t)t * db + (size_t)s * p.Kp + jb;
        *(u16x4*)gh = pr;
        *(u16x4*)(gh + p.Hg) = pz;
        *(u16x4*)(gh + 2 * p.Hg) = pnr;
        u16* ght = p.dgh_t2 + (size_t)(i & 1) * dt;
        *(u16x4*)(ght + tile_off(s, jb, p.Kp)) = pr;
        *(u16x4*)(ght + tile_off(s, p.Hg + jb, p.Kp)) = pz;
        *(u16x4*)(ght + tile_off(s, 2 * p.Hg + jb, p.Kp)) = pnr;
      }
    }
    stamp(i, 4);
    if (i < p.T) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(i, 5);
      xcd_barrier(p.sync, xcd, ++bar);
      // (requested AFTER the wait, unlike the forward sweep: six scattered loads per lane ahead of the arrival counter's traffic cost 3 us of
      //  wait per call here -- 803 vs 637 us per 51-call sweep; they have the whole product phase to arrive)
      fetch_epi(i + 1);
    }
  }
}

}  // namespace nr
