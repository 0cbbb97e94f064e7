// Backward of the title encoder's attention core for gfx950, DMA form (round 5): autograd of ScaledDotProductAttention
// (src/model/general/attention/multihead_self.py:15-23) for 20-token titles from the head-major saves of qkv_proj_kernel (k_proj.h).
//
// What the round-4 kernel (attn_bwd_kernel<20, 4, ., TILE>, k_bwd.h) spent its time on (profiles/r05_attn_bwd_timeline.txt): of ~6,100 cycles
// per (title, head) pair and wave, 1,400 went into moving the pair's operands registers -> wave-private LDS (8-byte pieces, the dropout hash and
// the direct term of dC per pair), 1,040 into ISSUING the next pair's 13 global loads (8-byte pieces, two thirds of the lanes), 1,060 into
// fragment reads + 12 identity MFMAs that transpose K, Q and dC; with the arithmetic switched off the kernel still took 486 of its 707 us.
// Here:
//   * a pair's Q | K | V block (2,400 contiguous bytes) goes global -> LDS directly (three global_load_lds_dwordx4 per pair, issued as soon as
//     the previous pair's fragments are in registers): no staging registers, no LDS stores, no per-lane address arithmetic;
//   * the title's dctx rows, its pooled-vector gradient and its pooling weights arrive the same way (one 14 KB copy per title, issued a title
//     ahead); ONE pass per title turns them into the head-major dC tile [15][20][20] (direct term + dropout, 16-byte pieces, all lanes busy);
//   * every MFMA operand is a plain or a TRANSPOSING LDS read (ds_read_b64_tr_b16 hands out K^T / Q^T / dC^T fragments): the 12 identity
//     MFMAs and their 24 conversions per pair are gone; lanes whose k-slots lie outside the head (d >= 20) read a zero block instead of
//     being masked in registers, row clamps are folded into per-lane offsets computed once;
//   * five waves per title = three rounds of five heads, no idle wave (four waves: 4 + 4 + 4 + 3); two barriers per title; the write-out of
//     title t and the dC pass of title t + 1 share one barrier interval, the stores drain behind the next title's arithmetic.
// Per pair: 28 MFMAs (was 40), 28 LDS reads + 12 LDS stores, 3 copy instructions (was 13 loads).
#pragma once
#include "nr_common.h"
#include "k_proj.h"

namespace nr {

template <int NW_>
struct AttnBwd2Geom {
  static constexpr int S = 20;
  static constexpr int NW = NW_;                        // waves per workgroup = heads per round (8: two rounds, 8 + 7 heads, four waves per SIMD; 5: three full rounds)
  static constexpr int NT = NW * 64;
  static constexpr int ROUNDS = (H + NW - 1) / NW;
  static_assert(S == 20 && DK == 20, "fragment offsets are written for 20 x 20 blocks");
  static constexpr int BLK = HM_BLK * 2;                // 800 B: one [20][20] bf16 block, row stride 40 B
  static constexpr int OPER = 3 * BLK;                  // 2,400 B: Q | K | V of a pair
  static constexpr int OPER_STRIDE = OPER + 32;         // transposing reads of rows 16 .. 19 with d-tile 1 run 22 B past a block
  static constexpr int GO_BYTES = D * 4;                // 1,200
  static constexpr int WT_BYTES = S * 4;                // 80
  static constexpr int GW_BYTES = GO_BYTES + WT_BYTES;  // 1,280 = 80 sixteen-byte pieces: the title's pooled-vector gradient and pooling weights
  static constexpr int GW_PIECES = GW_BYTES / 16;
  static constexpr int TOUCH_BYTES = 512;               // landing area of the 4-byte copies that pull the next title's dctx lines into the cache (never read)
  static constexpr int DCTX_LINES = S * KP * 2 / 128;   // 100 cache lines
  static constexpr int DCH_BYTES = (H + 1) * BLK;       // head-major dC tile + a dummy block (the dC pass writes its padding quad there; zeroed once: slack of the transposing reads)
  static constexpr int LDG = 3 * KP;                    // 960 columns of a dqkv row (dQ | dK | dV blocks of KP)
  static constexpr int TROW = LDG * 2 + 16;             // 1,936 B per staged dqkv row
  static constexpr int TILE_BYTES = S * TROW;           // 38,720
  static constexpr int ZERO_BYTES = 2 * BLK + 16;       // lanes without a k-slot read zeros at block offsets 0, 800 and 1,600
  static constexpr int SMEM = NW * OPER_STRIDE + GW_BYTES + DCH_BYTES + TILE_BYTES + ZERO_BYTES + TOUCH_BYTES;    // 74,384 B at eight waves: two workgroups per CU
  static_assert(2 * SMEM <= 163840 && OPER_STRIDE % 16 == 0 && GW_BYTES % 16 == 0 && DCH_BYTES % 16 == 0 && TILE_BYTES % 16 == 0, "LDS budget / alignment");
  static constexpr int CPR = (D + 7) / 8;               // 38 sixteen-byte pieces cover the 300 real columns of a dctx row
  static constexpr int ASM_IT = (S * CPR + NT - 1) / NT;          // dctx pieces per thread in the dC pass (2 at eight waves)
  static constexpr int WO_PIECES = S * (LDG / 8);       // 2,400 sixteen-byte pieces of the title's dqkv rows (contiguous in global memory)
  static constexpr int WO_IT = (WO_PIECES + NT - 1) / NT;         // per thread (5 at eight waves)
};

struct AttnBwd2Params {
  const u16* qkv;          // [n_seq][H][3][20][20] head-major Q, K, V (16-byte aligned)
  const u16* dctx;         // [n_seq * 20][KP] bf16: dpre @ Wa, the GEMM part of the pooling backward (16-byte aligned, row stride KP)
  const float* attn_w;     // [n_seq][20] pooling weights of the forward
  const float* g_out;      // [n_seq][D] gradient of the pooled vector
  u16* dqkv;               // [n_seq * 20][LDG] bf16
  int64_t n_seq;
  const int32_t* key_len;  // optional [n_seq]
  DropCfg dc;              // dropout site 2
  int debug;               // DBG instantiation (NR_ATTNB_DEBUG): 1 no global -> LDS copies, 4 no dqkv stores, 16 no arithmetic (8: nothing off)
  unsigned long long* stamps;   // DBG instantiation: [2 workgroups][NW waves][4 titles][ROUNDS][12] cycle-counter stamps (tools/attnb_timeline.py)
};

template <int NW, bool DBG>
__global__ __launch_bounds__(AttnBwd2Geom<NW>::NT, NW == 8 ? 4 : (NW == 5 ? 3 : 2)) void attn_bwd2_kernel(AttnBwd2Params p) {
  using Gm = AttnBwd2Geom<NW>;
  constexpr int S = Gm::S;
  const int dbg = DBG ? p.debug : 0;
  p.dc = drop_resolve(p.dc);
  NR_SMEM_DECL(smem);
  const int l = lane_id(), w = wave_id(), g = l >> 4, li = l & 15, tid = (int)threadIdx.x;
  unsigned char* const oper = smem + w * Gm::OPER_STRIDE;             // wave-private: Q | K | V of the current pair
  unsigned char* const gw = smem + Gm::NW * Gm::OPER_STRIDE;          // g_out row | pooling weights of the NEXT title to be assembled
  unsigned char* const dch = gw + Gm::GW_BYTES;                       // dC of the current title, head-major
  unsigned char* const tile = dch + Gm::DCH_BYTES;                    // dqkv rows of the current title
  unsigned char* const zero = tile + Gm::TILE_BYTES;
  unsigned char* const touch = zero + Gm::ZERO_BYTES;

  // ---- one-time LDS state: zero block, slack behind the operand / dC blocks (finite), K padding of the staged rows ---------------------------
  for (int i = tid; i < Gm::ZERO_BYTES / 8; i += Gm::NT) *(u16x4*)(zero + i * 8) = u16x4{0, 0, 0, 0};
  if (l < 4) *(u16x4*)(oper + Gm::OPER + l * 8) = u16x4{0, 0, 0, 0};
  if (tid < Gm::BLK / 8) *(u16x4*)(dch + H * Gm::BLK + tid * 8) = u16x4{0, 0, 0, 0};
  {
    constexpr int PADQ = (KP - D) / 4;
    for (int i = tid; i < S * 3 * PADQ; i += Gm::NT) {
      const int r = i / (3 * PADQ), c = i - r * (3 * PADQ);
      *(u16x4*)(tile + r * Gm::TROW + ((c / PADQ) * KP + D + (c % PADQ) * 4) * 2) = u16x4{0, 0, 0, 0};
    }
  }

  // ---- copies ---------------------------------------------------------------------------------------------------------------------------------
  auto fetch_pair = [&](int64_t seq, int hd) {            // 150 sixteen-byte pieces: lanes 0 .. 63, 0 .. 63, 0 .. 21
    if (dbg & 1) return;
    // (wave-uniform 64-bit base + unsigned 32-bit lane offset: the copy takes the base in scalar registers; per-lane 64-bit addresses of the
    //  three copy streams were what the register allocator spilled under the 128-register cap)
    const unsigned char* base = (const unsigned char*)(p.qkv + (seq * H + hd) * HM_PAIR);
    const unsigned lo = (unsigned)l * 16u;
    NR_GLDS16_S(base, lo, oper);
    NR_GLDS16_S(base, lo + 1024u, oper + 1024);
    if (l < Gm::OPER / 16 - 128) NR_GLDS16_S(base, lo + 2048u, oper + 2048);
  };
  auto fetch_gw = [&](int64_t seq) {                      // 75 + 5 pieces, wave 0: lanes 0 .. 63, 0 .. 15
    if ((dbg & 1) || w != 0) return;
    const unsigned char* s_go = (const unsigned char*)(p.g_out + seq * D);
    const unsigned char* s_wt = (const unsigned char*)(p.attn_w + seq * S);
    constexpr int P1 = Gm::GO_BYTES / 16;                   // 75: lanes 0 .. 10 of the second copy finish the g_out row, lanes 11 .. 15 take the weights
    const unsigned lo = (unsigned)l * 16u;
    NR_GLDS16_S(s_go, lo, gw);
    if (l < P1 - 64) NR_GLDS16_S(s_go, lo + 1024u, gw + 1024);
    else if (l < Gm::GW_PIECES - 64) NR_GLDS16_S(s_wt, lo - (unsigned)(P1 - 64) * 16u, gw + 1024);
  };
  // The title's dctx rows for the dC pass: ASM_IT sixteen-byte pieces per thread, straight into registers at the top of the pass (through LDS
  // like the operands they cost 12.8 KB per workgroup -- the difference between five and eight waves per title; held in registers across
  // the pairs they do not fit under the 128-register cap of four waves per SIMD).  Their HBM latency is taken a title earlier: in round 0 one
  // wave TOUCHES the next title's 100 cache lines with 4-byte global -> LDS copies into a landing area nobody reads, so the loads of the pass
  // hit the L2 (the copies are asynchronous like every other: nothing waits for them but the title barrier).
  auto touch_dctx = [&](int64_t seq) {
    if ((dbg & 1) || w != 1 % Gm::NW) return;
    const unsigned char* base = (const unsigned char*)(p.dctx + seq * S * KP);
    const unsigned lo = (unsigned)l * 128u;
    NR_GLDS4_S(base, lo, touch);
    if (l < Gm::DCTX_LINES - 64) NR_GLDS4_S(base, lo + 64u * 128u, touch + 256);
  };
  u16x8 dgp[Gm::ASM_IT];
  auto prefetch_dctx = [&](int64_t seq) {
    const BufRsrc rd = make_buf(p.dctx + seq * S * KP, (dbg & 1) ? 0u : (uint32_t)(S * KP * 2));
    int tq = tid;
    NR_OPAQUE(tq);
#pragma unroll
    for (int i = 0; i < Gm::ASM_IT; ++i) {
      const int idx = tq + i * Gm::NT;
      const int r = (idx * 1725) >> 16, pc = idx - r * Gm::CPR;                     // idx / 38 for idx < 760 (larger: past the resource, zeros)
      dgp[i] = buf_load16<0>(rd, idx < S * Gm::CPR ? (uint32_t)((r * KP + pc * 8) * 2) : 0xFFFFFFF0u);
    }
  };

  // ---- per-lane fragment offsets, computed once -------------------------------------------------------------------------------------------------
  // Row fragment of a [20][20] block (A / B operand whose k-slots are the head dim, natural order d = 8 g + j) for token tile t: row 16 t + li
  // (clamped to 19: duplicates whose probabilities / gradients are exact zeros), halves d = 8 g .. 8 g + 3 and 8 g + 4 .. 8 g + 7; halves
  // with d >= 20 point at the zero block.  Offsets are relative to the block start; the zero offsets relative to `oper` (block offsets 0,
  // 800, 1,600 keep them inside the zero block) -- the dC tile gets its own pair below.
  const int zoff = (int)(zero - oper);
  int lo_off[2], hi_off[2], tr_off[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = 16 * t + li < S ? 16 * t + li : S - 1;
    lo_off[t] = g < 3 ? row * 40 + 16 * g : zoff;
    hi_off[t] = g < 2 ? row * 40 + 16 * g + 8 : zoff;
    // transposing read of k-rows 16 t + 4 g .. + 3 (clamped), columns 16 dt .. 16 dt + 15 (dt through the immediate): this lane supplies
    // the piece &M[16 t + 4 g + (li >> 2)][16 dt + 4 (li & 3)] and receives M[16 t + 4 g + 0 .. 3][16 dt + li]
    const int trow = 16 * t + 4 * g + (li >> 2) < S ? 16 * t + 4 * g + (li >> 2) : S - 1;
    tr_off[t] = trow * 40 + 8 * (li & 3);
  }
  // dC tile: block of head hd at dch + hd * 800; zero lanes mask the head offset away so that every head reads the zero block
  const int dc_lo_base[2] = {g < 3 ? (int)(dch - smem) + lo_off[0] : (int)(zero - smem), g < 3 ? (int)(dch - smem) + lo_off[1] : (int)(zero - smem)};
  const int dc_hi_base[2] = {g < 2 ? (int)(dch - smem) + hi_off[0] : (int)(zero - smem), g < 2 ? (int)(dch - smem) + hi_off[1] : (int)(zero - smem)};
  const int dc_lo_sel = g < 3 ? -1 : 0, dc_hi_sel = g < 2 ? -1 : 0;
  // free k-slot d = 20 (element 4 of the g == 2 lanes) carries the key mask: q = 1, k = 0 for a live key, -29952 for a padded one
  const uint32_t qone = g == 2 ? (uint32_t)BF16_ONE : 0u;

  const float inv_sqrt_dk = 1.0f / sqrtf((float)DK);
  const float c2 = LOG2E * inv_sqrt_dk, clamp2 = EXP_CLAMP * LOG2E;
  const u16x4 Z4 = u16x4{0, 0, 0, 0};
  u16x8 identc;                                           // identity on the CL k-slots (see k_bwd.h): transposes packed P^T / dS^T tiles on the matrix core
#pragma unroll
  for (int j = 0; j < 8; ++j) identc[j] = (j < 4 && 4 * g + j == li) ? BF16_ONE : (u16)0;

  auto ld4 = [&](const unsigned char* a) -> u16x4 { return *(const u16x4*)a; };

  int rnd = 0, it_t = 0;
  auto stamp = [&](int k) {
    if (DBG && p.stamps != nullptr && l == 0 && blockIdx.x < 2 && it_t < 4)
      p.stamps[((((size_t)blockIdx.x * Gm::NW + w) * 4 + it_t) * Gm::ROUNDS + rnd) * 12 + k] = __builtin_readcyclecounter();
  };

  // ---- the dC pass: dctx rows (registers) + attn_w (x) g_out, dropout 2, -> head-major bf16 tile --------------------------------------------------
  auto assemble = [&](int64_t seq) {
    const float* go = (const float*)gw;
    const float* wt = (const float*)(gw + Gm::GO_BYTES);
    const uint64_t qbase = (uint64_t)(seq * S) * D4;       // dropout quad index of the title's first element: scalar; the lane adds 32 bits
    // every wave waits for ALL of its dctx pieces here (a counted wait: the write-out's stores, issued behind them, stay in flight) -- also the
    // waves that skip the second piece below: a load still pending on some path makes the compiler drain the whole counter, stores included,
    // in front of the first instruction that reuses its destination registers (it did: in front of the first pair's fragment reads)
#pragma unroll
    for (int i = 0; i < Gm::ASM_IT; ++i) asm volatile("" : "+v"(dgp[i]));
#pragma unroll
    for (int i = 0; i < Gm::ASM_IT; ++i) {
      int idx = tid + i * Gm::NT;
      NR_OPAQUE(idx);                                      // (row / piece / addresses recomputed here: hoisted out of the title loop they cost ~20 registers)
      if (idx < S * Gm::CPR) {
        const int r = (idx * 1725) >> 16, pc = idx - r * Gm::CPR, c = pc * 8;      // idx / 38 for idx < 760
        const u16x8 dg = dgp[i];
        const float wr = wt[r];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int cc = c + 4 * hf;                       // a quad never straddles heads (20 % 4 == 0) and is one dropout quad
          // the quad of columns 300 .. 303 (last piece of a row) is computed like the others -- its g_out values are the pooling weights that
          // follow in the LDS block -- and lands in the dummy block behind the 15 heads: no branch
          const f32x4 g4 = *(const f32x4*)(go + cc);
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = bf2f(dg[4 * hf + j]) + wr * g4[j];
          if (p.dc.enabled) v = v * drop_mul4(p.dc, 2u, qbase + (uint32_t)(r * D4 + (cc >> 2)));
          const int hd = (cc * 3277) >> 16, d = cc - hd * DK;                     // cc / 20 for cc < 304
          *(u16x4*)(dch + hd * Gm::BLK + (r * DK + d) * 2) = pack4(v);
        }
      }
    }
  };
  // the title's rows are 38,400 contiguous bytes of dqkv: piece idx = tid + NT i goes to byte 16 idx, from tile row idx / 120
  auto writeout = [&](int64_t seq) {
    if (dbg & 4) return;
    const BufRsrc rs = make_buf(p.dqkv + seq * S * Gm::LDG, S * Gm::LDG * 2);
    int tq = tid;
    NR_OPAQUE(tq);
#pragma unroll
    for (int i = 0; i < Gm::WO_IT; ++i) {
      const int idx = tq + i * Gm::NT;
      const int r = (idx * 2185) >> 18;                     // idx / 120 for idx < 2,600
      buf_store16<0>(rs, idx < Gm::WO_PIECES ? (uint32_t)(tq * 16) : 0xFFFFFFF0u, *(const u16x8*)(tile + idx * 16 + r * 16), (uint32_t)(i * Gm::NT * 16));
    }
  };

  int64_t seq = blockIdx.x;
  if (seq >= p.n_seq) return;
  fetch_gw(seq);
  fetch_pair(seq, w);
  prefetch_dctx(seq);
  NR_WAIT_VMCNT(0);                                       // (the copies are asm: the barrier's own fence does not know them)
  __syncthreads();                                        // the first title's g_out row / weights have landed (every wave drains its own copies, then meets)
  assemble(seq);
  while (true) {
    rnd = 0;
    stamp(2);
    __syncthreads();                                      // dC tile complete; g_out row consumed; the previous title's rows have been read out of the tile
    const int klen = p.key_len != nullptr ? uniform(clamp_len(p.key_len[seq], S)) : S;
    const uint32_t kmask[2] = {(g == 2 && li >= klen) ? (uint32_t)BF16_NEG_BIG : 0u, (g == 2 && 16 + li >= klen) ? (uint32_t)BF16_NEG_BIG : 0u};
    const int64_t nseq = seq + gridDim.x;
#pragma unroll 1
    for (rnd = 0; rnd < Gm::ROUNDS; ++rnd) {
      const int hd = w + Gm::NW * rnd;
      if (hd >= H) break;                                 // (NW = 4: the last round has three heads; this wave's next copy is already in flight)
      stamp(3);
      // this pair's Q | K | V copy (issued a pair ago) has landed.  Round 0: the barrier in front of the dC pass drained it already -- what is
      // outstanding now are the previous title's dqkv stores, which nothing here has to wait for
      if (rnd > 0) NR_WAIT_VMCNT(0);
      wave_barrier();                                     // (no instruction: the wave waits as one; the CPU emulator runs lanes in turn)
      // ---- fragments ----
      u16x8 kf[2], qf[2], va[2], cp[2];
      u16x4 kcl[2][2], qcl[2][2], ccl[2][2];
      const int hoff = hd * Gm::BLK;
      const unsigned char* dcb = dch + hoff;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        qf[t] = cat8(ld4(oper + lo_off[t]), ld4(oper + hi_off[t]));
        kf[t] = cat8(ld4(oper + Gm::BLK + lo_off[t]), ld4(oper + Gm::BLK + hi_off[t]));
        va[t] = cat8(ld4(oper + 2 * Gm::BLK + lo_off[t]), ld4(oper + 2 * Gm::BLK + hi_off[t]));
        cp[t] = cat8(ld4(smem + dc_lo_base[t] + (hoff & dc_lo_sel)), ld4(smem + dc_hi_base[t] + (hoff & dc_hi_sel)));
        // (asm form of the transposing read: given the builtin, the compiler drains vmcnt in front of it -- i.e. waits for the previous title's
        //  dqkv stores in round 0; the explicit waits above / below order these reads against the copies and their consumers)
        const u16* trq = (const u16*)(oper + tr_off[t]);
        const u16* trc = (const u16*)(dcb + tr_off[t]);
        qcl[t][0] = lds_tr16_b64_async<0>(trq);
        qcl[t][1] = lds_tr16_b64_async<32>(trq);
        kcl[t][0] = lds_tr16_b64_async<Gm::BLK>(trq);
        kcl[t][1] = lds_tr16_b64_async<Gm::BLK + 32>(trq);
        ccl[t][0] = lds_tr16_b64_async<0>(trc);
        ccl[t][1] = lds_tr16_b64_async<32>(trc);
        // the key-mask slot (only the g == 2 lanes have non-zero constants; their upper halves are zeros from the zero block)
        qf[t] = or_dword2(qf[t], qone);
        kf[t] = or_dword2(kf[t], kmask[t]);
      }
      NR_WAIT_LGKMCNT(0);                                 // the fragments are in registers: the operand buffer may take the next pair
      wave_barrier();
      NR_SCHED_BARRIER();
      if (hd + Gm::NW < H) fetch_pair(seq, hd + Gm::NW);
      else if (nseq < p.n_seq) fetch_pair(nseq, w);
      if (rnd == 0 && nseq < p.n_seq) { fetch_gw(nseq); touch_dctx(nseq); }      // (this title's g_out row was consumed before the barrier above)
      stamp(4);

      if (!(dbg & 16)) {
      u16x4 dsT[2][2];    // CL(dS^T)  [key tile][query tile]
      u16x4 dsN[2][2];    // CL(dS)    [query tile][key tile]
      u16x4 pN[2][2];     // CL(P)     [query tile][key tile]
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x4 pT[2];
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
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
        f32x4 dPT[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          pT[kt] = pT[kt] * rden;
          dPT[kt] = mfma_16x16x32_bf16(va[kt], cp[qt], f32x4{0.f, 0.f, 0.f, 0.f});      // dP^T[key][q] = sum_dv V[key][dv] dC[q][dv]
#pragma unroll
          for (int r = 0; r < 4; ++r) dot += pT[kt][r] * dPT[kt][r];
        }
        dot = sum_rows4(dot);
        const bool qok = qt * 16 + li < S;                // padded queries are columns here: zeroed before they reach the contractions over q
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          f32x4 ds;
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[r] = pT[kt][r] * (dPT[kt][r] - dot) * inv_sqrt_dk;
          dsT[kt][qt] = qok ? pack4(ds) : Z4;
          pN[kt][qt] = qok ? pack4(pT[kt]) : Z4;          // still CL(P^T) [key tile][query tile]; transposed below
        }
        NR_SCHED_BARRIER();
      }
      stamp(5);
      {
        u16x4 tp[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            tp[qt][kt] = pack4(mfma_16x16x32_bf16(cat8(pN[kt][qt], Z4), identc, f32x4{0.f, 0.f, 0.f, 0.f}));
            dsN[qt][kt] = pack4(mfma_16x16x32_bf16(cat8(dsT[kt][qt], Z4), identc, f32x4{0.f, 0.f, 0.f, 0.f}));
          }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) pN[qt][kt] = tp[qt][kt];
        NR_SCHED_BARRIER();
      }
      stamp(6);
      // dQ^T[d][q]   = sum_key K^T[d][key] dS^T[key][q] : A = K^T fragments (k = key), B = CL(dS^T) (k = key)
      // dK^T[d][key] = sum_q   Q^T[d][q]   dS[q][key]   : A = Q^T,  B = CL(dS)
      // dV^T[dv][key]= sum_q   dC^T[dv][q] P[q][key]    : A = dC^T, B = CL(P)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int d0 = dt * 16 + 4 * g;
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
          const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
          const f32x4 aq = mfma_16x16x32_bf16(cat8(kcl[0][dt], kcl[1][dt]), cat8(dsT[0][ot], dsT[1][ot]), z);
          const f32x4 ak = mfma_16x16x32_bf16(cat8(qcl[0][dt], qcl[1][dt]), cat8(dsN[0][ot], dsN[1][ot]), z);
          const f32x4 av = mfma_16x16x32_bf16(cat8(ccl[0][dt], ccl[1][dt]), cat8(pN[0][ot], pN[1][ot]), z);
          const int tok = ot * 16 + li;
          if (d0 < DK && tok < S) {
            u16* dst = (u16*)(tile + tok * Gm::TROW) + hd * DK + d0;
            *(u16x4*)dst = pack4(aq);
            *(u16x4*)(dst + KP) = pack4(ak);
            *(u16x4*)(dst + 2 * KP) = pack4(av);
          }
        }
      }
      }                     // !(dbg & 16)
      stamp(7);
    }
    rnd = Gm::ROUNDS - 1;
    stamp(8);
    NR_WAIT_VMCNT(0);                                     // this wave's copies for the next title (first pair, g_out row, line touches) have landed
    __syncthreads();                                      // the title's rows are complete in the tile; the next title's g_out row has landed
    stamp(9);
    ++it_t;
    if (nseq >= p.n_seq) break;
    rnd = 0;
    stamp(0);
    // the next title's dctx pieces are requested BEFORE the write-out (they are L2 hits after the touches and fly behind its LDS reads), the
    // stores are issued behind them: the dC pass then waits for the loads only
    prefetch_dctx(nseq);
    writeout(seq);
    stamp(1);
    assemble(nseq);
    seq = nseq;
  }
  writeout(seq);
}

}  // namespace nr
