// Shared compile-time geometry and small device helpers of the NRMS engine kernels.
#pragma once
#include <nr_prims.h>   // resolved via -I: csrc/ for the product, tests/emu/ for the CPU emulation build
#include "../../include/nr_engine.h"

namespace nr {

constexpr int D = NR_D;            // 300 embedding / model dim
constexpr int KP = NR_KP;          // 320: K padded to 10 MFMA k-steps of 32
constexpr int KSTEPS = KP / 32;    // 10
constexpr int XS = KP + 8;         // LDS row stride (elements) of a token tile: 656 B -> conflict-free b128 fragment reads
constexpr int H = NR_HEADS;        // 15
constexpr int DK = NR_DK;          // 20
constexpr int HG = 4;              // heads per head-group (4*20 = 80 = 5 n-tiles of 16)
constexpr int NGROUPS = (H + HG - 1) / HG;   // 4 groups: 4,4,4,3 heads
constexpr int NP = NR_NP;          // 304 rows per packed W block
constexpr int QP = NR_QP;          // 208
constexpr int QS = HG * DK + 8;    // 88: LDS row stride (elements) of the per-group Q / K tiles (176 B)
constexpr int WG = 256;            // threads per workgroup of the 4-wave kernels
constexpr float EXP_CLAMP = 80.0f; // exp() argument clamp: keeps sum_j exp(s_j) finite in fp32 for S <= 64 (the reference overflows to inf/nan there)
constexpr int D4 = D / 4;          // 75 float4 per embedding row

static_assert(D % 4 == 0 && DK % 4 == 0 && (HG * DK) % 16 == 0, "geometry");

// ---- counter-based RNG for dropout: Philox4x32-7, one call per 4 consecutive elements -------------
__device__ __forceinline__ void philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                             uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = mulhi_u32(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = mulhi_u32(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct DropCfg {
  uint32_t k0, k1;     // seed
  uint32_t thresh;     // drop iff rand32 < thresh  (thresh = p * 2^32)
  float scale;         // 1/(1-p)
  int enabled;
};

// keep-flags (bit j set = keep) for elements 4*quad .. 4*quad+3 of dropout site `site`
__device__ __forceinline__ uint32_t drop_keep4(const DropCfg& dc, uint32_t site, uint64_t quad) {
  uint32_t r[4];
  philox4x32_7((uint32_t)quad, (uint32_t)(quad >> 32), site, 0x6e72u, dc.k0, dc.k1, r);
  uint32_t m = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) m |= (r[j] >= dc.thresh ? 1u : 0u) << j;
  return m;
}

// ---- balanced split of (column-group, token-tile) GEMM units over the nw waves of a workgroup -----------------------
// Column groups are gs (1 or 2) consecutive 16-row n-tiles (a trailing single when the tile count is odd).  Tile-units are
// ordered (group, token-tile, tile-in-group); wave w owns tile-unit range [w*TU/nw, (w+1)*TU/nw) and a
// (group, token-tile) unit belongs to the wave that owns its first tile-unit.
__device__ __forceinline__ void unit_range(int ntiles, int MT, int w, int nw, int cg, int& G, int& m_begin, int& m_end,
                                           int gs = 2) {
  int TU = ntiles * MT;
  int lo = (w * TU) / nw, hi = ((w + 1) * TU) / nw;
  G = ntiles - cg * gs;
  G = G > gs ? gs : G;
  int base = cg * gs * MT;
  int a = lo - base, b = hi - base;
  m_begin = a <= 0 ? 0 : (a + G - 1) / G;
  m_end = b <= 0 ? 0 : (b + G - 1) / G;
  if (m_begin > MT) m_begin = MT;
  if (m_end > MT) m_end = MT;
}

__device__ __forceinline__ u16x4 pack4(f32x4 v) {
  u16x4 o;
  o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
  return o;
}

__device__ __forceinline__ u16x8 cat8(u16x4 lo, u16x4 hi) {
  u16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}

}  // namespace nr
