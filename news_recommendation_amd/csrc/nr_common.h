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
constexpr int QKP = (QP + 31) / 32 * 32;     // 224: query dim padded to the MFMA k-step (contraction dim of dctx = dpre @ Wa)
constexpr int QS = HG * DK + 8;    // 88: LDS row stride (elements) of the per-group Q / K tiles (176 B)
constexpr int WG = 256;            // threads per workgroup of the 4-wave kernels
constexpr float EXP_CLAMP = 80.0f; // exp() argument clamp: keeps sum_j exp(s_j) finite in fp32 for S <= 64 (the reference overflows to inf/nan there)
constexpr float LOG2E = 1.4426950408889634f;
constexpr u16 BF16_NEG_BIG = 0xC6EA;   // -29952.0: pad-key marker (exp2 of it, scaled, underflows to exactly 0)
constexpr u16 BF16_ONE = 0x3F80;
constexpr int D4 = D / 4;          // 75 float4 per embedding row

static_assert(D % 4 == 0 && DK % 4 == 0 && (HG * DK) % 16 == 0, "geometry");

// ---- counter-based RNG for dropout: two rounds of a 32-bit avalanche mixer per 4 consecutive elements -----------------
// The kernels are VALU-bound (rocprofv3 PMC: ~15 VALU instructions per MFMA in the forward kernel with Philox4x32-7, a third
// of them RNG), so the generator is as cheap as a stateless one gets: counter = element quad index, key = (seed, site);
// r0 = mix(counter ^ key), r1 = a one-multiply remix of r0 give 4 x 16 random bits, element j is dropped iff its 16 bits < p * 2^16.
// mix32 is the "lowbias32" integer finaliser (xorshift-multiply, full avalanche).  v_mul_lo_u32 is a QUARTER-rate instruction on
// CDNA (16 cycles per wave): the first version spent 6 of them per quad (two on spreading the counter, two full finalisers) and the
// ISA histogram of the training forward kernel showed ~1000 per head group, a third of its VALU time; this one spends 3.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

struct DropCfg {
  uint32_t k0, k1;     // seed
  uint32_t thresh;     // drop iff rand16 < thresh  (thresh = p * 2^16)
  float scale;         // 1/(1-p)
  int enabled;
  const uint32_t* ctr; // optional device-resident step counter (nr_set_step_counter): folded into the key by drop_resolve(), so that a
                       // training step captured ONCE into a HIP graph (seed baked into the kernel nodes) draws new masks at every replay
};

// The kernels call this once, on their by-value parameter copy, before any mask is drawn.
__device__ __forceinline__ DropCfg drop_resolve(DropCfg dc) {
  if (dc.ctr != nullptr) {
    const uint32_t c = *dc.ctr;
    dc.k0 ^= c * 0x9E3779B1u;
    dc.k1 += c * 0x85EBCA77u;
  }
  return dc;
}

// the two 32-bit words of a quad: r0 = lowbias32 of (counter ^ key ^ site), r1 = one more multiply-xorshift of (r0 ^ key')
__device__ __forceinline__ void drop_words(const DropCfg& dc, uint32_t site, uint64_t quad, uint32_t& r0, uint32_t& r1) {
  const uint32_t lo = (uint32_t)quad, hi = (uint32_t)(quad >> 32);
  r0 = mix32(lo ^ ((hi << 16) | (hi >> 16)) ^ dc.k0 ^ (site * 0x85EBCA77u));     // (site is a literal at every call site: folded)
  uint32_t x = (r0 ^ dc.k1) * 0x9E3779B1u;
  r1 = x ^ (x >> 15);
}

// keep-flags (bit j set = keep) for elements 4*quad .. 4*quad+3 of dropout site `site`
__device__ __forceinline__ uint32_t drop_keep4(const DropCfg& dc, uint32_t site, uint64_t quad) {
  uint32_t r0, r1;
  drop_words(dc, site, quad, r0, r1);
  // 16-bit fields compared in place: (r >> 16) >= t  <=>  r >= (t << 16);  (r & 0xFFFF) >= t  <=>  (r << 16) >= (t << 16)
  const uint32_t t16 = dc.thresh << 16;
  uint32_t m = 0;
  m |= ((r0 << 16) >= t16 ? 1u : 0u);
  m |= (r0 >= t16 ? 2u : 0u);
  m |= ((r1 << 16) >= t16 ? 4u : 0u);
  m |= (r1 >= t16 ? 8u : 0u);
  return m;
}

// the same decision as per-element multipliers (scale or 0): saves the mask round trip in the hot loops
__device__ __forceinline__ f32x4 drop_mul4(const DropCfg& dc, uint32_t site, uint64_t quad) {
  uint32_t r0, r1;
  drop_words(dc, site, quad, r0, r1);
  const uint32_t t16 = dc.thresh << 16;            // see drop_keep4
  f32x4 m;
  m[0] = (r0 << 16) >= t16 ? dc.scale : 0.0f;
  m[1] = r0 >= t16 ? dc.scale : 0.0f;
  m[2] = (r1 << 16) >= t16 ? dc.scale : 0.0f;
  m[3] = r1 >= t16 ? dc.scale : 0.0f;
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

__device__ __forceinline__ int clamp_len(int v, int S) { return v < 0 ? 0 : (v > S ? S : v); }

// Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with a private L2.  A kernel whose NEIGHBOURING work items touch
// the same cache lines (attn_bwd: the 15 heads of a token row are 40-byte pieces of one 600-byte row, read and written by 15 waves)
// renumbers its workgroups XCD-major: the returned id is this workgroup's position when the grid is listed XCD by XCD, so consecutive
// ids share an L2 -- the pieces of a line are then fetched once and merged into whole-line writes instead of once per XCD.
__device__ __forceinline__ int xcd_major_block(int b, int nb) {
  const int x = b & 7, q = nb >> 3, r = nb & 7;        // XCD x holds q + (x < r) workgroups
  return x * q + (x < r ? x : r) + (b >> 3);
}

__device__ __forceinline__ u16x4 pack4(f32x4 v) {
  u16x4 o;
  o[0] = f2bf(v[0]); o[1] = f2bf(v[1]); o[2] = f2bf(v[2]); o[3] = f2bf(v[3]);
  return o;
}

// ORs a 32-bit pattern into elements 4 | 5 (dword 2) of an operand fragment: one v_or_b32
__device__ __forceinline__ u16x8 or_dword2(u16x8 f, uint32_t v) {
  typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
  u32x4v x = __builtin_bit_cast(u32x4v, f);
  x[2] |= v;
  return __builtin_bit_cast(u16x8, x);
}

__device__ __forceinline__ u16x8 cat8(u16x4 lo, u16x4 hi) {
  u16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}

// "Tile order" of a bf16 MFMA operand M[R][K] (R % 16 == 0, K % 32 == 0): the 16 x 32 block (row tile, k-step) is stored as the 64
// lanes' 16-byte fragments back to back, i.e. element (r, k) sits at tile_off(r, k, K).  A wave then fetches one operand fragment
// per k-step as ONE contiguous 1 KB request.  From row-major rows the same fragment is 64 separate 16-byte pieces in 16 different
// cache lines with consecutive lanes in different rows, which the texture-address unit serialises: the GRU step kernels ran at 4x their
// L1 request-rate floor that way (22 us), independent of prefetch depth, XCD locality or occupancy.  Every packed weight operand
// (Wp, Wap, WaT, Wc/Wd, W_hh) is stored this way; wave fragment (row tile T, k-step ks) = base + (T * (K/32) + ks) * 512 + lane * 8.
__device__ __host__ __forceinline__ size_t tile_off(int r, int k, int K) {
  return ((size_t)(r >> 4) * (K >> 5) + (k >> 5)) * 512 + (((k & 31) >> 3) * 16 + (r & 15)) * 8 + (k & 7);
}

}  // namespace nr
