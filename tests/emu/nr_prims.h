// Wave-level CPU emulator of the device primitives used by the engine's HIP
// kernels.  TEST INFRASTRUCTURE ONLY: it exists so that kernel *logic* (tiling,
// fragment indexing, barriers, reductions) can be debugged bit-for-bit in the
// GPU-less build container.  It shadows news_recommendation_amd/csrc/nr_prims.h
// on the include path of the emulation build (tests/emu/build_emu.sh) and is
// never compiled into, or loaded by, the product library.
//
// Execution model: one OS thread; every GPU thread of a workgroup is a fiber
// (hand-rolled x86-64 context switch).  Fibers run until they reach a
// collective (workgroup barrier, wave shuffle, MFMA) and then yield, so a
// missing barrier shows up as stale data just as (more aggressively than) on
// hardware.  Workgroups run one after another.
//
// MFMA semantics follow /opt/skills/guides/cdna_hip_programming.md section 3:
//   v_mfma_f32_16x16x32_bf16: lane l holds A[i=l&15][k=(l>>4)*8+j], B[k=(l>>4)*8+j][n=l&15],
//   C/D: col = l&15, row = (l>>4)*4 + reg.
#pragma once
#include <utility>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <functional>

#define NR_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

namespace nr_emu {

struct Dim3 { unsigned x, y, z; };

struct PendingCopy { const void* src; void* dst; };

struct Fiber {
  void* sp = nullptr;
  unsigned char* stack = nullptr;
  bool done = false;
  Dim3 tid{0, 0, 0};
  std::vector<PendingCopy> dma;   // issued, not yet landed global -> LDS copies of this lane (oldest first)
};

struct WaveState {
  int arrived = 0;
  unsigned gen = 0;
  uint32_t stage[64][12];   // a(4 dwords) b(4 dwords) c(4 dwords) or shuffle payload
};

struct BlockState {
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  int nthreads = 0;
  int alive = 0;
  int arrived = 0;
  unsigned gen = 0;
  int cur = 0;
  void* sched_sp = nullptr;
  Dim3 bid{0, 0, 0}, bdim{1, 1, 1}, gdim{1, 1, 1};
  unsigned char* smem = nullptr;
  size_t smem_bytes = 0;
  std::function<void()> body;
};

extern BlockState* g_blk;

extern "C" void nr_emu_switch(void** save_sp, void* load_sp);

inline Fiber& cur_fiber() { return g_blk->fibers[g_blk->cur]; }
inline void yield() { Fiber& f = cur_fiber(); nr_emu_switch(&f.sp, g_blk->sched_sp); }

// LDS-DMA model: a copy lands as LATE as the program allows -- at the s_waitcnt vmcnt(n) that covers it (all but the newest n of the lane's
// copies land), at a __syncthreads() (the compiler's fence drains vmcnt) or when the lane exits -- so that a wait that is too weak, or a
// raw barrier used where a drain was needed, shows up as stale LDS data.
inline void dma_land(size_t keep) {
  Fiber& f = cur_fiber();
  if (f.dma.size() <= keep) return;
  const size_t n = f.dma.size() - keep;
  for (size_t i = 0; i < n; ++i)
    if (f.dma[i].src != nullptr) memcpy(f.dma[i].dst, f.dma[i].src, 16);        // (nullptr: a buffer STORE -- it only occupies a slot of the in-order counter)
  f.dma.erase(f.dma.begin(), f.dma.begin() + n);
}
inline void dma_issue(const void* src, void* dst) { cur_fiber().dma.push_back(PendingCopy{src, dst}); }

inline void block_sync_raw() {
  BlockState* b = g_blk;
  unsigned my = b->gen;
  if (++b->arrived == b->alive) { b->arrived = 0; b->gen++; return; }
  while (b->gen == my) yield();
}
inline void block_sync() {
  dma_land(0);
  block_sync_raw();
}
inline void wave_sync() {
  BlockState* b = g_blk;
  int w = b->cur / 64;
  WaveState& ws = b->waves[w];
  int lanes = std::min(64, b->nthreads - w * 64);
  unsigned my = ws.gen;
  if (++ws.arrived == lanes) { ws.arrived = 0; ws.gen++; return; }
  while (ws.gen == my) yield();
}

void launch(Dim3 grid, Dim3 block, size_t smem_bytes, std::function<void()> body);

}  // namespace nr_emu

#define threadIdx (nr_emu::cur_fiber().tid)
#define blockIdx (nr_emu::g_blk->bid)
#define blockDim (nr_emu::g_blk->bdim)
#define gridDim (nr_emu::g_blk->gdim)
#define __syncthreads() nr_emu::block_sync()

// dynamic LDS
#define NR_SMEM_DECL(name) unsigned char* name = nr_emu::g_blk->smem

// launch: NR_LAUNCH(kernel, grid_x, block_x, smem_bytes, stream, args...)
#define NR_LAUNCH(kern, gx, bx, smem, stream, ...)                                           \
  nr_emu::launch(nr_emu::Dim3{(unsigned)(gx), 1, 1}, nr_emu::Dim3{(unsigned)(bx), 1, 1},     \
                 (size_t)(smem), [&]() { kern(__VA_ARGS__); })

#define NR_SCHED_BARRIER() ((void)0)
#define NR_OPAQUE(x) ((void)0)
#define NR_ONE_WAVE_PER_SIMD

// emulation of global_load_lds_dwordx4: every lane copies its 16 B to lds_base + 16 * lane
#define NR_GLDS16(gptr, lds_base) nr_emu::dma_issue((const void*)(gptr), (unsigned char*)(lds_base) + 16 * (threadIdx.x & 63))
#define NR_GLDS16_S(base, voff, lds_dst) nr_emu::dma_issue((const unsigned char*)(base) + (voff), (unsigned char*)(lds_dst) + 16 * (threadIdx.x & 63))
__forceinline__ uint32_t lds_addr32(const void*) { return 0; }
#define NR_GLDS16_L(base, voff, lds_base, lds_base32, off) NR_GLDS16_S(base, voff, (unsigned char*)(lds_base) + (off))
#define NR_GLDS4_S(base, voff, lds_dst) nr_emu::dma_issue(nullptr, nullptr)
#define NR_GLDS4(gptr, lds_base) nr_emu::dma_issue(nullptr, nullptr)      // a cache-line touch: one slot of the in-order counter, no data anybody reads
#define NR_WAIT_VMCNT(n) nr_emu::dma_land(n)
#define NR_WAIT_LGKMCNT(n) ((void)0)
#define NR_BARRIER_RAW() nr_emu::block_sync_raw()

#define NR_LAUNCH2(kern, gx, gy, bx, smem, stream, ...)                                                \
  nr_emu::launch(nr_emu::Dim3{(unsigned)(gx), (unsigned)(gy), 1}, nr_emu::Dim3{(unsigned)(bx), 1, 1},     \
                 (size_t)(smem), [&]() { kern(__VA_ARGS__); })

namespace nr {

typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__forceinline__ int lane_id_fresh() { return (int)(threadIdx.x & 63); }
__forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

__forceinline__ float bf2f(u16 h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
__forceinline__ u16 f2bf(float f) {   // round-to-nearest-even, as v_cvt_pk_bf16_f32
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

__forceinline__ f32x4 mfma_16x16x32_bf16(u16x8 a, u16x8 b, f32x4 c) {
  nr_emu::BlockState* blk = nr_emu::g_blk;
  int l = lane_id();
  nr_emu::WaveState& ws = blk->waves[blk->cur / 64];
  memcpy(&ws.stage[l][0], &a, 16);
  memcpy(&ws.stage[l][4], &b, 16);
  nr_emu::wave_sync();
  // D[row][col], col = l&15, row = (l>>4)*4 + r ; A[i][k], lane (i, g) holds k = g*8+j
  f32x4 d = c;
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int g = 0; g < 4; ++g) {
      u16 av[8], bv[8];
      memcpy(av, &ws.stage[g * 16 + row][0], 16);
      memcpy(bv, &ws.stage[g * 16 + col][4], 16);
      for (int j = 0; j < 8; ++j) acc += bf2f(av[j]) * bf2f(bv[j]);
    }
    d[r] = acc;
  }
  nr_emu::wave_sync();
  return d;
}

// v_mfma_f32_32x32x16_bf16 (cdna_hip_programming.md section 3): lane l holds A[i = l & 31][k = 8 (l >> 5) + j], B[k][n = l & 31];
// C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)
__forceinline__ f32x16 mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c) {
  nr_emu::BlockState* blk = nr_emu::g_blk;
  int l = lane_id();
  nr_emu::WaveState& ws = blk->waves[blk->cur / 64];
  memcpy(&ws.stage[l][0], &a, 16);
  memcpy(&ws.stage[l][4], &b, 16);
  nr_emu::wave_sync();
  f32x16 d = c;
  const int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int h = 0; h < 2; ++h) {
      u16 av[8], bv[8];
      memcpy(av, &ws.stage[h * 32 + row][0], 16);
      memcpy(bv, &ws.stage[h * 32 + col][4], 16);
      for (int j = 0; j < 8; ++j) acc += bf2f(av[j]) * bf2f(bv[j]);
    }
    d[r] = acc;
  }
  nr_emu::wave_sync();
  return d;
}

// ds_read_b64_tr_b16 (see csrc/nr_prims.h): in each group of 16 lanes, lane i gets element j = P_{4 j + i / 4}[i % 4] of the 16 pieces P
__forceinline__ u16x4 lds_tr16_b64(const u16* piece) {
  nr_emu::BlockState* blk = nr_emu::g_blk;
  const int l = lane_id();
  nr_emu::WaveState& ws = blk->waves[blk->cur / 64];
  memcpy(&ws.stage[l][0], piece, 8);
  nr_emu::wave_sync();
  u16x4 o;
  const int grp = l & ~15, i = l & 15;
  for (int j = 0; j < 4; ++j) {
    u16 pc[4];
    memcpy(pc, &ws.stage[grp + 4 * j + i / 4][0], 8);
    o[j] = pc[i % 4];
  }
  nr_emu::wave_sync();
  return o;
}

template <int OFF>
__forceinline__ u16x4 lds_tr16_b64_async(const u16* piece) { return lds_tr16_b64((const u16*)((const unsigned char*)piece + OFF)); }     // (the emulator's LDS reads complete at once)

template <int OFF>
__forceinline__ u16x8 lds_read16_async(const void* p) { u16x8 v; memcpy(&v, (const unsigned char*)p + OFF, 16); return v; }
__forceinline__ void lds_add_f32(float* p, float v) { *p += v; }
template <int V> struct StaticIdx { static constexpr int value = V; };
template <class F, int... Is> __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(StaticIdx<Is>{}), ...); }
template <int N, class F> __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__forceinline__ float shfl_xor(float v, int mask) {
  nr_emu::BlockState* blk = nr_emu::g_blk;
  int l = lane_id();
  nr_emu::WaveState& ws = blk->waves[blk->cur / 64];
  memcpy(&ws.stage[l][0], &v, 4);
  nr_emu::wave_sync();
  float r; memcpy(&r, &ws.stage[(l ^ mask) & 63][0], 4);
  nr_emu::wave_sync();
  return r;
}
__forceinline__ float sum_rows4(float v) { v += shfl_xor(v, 32); v += shfl_xor(v, 16); return v; }
__forceinline__ float shfl(float v, int src) {
  nr_emu::BlockState* blk = nr_emu::g_blk;
  int l = lane_id();
  nr_emu::WaveState& ws = blk->waves[blk->cur / 64];
  memcpy(&ws.stage[l][0], &v, 4);
  nr_emu::wave_sync();
  float r; memcpy(&r, &ws.stage[src & 63][0], 4);
  nr_emu::wave_sync();
  return r;
}

__forceinline__ uint32_t shfl_u32(uint32_t v, int src) {
  float f; memcpy(&f, &v, 4);
  f = shfl(f, src);
  uint32_t r; memcpy(&r, &f, 4);
  return r;
}

template <int MODE>
__forceinline__ float row_xchg(float v) {            // partner: lane ^ 1, lane ^ 2, i <-> 7 - i (groups of 8), i <-> 15 - i (rows of 16)
  nr_emu::BlockState* blk = nr_emu::g_blk;
  const int l = lane_id();
  nr_emu::WaveState& ws = blk->waves[blk->cur / 64];
  const int partner = MODE == 0 ? (l ^ 1) : MODE == 1 ? (l ^ 2) : MODE == 2 ? ((l & ~7) | (7 - (l & 7))) : ((l & ~15) | (15 - (l & 15)));
  memcpy(&ws.stage[l][0], &v, 4);
  nr_emu::wave_sync();
  float o; memcpy(&o, &ws.stage[partner][0], 4);
  nr_emu::wave_sync();
  return o;
}
__forceinline__ float sum_row16(float v) {
  v += row_xchg<0>(v); v += row_xchg<1>(v); v += row_xchg<2>(v); v += row_xchg<3>(v);
  return v;
}
__forceinline__ void wave_barrier() { nr_emu::wave_sync(); }
__forceinline__ void fence_agent() {}

__forceinline__ float fast_exp(float x) { return expf(x); }
__forceinline__ float fast_tanh(float x) { return tanhf(x); }
__forceinline__ float fast_exp2(float x) { return exp2f(x); }
__forceinline__ float fast_rcp(float x) { return 1.0f / x; }

__forceinline__ void atomic_add(float* p, float v) { *p += v; }
__forceinline__ int atomic_exch(int* p, int v) { int o = *p; *p = v; return o; }
__forceinline__ int atomic_add_i32(int* p, int v) { int o = *p; *p += v; return o; }
__forceinline__ uint32_t ld_relaxed_u32(const uint32_t* p) { return *p; }
__forceinline__ void atomic_cas_u32(uint32_t* p, uint32_t expect, uint32_t v) { if (*p == expect) *p = v; }
__forceinline__ int uniform(int v) { return v; }
__forceinline__ uint64_t ballot(bool pred) {
  nr_emu::BlockState* blk = nr_emu::g_blk;
  int l = lane_id();
  nr_emu::WaveState& ws = blk->waves[blk->cur / 64];
  ws.stage[l][0] = pred ? 1u : 0u;
  nr_emu::wave_sync();
  uint64_t m = 0;
  int lanes = std::min(64, blk->nthreads - (blk->cur / 64) * 64);
  for (int i = 0; i < lanes; ++i) m |= (uint64_t)(ws.stage[i][0] & 1u) << i;
  nr_emu::wave_sync();
  return m;
}

// raw buffer access (see csrc/nr_prims.h): offsets at or past the byte count read zeros / store nothing
struct BufRsrc { unsigned char* base; uint32_t nbytes; };
__forceinline__ BufRsrc make_buf(const void* base, uint32_t nbytes) { return BufRsrc{(unsigned char*)base, nbytes}; }
template <int IMM = 0> __forceinline__ u16x8 buf_load16(BufRsrc r, uint32_t off, uint32_t soff = 0) {
  u16x8 v = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  if ((uint64_t)off + IMM + soff + 16 <= r.nbytes) memcpy(&v, r.base + off + IMM + soff, 16);
  return v;
}
template <int IMM = 0> __forceinline__ f32x4 buf_load16f(BufRsrc r, uint32_t off) {
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if ((uint64_t)off + IMM + 16 <= r.nbytes) memcpy(&v, r.base + off + IMM, 16);
  return v;
}
__forceinline__ float buf_load4f(BufRsrc r, uint32_t off) {
  float v = 0.0f;
  if ((uint64_t)off + 4 <= r.nbytes) memcpy(&v, r.base + off, 4);
  return v;
}
// A buffer store is one more entry of the wave's in-order memory counter (gfx950: vmcnt counts stores too): kernels that leave stores in flight
// across a COUNTED wait for their LDS-DMA copies (qkv_proj_kernel) count on that, so the emulator counts them as well
template <int IMM = 0> __forceinline__ void buf_store16(BufRsrc r, uint32_t off, u16x8 v, uint32_t soff = 0) {
  if ((uint64_t)off + IMM + soff + 16 <= r.nbytes) memcpy(r.base + off + IMM + soff, &v, 16);
  nr_emu::dma_issue(nullptr, nullptr);
}
template <int IMM = 0> __forceinline__ void buf_store8(BufRsrc r, uint32_t off, u16x4 v, uint32_t soff = 0) {
  if ((uint64_t)off + IMM + soff + 8 <= r.nbytes) memcpy(r.base + off + IMM + soff, &v, 8);
  nr_emu::dma_issue(nullptr, nullptr);
}

__forceinline__ void buf_store4f(BufRsrc r, uint32_t off, float v, uint32_t soff = 0) {
  if ((uint64_t)off + soff + 4 <= r.nbytes) memcpy(r.base + off + soff, &v, 4);
  nr_emu::dma_issue(nullptr, nullptr);
}
__forceinline__ void buf_store2(BufRsrc r, uint32_t off, u16 v, uint32_t soff = 0) {
  if ((uint64_t)off + soff + 2 <= r.nbytes) memcpy(r.base + off + soff, &v, 2);
  nr_emu::dma_issue(nullptr, nullptr);
}

template <typename T> __forceinline__ T ld_nt(const T* p) { return *p; }
template <typename T> __forceinline__ void st_nt(T* p, T v) { *p = v; }

inline int set_max_dynamic_lds(const void*, int) { return 0; }
inline int device_cus() { return 3; }      // few "CUs": the persistent kernels run several iterations per wave even in small tests

__forceinline__ uint32_t mulhi_u32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }


}  // namespace nr
