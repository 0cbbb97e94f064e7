// Device primitives for the gfx950 (CDNA4 / MI355X) kernels of the NRMS scoring engine.
// Wave = 64 lanes.  MFMA fragment layout (v_mfma_f32_16x16x32_bf16):
//   A: lane l holds A[i = l&15][k = (l>>4)*8 + j], j = 0..7   (8 bf16 = 4 VGPRs)
//   B: lane l holds B[k = (l>>4)*8 + j][n = l&15]
//   C/D: lane l, reg r  ->  row = (l>>4)*4 + r, col = l&15      (4 fp32)
#pragma once
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

#define NR_SMEM_DECL(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

#define NR_LAUNCH(kern, gx, bx, smem, stream, ...) \
  hipLaunchKernelGGL(kern, dim3((unsigned)(gx)), dim3((unsigned)(bx)), (size_t)(smem), (stream), __VA_ARGS__)

// one wave per SIMD: the whole 512-entry register file for a register-resident kernel
#define NR_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// makes a 32-bit per-lane value opaque to the optimiser at this point: it is recomputed here instead of being hoisted out of the enclosing loop
#define NR_OPAQUE(x) asm volatile("" : "+v"(x))
#define NR_ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 1)))

// async global -> LDS copy of 16 B per lane (global_load_lds_dwordx4): lane i's 16 B land at lds_base + 16 i (wave-uniform base,
// lane-linear destination); completion is tracked by vmcnt and drained by the next __syncthreads()
#define NR_GLDS16(gptr, lds_base) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lds_base), 16, 0, 0)

// the 4-byte form (lane i's dword lands at lds_base + 4 i): used to TOUCH cache lines -- an asynchronous prefetch into the L2 whose landing area
// nobody reads
#define NR_GLDS4(gptr, lds_base) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lds_base), 4, 0, 0)

// The same copies with the source as (wave-uniform 64-bit base in SGPRs) + (32-bit byte offset per lane): written in asm because the builtin
// takes one flat pointer and the compiler then keeps a 64-bit VGPR address pair per copy stream alive across the loop.  Invisible to the
// compiler's waitcnt bookkeeping: the caller waits (NR_WAIT_VMCNT) before it reads the landing area AND before any barrier that is to publish
// it -- __syncthreads() does not drain copies it cannot see.  M0 (the LDS destination) is saved and restored; s_nop 4 covers a base freshly
// written by a v_readfirstlane.
__device__ __forceinline__ void glds16_sbase(const void* base, uint32_t voff, void* lds_dst) {
  uint32_t keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds_dst)) : "memory");      // (low 32 bits of a generic LDS pointer = the LDS byte address)
}
__device__ __forceinline__ void glds4_sbase(const void* base, uint32_t voff, void* lds_dst) {
  uint32_t keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds_dst)) : "memory");      // (low 32 bits of a generic LDS pointer = the LDS byte address)
}

// The lean form for loops that issue a copy per MFMA: M0 is declared clobbered instead of saved / restored and the 5-slot nop in front is gone --
// the caller's base and destination must come from SCALAR arithmetic (a base freshly written by v_readfirstlane needs the form above).
__device__ __forceinline__ void glds16_lean(const void* base, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
}
// (destination = LDS byte address of `lds_base`, read once by the caller with lds_addr32(), + a scalar byte offset)
__device__ __forceinline__ uint32_t lds_addr32(const void* lds_ptr) { return (uint32_t)(uintptr_t)lds_ptr; }
#define NR_GLDS16_L(base, voff, lds_base, lds_base32, off) glds16_lean((base), (voff), (lds_base32) + (uint32_t)(off))
#define NR_GLDS16_S(base, voff, lds_dst) glds16_sbase((base), (voff), (lds_dst))
#define NR_GLDS4_S(base, voff, lds_dst) glds4_sbase((base), (voff), (lds_dst))

// Counted wait for a RING of such copies: returns when at most n of this wave's vector-memory loads (LDS-DMA included; they complete in
// issue order) are still in flight.  s_waitcnt simm16 on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14; the other two
// counters are left unconstrained.  NR_BARRIER_RAW is s_barrier WITHOUT the fence __syncthreads() implies (that fence drains vmcnt to 0,
// i.e. the whole ring): the caller orders its own LDS traffic -- every wave waits for its own copies of a slot (NR_WAIT_VMCNT), then the
// barrier makes them mutually visible; the asm clobbers keep the compiler from moving LDS accesses across either.
#define NR_WAIT_VMCNT(n) do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | ((((n) >> 4) & 3) << 14)); asm volatile("" ::: "memory"); } while (0)
// likewise for the LDS counter: at most n of this wave's LDS operations (in issue order) still outstanding
#define NR_WAIT_LGKMCNT(n) __builtin_amdgcn_s_waitcnt(0xC07F | (((n) & 15) << 8))
#define NR_BARRIER_RAW() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

#define NR_LAUNCH2(kern, gx, gy, bx, smem, stream, ...) \
  hipLaunchKernelGGL(kern, dim3((unsigned)(gx), (unsigned)(gy)), dim3((unsigned)(bx)), (size_t)(smem), (stream), __VA_ARGS__)

namespace nr {

typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
// the same number from the hardware lane counter (all lanes active): a fresh definition that does not depend on the work-item id register, for
// a late phase of a kernel whose early phase would otherwise keep lane-derived values alive (or spill them) across its register peak
__device__ __forceinline__ int lane_id_fresh() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// wave index inside the workgroup as a SCALAR: threadIdx.x >> 6 is the same in all 64 lanes, but the compiler only knows that after
// readfirstlane; everything derived from it (tile / sequence / pair indices, base pointers, loop bounds, branches) then runs on the
// scalar unit instead of costing vector instructions in VALU-bound kernels.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ float bf2f(u16 h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }  // RNE (v_cvt_pk_bf16_f32)

__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(u16x8 a, u16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// v_mfma_f32_32x32x16_bf16: A[32][16], B[16][32]; lane l holds A[i = l & 31][k = 8 (l >> 5) + j], B[k = 8 (l >> 5) + j][n = l & 31], j = 0..7;
// C/D (16 fp32): col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5) -- four quads of 4 consecutive rows per lane.
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// ds_read_b64_tr_b16: LDS transpose read.  Every lane passes the 8-byte-aligned LDS address of a piece of 4 consecutive 16-bit elements;
// inside each group of 16 lanes the 16 pieces P_0 .. P_15 form a 4 x 16 matrix M[r][4 q + e] = P_{4 r + q}[e], and lane i of the group
// receives its COLUMN i: element j = M[j][i] = P_{4 j + i / 4}[i % 4].  With P_{4 r + q} = &T[k0 + r][n0 + 4 q] of a row-major tile T this
// hands lane i the four k-consecutive values T[k0 .. k0 + 3][n0 + i]: half of an MFMA operand fragment whose contraction index runs along
// the ROWS of T, without a transposed copy.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u16x4 lds_tr16_b64(const u16* piece) {
  return __builtin_bit_cast(u16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)piece));
}

// The same read as inline asm, for kernels that keep LDS-DMA copies in flight (a copy RING): given the builtin form, the compiler's waitcnt
// pass sees an LDS read without a memory operand and drains vmcnt to 0 in front of it whenever a global_load_lds is outstanding -- every
// fragment read then waits for the whole ring (measured: the TN form of gemm_ring_kernel at 0.23 of the MFMA peak).  Contract: the caller
// orders the read against the copies itself (counted vmcnt wait + barrier before the first read of a chunk) and waits for the RESULT with
// NR_WAIT_LGKMCNT before its first use -- the compiler inserts neither.
// OFF: compile-time byte offset added to `piece` through the instruction's offset field (no address arithmetic, no register per offset).
template <int OFF>
__device__ __forceinline__ u16x4 lds_tr16_b64_async(const u16* piece) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 8 == 0, "ds_read offset field");
  u16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)piece), "n"(OFF) : "memory");
  return v;
}

// A plain 16-byte LDS read under the same contract (result awaited by the caller with NR_WAIT_LGKMCNT): lets a kernel keep fragment reads of LATER
// k-steps in flight across its MFMAs.  (Given `a = frag(k); an = frag(k + 1); mfma(a ..); a = an`, the compiler under register pressure merges the two
// registers and waits lgkmcnt(0) right after each read: one LDS latency per k-step, k_pool4.h round 6.)
template <int OFF>
__device__ __forceinline__ u16x8 lds_read16_async(const void* p) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read offset field");
  u16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p), "n"(OFF) : "memory");
  return v;
}
// *p += v on an LDS float, no return value, nothing to wait for (ds_add_f32; the location belongs to this wave or the addition order does not matter)
__device__ __forceinline__ void lds_add_f32(float* p, float v) {
  asm volatile("ds_add_f32 %0, %1" : : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)p), "v"(v) : "memory");
}
// f(IntTag<0>{}), f(IntTag<1>{}), ... f(IntTag<N - 1>{}): a loop whose index is a compile-time constant in the body (instruction offset fields)
template <int V> struct StaticIdx { static constexpr int value = V; };
template <class F, int... Is> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(StaticIdx<Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
// sum over the four 16-lane rows of the wave (= over lane ^ 16 and lane ^ 32), result in every lane: two VALU lane swaps
// (v_permlane32_swap / v_permlane16_swap, gfx950) instead of two ds_bpermute round trips through the LDS crossbar
__device__ __forceinline__ float sum_rows4(float v) {
  // inline asm: the clang builtins return only the first of the two swapped registers (ROCm 7.2).  v_permlane32_swap exchanges
  // a[32:63] <-> b[0:31], v_permlane16_swap the odd 16-lane rows of a with the even rows of b; with a == b on entry,
  // a + b afterwards is the pairwise sum in every lane.  s_nop covers the VALU-write -> lane-swap hazards the assembler
  // does not see inside inline asm.
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  a += b;
  b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ uint32_t shfl_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
// Lane exchange inside a row of 16 lanes (lanes with equal l >> 4) as ONE DPP-modified move: MODE 0: lane ^ 1, 1: lane ^ 2 (quad_perm),
// 2: i <-> 7 - i inside each group of 8 (row_half_mirror), 3: i <-> 15 - i (row_mirror).  Every mode pairs lanes that differ in bit
// MODE of the lane index and agree in the higher bits: enough for butterflies and recursive halving, no ds_bpermute round trip.
template <int MODE>
__device__ __forceinline__ float row_xchg(float v) {
  constexpr int ctrl = MODE == 0 ? 0xB1 : MODE == 1 ? 0x4E : MODE == 2 ? 0x141 : 0x140;
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));
}
// sum over the 16 lanes of a row, result in every lane of the row
__device__ __forceinline__ float sum_row16(float v) {
  v += row_xchg<0>(v); v += row_xchg<1>(v); v += row_xchg<2>(v); v += row_xchg<3>(v);
  return v;
}

// orders this wave's LDS traffic for cross-lane exchange through LDS (hardware keeps a wave's DS ops in order;
// this only stops the compiler from moving accesses across the point)
__device__ __forceinline__ void wave_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// agent-scope release + acquire: this wave's global stores are visible device-wide and its L1 is invalidated, so data written to
// global memory by other waves (before a barrier) is re-read from L2, never from a stale L1 line
__device__ __forceinline__ void fence_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); }

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); exact limits at +-inf, abs error ~1e-7 in fp32
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ int atomic_exch(int* p, int v) { return atomicExch(p, v); }
__device__ __forceinline__ int atomic_add_i32(int* p, int v) { return atomicAdd(p, v); }
// a word other kernels / the host change behind this kernel's back (fault words): read from the L2, compare-and-swap there
__device__ __forceinline__ uint32_t ld_relaxed_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void atomic_cas_u32(uint32_t* p, uint32_t expect, uint32_t v) { atomicCAS(p, expect, v); }
// a value the program knows to be identical in all lanes, as a scalar (see wave_id())
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// 64-bit mask of the lanes whose predicate is true
__device__ __forceinline__ uint64_t ballot(bool pred) { return __ballot(pred); }

// Raw buffer access: a 128-bit resource in SGPRs (base pointer, byte count) + ONE 32-bit byte offset per lane + an immediate -- no 64-bit
// address pair per lane and no compare-and-branch around the access: an offset at or past the byte count reads zeros / stores nothing.
// (Kernels whose lanes keep 64-bit addresses per row spill them; a spill reload sits in the same in-order counter as the stores before it
// and waits for all of them.)  nbytes and off must be multiples of the access size.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_buf(const void* base, uint32_t nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nbytes, 0x00020000);
}
template <int IMM = 0> __device__ __forceinline__ u16x8 buf_load16(BufRsrc r, uint32_t off, uint32_t soff = 0) {
  return __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off + IMM, soff, 0));
}
template <int IMM = 0> __device__ __forceinline__ f32x4 buf_load16f(BufRsrc r, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off + IMM, 0, 0));
}
__device__ __forceinline__ float buf_load4f(BufRsrc r, uint32_t off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
// soff: a wave-uniform byte offset (scalar operand of the instruction)
// A store of more than 8 bytes reads its data registers AFTER it has issued: a VALU instruction that overwrites them needs wait states in between.
// LLVM inserts them -- except when the scalar-offset field holds a REGISTER (GCNHazardRecognizer::createsVALUHazard: "this hazard only exists if the
// instruction is not using a register in the soffset field"), which is every soff that is not an inline constant.  On MI355X the hazard exists there
// too: round 6, k_convgemm.h -- `buffer_store_dwordx4 v[82:85], v91, s[8:11], s51 offen` followed by `v_or_b32 v84, ..` stored the NEW v84 in the last
// lanes of each 16 (4 wrong rows per wave and tile; tools/isa_store_hazard_audit.py finds such pairs in the device assembly).  Guard: an empty-bodied
// two-wait-state asm that READS the data, so that no write to those registers can be scheduled in front of it.
template <int IMM = 0> __device__ __forceinline__ void buf_store16(BufRsrc r, uint32_t off, u16x8 v, uint32_t soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, off + IMM, soff, 0);
  if (!(__builtin_constant_p(soff) && soff <= 64u)) asm volatile("s_nop 1" : : "v"(v));
}
template <int IMM = 0> __device__ __forceinline__ void buf_store8(BufRsrc r, uint32_t off, u16x4 v, uint32_t soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, off + IMM, soff, 0);
}

__device__ __forceinline__ void buf_store4f(BufRsrc r, uint32_t off, float v, uint32_t soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, off, soff, 0);
}
__device__ __forceinline__ void buf_store2(BufRsrc r, uint32_t off, u16 v, uint32_t soff = 0) {
  __builtin_amdgcn_raw_buffer_store_b16((short)v, r, off, soff, 0);
}

template <typename T> __device__ __forceinline__ T ld_nt(const T* p) { return __builtin_nontemporal_load(p); }
template <typename T> __device__ __forceinline__ void st_nt(T* p, T v) { __builtin_nontemporal_store(v, p); }   // streaming store: not re-read soon

// > 64 KiB of dynamic LDS needs an opt-in per kernel function.  The attribute sticks, so the runtime call is made once per (kernel, size,
// device) and remembered: the GRU sweep alone issues ~100 such launches per training step.
inline int set_max_dynamic_lds(const void* kernel, int bytes) {
  struct Seen { const void* k; int bytes, dev; };
  static Seen seen[256];
  static int n_seen = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lk(mu);
  for (int i = 0; i < n_seen; ++i)
    if (seen[i].k == kernel && seen[i].dev == dev && seen[i].bytes >= bytes) return 0;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
  if (n_seen < 256) seen[n_seen++] = Seen{kernel, bytes, dev};
  return 0;
}

__device__ __forceinline__ uint32_t mulhi_u32(uint32_t a, uint32_t b) { return __umulhi(a, b); }

// compute units of the current device: the grid of the persistent kernels (one workgroup per CU)
inline int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus[dev];
}


}  // namespace nr
