// XCD-local synchronisation for persistent kernels on gfx950 (MI355X: 8 XCDs x 32 CUs, one 4 MB L2 per XCD).
//
// A grid-wide barrier has to publish data ACROSS the eight non-coherent L2s (an L2 write-back + invalidate per phase: the round-2 persistent
// GRU lost 3 us per step to it).  Inside ONE XCD every CU talks to the same L2: data written with plain stores is in that L2 once the
// storing wave's vmcnt has drained, and a reader on another CU of the same XCD sees it as long as it does not go through its own (never
// refreshed) vector L1 -- an L1-bypassing load (nt / sc1: served by the L2) or an L1-bypassing global -> LDS copy.  So a phase boundary
// between workgroups of one XCD needs no cache maintenance at all: drain vmcnt, one L2 atomic per workgroup, a relaxed poll.
//
// Nothing here relies on WHICH XCD a workgroup lands on: a workgroup reads its own XCC id from the hardware register and joins that XCD's
// team (a ticket from that XCD's counter); a team must come out complete (NR_XCD_TEAM workgroups, checked: an incomplete or overfull team
// raises the error word and every wait gives up) -- the kernel's result is then garbage; it never hangs and never silently mixes data across
// L2s.  Every spin is bounded.  What happens to the garbage: besides the per-launch error word (zeroed with the team words before every launch)
// a failure ORs its code into a STICKY word that no launch clears (XcdSync::sticky; the process's fault words, nr_set_fault_words), and the
// optimiser kernels are gated on those words (k_optim.h): from the failed sweep on, no Adam update is applied until the host has looked
// (nr_fault_state), repeated the step on the step-per-launch kernels and cleared the words (train_fast.py).  Nothing falls back by itself.
#pragma once
#include "nr_common.h"

namespace nr {

constexpr int NR_XCDS = 8;
constexpr int NR_XCD_TEAM = 32;          // CUs (= resident workgroups of a one-per-CU grid) per XCD
constexpr uint32_t NR_XCD_SPIN_LIMIT = 1u << 22;      // polls (with s_sleep) before a wait gives up: ~1 s
constexpr uint32_t NR_XCD_SPIN_LIMIT_FAULT = 1u << 12; // ... of a launch with an injected fault (nr_debug_gru_fault): the test need not wait a second per barrier

// state words of one launch (zeroed by the launcher with a memset node before the kernel): [0..7] tickets, [8..15] arrival counters, [16] error
struct XcdSync {
  unsigned int* w;
  unsigned int* sticky;    // one word OUTSIDE the per-launch block: every error code is ORed into it as well and stays until the host clears it
                           // (nr_fault_clear); null in the probe
  int fault;               // debug (nr_debug_gru_fault): workgroup 0 never arrives at a barrier, its team mates' waits give up after a short spin
};

__device__ __forceinline__ void xcd_raise(XcdSync s, unsigned code) {
  __hip_atomic_fetch_or(s.w + 16, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // OR: a later code does not erase an earlier one
  if (s.sticky != nullptr) __hip_atomic_fetch_or(s.sticky, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int xcc_id() {
#ifdef NR_EMU
  return (int)(blockIdx.x & 7);
#else
  return (int)(__builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 0xF);      // XCC_ID: bits 3:0
#endif
}

// joins this workgroup to its XCD's team; returns the slot (0 .. NR_XCD_TEAM - 1), or -1 (error word raised) when the team is overfull.
// Every thread of the workgroup gets the same answer (through LDS word `bcast`).
__device__ __forceinline__ int xcd_join(XcdSync s, int xcd, int* bcast) {
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(s.w + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t >= (unsigned)NR_XCD_TEAM) xcd_raise(s, 1u);
    *bcast = t < (unsigned)NR_XCD_TEAM ? (int)t : -1;
  }
  __syncthreads();
  const int slot = *bcast;
  __syncthreads();
  return slot;
}

// Phase boundary between the workgroups of one XCD.  Contract: every wave has drained its own stores (s_waitcnt vmcnt(0)) BEFORE the call
// -- the stores are then in the XCD's L2; `phase` is the running number of the call, from 1 (the arrival counter only grows).  Returns false when the wait gave up (error word raised).
__device__ __forceinline__ bool xcd_barrier(XcdSync s, int xcd, unsigned phase) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0 && !(s.fault != 0 && blockIdx.x == 0)) {
    __hip_atomic_fetch_add(s.w + 8 + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = phase * (unsigned)NR_XCD_TEAM;
    const uint32_t limit = s.fault != 0 ? NR_XCD_SPIN_LIMIT_FAULT : NR_XCD_SPIN_LIMIT;
    uint32_t spins = 0;
    while (__hip_atomic_load(s.w + 8 + xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > limit || (spins & 1023u) == 0 && __hip_atomic_load(s.w + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        xcd_raise(s, 2u);
        ok = false;
        break;
      }
    }
  }
  __syncthreads();
  return ok;       // (only thread 0 knows; callers that must stop read the error word)
}

// ---- probe (tools/xcd_probe.py): each workgroup writes a record per phase, its team mates read all 32 records of the XCD back with L1-bypassing
// loads and check the stamps; out[wg] = number of stale / wrong words seen, out[256 + wg] = XCC id, out[512 + wg] = slot ------------------------
__global__ __launch_bounds__(512) void xcd_probe_kernel(XcdSync s, unsigned int* rec, unsigned int* out, int phases) {
  __shared__ int bcast;
  const int xcd = xcc_id();
  const int slot = xcd_join(s, xcd, &bcast);
  if (threadIdx.x == 0) { out[256 + blockIdx.x] = (unsigned)xcd; out[512 + blockIdx.x] = (unsigned)slot; }
  if (slot < 0) return;
  unsigned bad = 0, bar = 0;                            // bar: running number of the barrier call (the arrival counter only grows)
  unsigned int* mine = rec + ((size_t)xcd * NR_XCD_TEAM + slot) * 512;
  for (int ph = 1; ph <= phases; ++ph) {
    mine[threadIdx.x] = (unsigned)ph * 65536u + (unsigned)slot * 256u + (threadIdx.x & 255u);       // plain store, 2 KB per workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xcd_barrier(s, xcd, ++bar);
    for (int m = 0; m < NR_XCD_TEAM; ++m) {
      const unsigned v = ld_nt(rec + ((size_t)xcd * NR_XCD_TEAM + m) * 512 + threadIdx.x);
      bad += v != (unsigned)ph * 65536u + (unsigned)m * 256u + (threadIdx.x & 255u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xcd_barrier(s, xcd, ++bar);                           // everybody has read phase ph before anybody overwrites it
  }
  atomicAdd(out + blockIdx.x, bad);
}

}  // namespace nr
