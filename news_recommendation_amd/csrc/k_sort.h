// Stable LSD radix sort of token ids (keys < num_rows, i.e. 17-19 bits) with their positions, for the embedding backward
// (autograd of nn.Embedding, src/model/NRMS/news_encoder.py:38: the segmented scatter reduces all occurrences of a row from adjacent
// positions, csrc/k_bwd.h embed_scatter_sorted_kernel).  Replaces torch.sort on int64 keys (rocPRIM merge sort: ~0.32 ms of GPU time per
// NRMS step at 542,720 tokens): the keys only carry ceil(log2(num_rows)) bits, so 2 (3 from 2^18 rows) counting-sort passes of <= 9
// bits suffice.
//
// One pass = histogram kernel -> single-workgroup exclusive scan of hist[digit][tile] -> scatter kernel.  Ranking inside a tile is
// wave-synchronous: a wave walks its 64 * KPT consecutive keys in rounds of 64; in a round the lanes with the same digit find each other
// with RB ballots (match mask), rank = popcount of the lower lanes in the mask, and the wave's running per-digit counter (LDS) supplies
// the offset of earlier rounds -- no atomics, stable by construction, and the 45 % padding tokens (digit 0 in every pass) cost the same
// as any other key.
#pragma once
#include "nr_common.h"

namespace nr {

constexpr int SORT_KPT = 8;                    // keys per thread
constexpr int SORT_TILE = WG * SORT_KPT;       // 2048 keys per workgroup
constexpr int SORT_MAXBINS = 512;              // <= 9 bits per pass

struct SortPass {
  const int64_t* ids;        // first pass: raw ids (clamped to [0, num_rows-1] like the forward gather); else nullptr
  const uint32_t* key_in;    // later passes
  const uint32_t* idx_in;
  uint32_t* key_out;         // intermediate passes
  uint32_t* idx_out;
  int64_t* ids_sorted;       // last pass
  int64_t* perm;
  int* hist;                 // [bins][n_tiles]: counts after the histogram kernel, per-digit exclusive prefixes over the tiles after the scan
  int* bin_tot;              // [bins]: total per digit (after the scan kernel)
  int64_t n, num_rows;
  int n_tiles, shift, bits;
};

__device__ __forceinline__ uint32_t sort_key(const SortPass& s, int64_t i) {
  if (s.ids != nullptr) {
    int64_t id = s.ids[i];
    id = id < 0 ? 0 : (id >= s.num_rows ? s.num_rows - 1 : id);
    return (uint32_t)id;
  }
  return s.key_in[i];
}

// Ranks the wave's keys of this tile: digit[k] / rank[k] (position among the wave's keys with the same digit) for round k; on return
// wcnt[digit] holds the wave's total per digit.  wcnt = this wave's LDS counter row (bins ints, zeroed by the caller).
__device__ __forceinline__ void sort_rank_wave(const SortPass& s, int64_t wave_base, int* wcnt, uint32_t (&key)[SORT_KPT],
                                               int (&digit)[SORT_KPT], int (&rank)[SORT_KPT]) {
  const int l = lane_id();
  const uint64_t lt = l == 0 ? 0ull : (~0ull >> (64 - l));
  const uint32_t dmask = (1u << s.bits) - 1u;
#pragma unroll
  for (int k = 0; k < SORT_KPT; ++k) {
    const int64_t i = wave_base + k * 64 + l;
    const bool valid = i < s.n;
    key[k] = valid ? sort_key(s, i) : 0u;
    const int d = (int)((key[k] >> s.shift) & dmask);
    uint64_t match = ballot(valid);
    for (int b = 0; b < s.bits; ++b) {
      const bool bit = (d >> b) & 1;
      const uint64_t bal = ballot(valid && bit);
      match &= bit ? bal : ~bal;
    }
    const int before = valid ? wcnt[d] : 0;
    wave_barrier();                                                     // every lane has read the counter before the leaders bump it
    if (valid && (match & lt) == 0) wcnt[d] = before + __builtin_popcountll(match);
    wave_barrier();
    digit[k] = valid ? d : -1;
    rank[k] = before + __builtin_popcountll(match & lt);
  }
}

constexpr int SORT_SMEM = 4 * SORT_MAXBINS * 4;     // one counter row per wave
constexpr int SORT_SCATTER_SMEM = SORT_SMEM + SORT_MAXBINS * 4 + WG * 4;     // + digit bases + scan scratch

__global__ __launch_bounds__(256) void sort_hist_kernel(SortPass s) {
  NR_SMEM_DECL(smem);
  int (*cnt)[SORT_MAXBINS] = (int (*)[SORT_MAXBINS])smem;
  const int bins = 1 << s.bits, w = wave_id();
  for (int i = threadIdx.x; i < 4 * SORT_MAXBINS; i += WG) (&cnt[0][0])[i] = 0;
  __syncthreads();
  uint32_t key[SORT_KPT];
  int digit[SORT_KPT], rank[SORT_KPT];
  sort_rank_wave(s, (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * 64 * SORT_KPT, cnt[w], key, digit, rank);
  __syncthreads();
  for (int d = threadIdx.x; d < bins; d += WG) s.hist[(size_t)d * s.n_tiles + blockIdx.x] = cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
}

// Exclusive prefix sums of the tile counts, one workgroup per digit: hist[d][0..n_tiles) -> in-place exclusive scan over the tiles
// (coalesced 256-wide sweeps with a running carry), bin_tot[d] = the digit's total.  The digits' own prefix (<= 512 values) is formed by
// every scatter workgroup in LDS.  (A single workgroup walking all bins x tiles counters took 0.49 ms per pass on MI355X: 530 dependent,
// uncoalesced global loads per thread.)
__device__ __forceinline__ int block_scan_incl(int* part, int t, int v) {       // Hillis-Steele over WG values; returns the inclusive sum
  part[t] = v;
  __syncthreads();
  for (int off = 1; off < WG; off <<= 1) {
    const int u = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += u;
    __syncthreads();
  }
  return part[t];
}

__global__ __launch_bounds__(256) void sort_scan_kernel(int* __restrict__ hist, int* __restrict__ bin_tot, int n_tiles) {
  NR_SMEM_DECL(smem);
  int* part = (int*)smem;                            // WG ints
  const int t = threadIdx.x;
  int* row = hist + (size_t)blockIdx.x * n_tiles;
  int carry = 0;
  for (int base = 0; base < n_tiles; base += WG) {
    const int i = base + t;
    const int v = i < n_tiles ? row[i] : 0;
    const int incl = block_scan_incl(part, t, v);
    if (i < n_tiles) row[i] = carry + incl - v;
    carry += part[WG - 1];
    __syncthreads();                                 // part[] is rewritten by the next sweep
  }
  if (t == 0) bin_tot[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(SortPass s) {
  NR_SMEM_DECL(smem);
  int (*cnt)[SORT_MAXBINS] = (int (*)[SORT_MAXBINS])smem;
  const int bins = 1 << s.bits, w = wave_id(), l = lane_id();
  for (int i = threadIdx.x; i < 4 * SORT_MAXBINS; i += WG) (&cnt[0][0])[i] = 0;
  __syncthreads();
  uint32_t key[SORT_KPT];
  int digit[SORT_KPT], rank[SORT_KPT];
  const int64_t wave_base = (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * 64 * SORT_KPT;
  sort_rank_wave(s, wave_base, cnt[w], key, digit, rank);
  __syncthreads();
  // digit bases: exclusive scan of the <= 512 digit totals (two digits per thread), in LDS behind the counters
  int* dbase = (int*)(smem + SORT_SMEM);             // [SORT_MAXBINS]
  int* part = dbase + SORT_MAXBINS;                  // [WG]
  {
    const int t = threadIdx.x;
    const int v0 = 2 * t < bins ? s.bin_tot[2 * t] : 0, v1 = 2 * t + 1 < bins ? s.bin_tot[2 * t + 1] : 0;
    const int incl = block_scan_incl(part, t, v0 + v1);
    dbase[2 * t] = incl - v0 - v1;
    dbase[2 * t + 1] = incl - v1;
  }
  __syncthreads();
  // per-digit base of each wave: digit base + prefix of (digit, this tile) + the totals of the lower waves
  for (int d = threadIdx.x; d < bins; d += WG) {
    int run = dbase[d] + s.hist[(size_t)d * s.n_tiles + blockIdx.x];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { const int c = cnt[ww][d]; cnt[ww][d] = run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SORT_KPT; ++k) {
    if (digit[k] < 0) continue;
    const int64_t i = wave_base + k * 64 + l;
    const int64_t pos = (int64_t)cnt[w][digit[k]] + rank[k];
    const uint32_t idx = s.ids != nullptr ? (uint32_t)i : s.idx_in[i];
    if (s.ids_sorted != nullptr) { s.ids_sorted[pos] = (int64_t)key[k]; s.perm[pos] = (int64_t)idx; }
    else { s.key_out[pos] = key[k]; s.idx_out[pos] = idx; }
  }
}

}  // namespace nr
