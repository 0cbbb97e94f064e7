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
  int* hist;                 // [bins][n_tiles]: counts after the histogram kernel, exclusive global offsets after the scan
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

// in-place exclusive prefix sum of data[0..n) by ONE workgroup (n = bins * n_tiles <= a few 10^5 ints, L2 resident)
__global__ __launch_bounds__(256) void sort_scan_kernel(int* __restrict__ data, int64_t n) {
  NR_SMEM_DECL(smem);
  int* part = (int*)smem;                            // WG ints
  const int t = threadIdx.x;
  const int64_t chunk = (n + WG - 1) / WG, lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
  int sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += data[i];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < WG; off <<= 1) {          // Hillis-Steele inclusive scan of the 256 partial sums
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - sum;
  for (int64_t i = lo; i < hi; ++i) { const int v = data[i]; data[i] = run; run += v; }
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
  // per-digit base of each wave: global offset of (digit, this tile) + the totals of the lower waves
  for (int d = threadIdx.x; d < bins; d += WG) {
    int run = s.hist[(size_t)d * s.n_tiles + blockIdx.x];
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
