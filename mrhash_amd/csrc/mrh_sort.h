// mrh_sort.h — stable LSD radix sort of the (voxel id, sdf) records of one LiDAR scan (mrh_lidar.h) and of the (position key,
// list entry) pairs of an extraction beyond k_block_rank's reach (mrh_mc.h).
//
// Why not rocPRIM here: a scan has ~10^5 .. 10^6 records and ~21 significant key bits.  rocPRIM's onesweep sort is built for
// 10^7+ items: at this size each of its three passes is a latency-bound chain (decoupled look-back over ~300 tiles, 26 us
// measured) and every call adds 5-7 hipMemsetAsync of look-back state (5 us each): 130 us of a 200 us scan
// (profiles/r03/lidar_kernel_stats_rocprim_sort.csv).  At this size the classic three-kernel pass is the better fit:
//   k_sort_hist      per tile (1024 records, one workgroup): 256-bin histogram of the pass's digit -> hist[digit][tile]
//   k_sort_scan      per digit (256 workgroups): exclusive scan of its row over the tiles + the row total; the scan over the
//                    256 totals is redone by every scatter workgroup in LDS
//   k_sort_scatter   per tile: every record's rank among the records of its digit that precede it — inside its wave by a
//                    ballot match (8 ballots give the lanes holding the same digit; rank = popcount of the lower ones), across
//                    the rounds of a wave and the waves of a tile by per-wave digit counters in LDS — plus the scanned
//                    histogram entry gives its final position.  Order of equal digits = tile, wave, round, lane = input
//                    order: the pass is STABLE, which is what keeps a voxel's records in ascending point index (D6).
// No atomics on global memory, no look-back, nothing to clear between calls except the LDS the kernels own.
#pragma once

#include "mrh_device.h"

namespace mrh {

constexpr int kSortTile = 1024;     // records per workgroup: a scan has ~10^6 records, smaller tiles mean more workgroups (640 for 0.65 M) and a short chain each
constexpr int kSortThreads = 256;   // 4 waves x 4 rounds x 64 lanes
constexpr int kSortRounds = kSortTile / kSortThreads;
constexpr u32 kSortScanMax = 1u << 30;  // histogram entries: 256 per tile, indexed in 32 bits like the records themselves (4 M tiles = every n < 2^32; the table
                                        // is sized from the scan, 1 KiB per 1 024 records); beyond: the call fails with MRH_ERR_CAPACITY

__device__ __forceinline__ u32 sort_lane() { return (u32) __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <typename K>
__global__ __launch_bounds__(kSortThreads) void k_sort_hist(const K* __restrict__ keys, const u32 n, const int shift, u32* __restrict__ hist,
                                                            const u32 ntiles) {
  __shared__ u32 s_h[256];
  const u32 tile = blockIdx.x;
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const u32 t0 = tile * kSortTile;
#pragma unroll 4
  for (int r = 0; r < kSortRounds; r++) {
    const u32 i = t0 + r * kSortThreads + threadIdx.x;
    if (i < n) atomicAdd(&s_h[(u32) (keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[threadIdx.x * ntiles + tile] = s_h[threadIdx.x];
}

// workgroup d: exclusive scan of digit d's row hist[d][0 .. ntiles) in place (records of that digit in earlier tiles) and the
// row's total -> totals[d].  256 workgroups, each a short block scan: the table is scanned in parallel, not by one workgroup
// walking 40 k entries (a first version did that: 60 us per pass, all of it load latency).
__global__ __launch_bounds__(256) void k_sort_scan(u32* __restrict__ hist, const u32 ntiles, u32* __restrict__ totals) {
  __shared__ u32 s_wave[4];
  u32* row = hist + (size_t) blockIdx.x * ntiles;
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 carry = 0;
  for (u32 base = 0; base < ntiles; base += 256) {
    const u32 i = base + threadIdx.x;
    const u32 v = i < ntiles ? row[i] : 0u;
    u32 incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 o = __shfl_up(incl, off);
      if ((int) lane >= off) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    u32 woff = 0, all = 0;
    for (u32 w = 0; w < 4; w++) { if (w < wave) woff += s_wave[w]; all += s_wave[w]; }
    if (i < ntiles) row[i] = carry + woff + incl - v;
    carry += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <typename K, typename V = float>
__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const K* __restrict__ keys_in, const V* __restrict__ vals_in, K* __restrict__ keys_out,
                                                               V* __restrict__ vals_out, const u32 n, const int shift,
                                                               const u32* __restrict__ hist_scanned, const u32 ntiles,
                                                               const u32* __restrict__ totals) {
  constexpr int NW = kSortThreads / 64;
  __shared__ u32 s_dig[256];       // records of all smaller digits (exclusive scan of the digit totals)
  __shared__ u32 s_cnt[NW][256];   // records of each digit seen so far by each wave; afterwards: the wave's first output position per digit
  const u32 tile = blockIdx.x;
  const u32 wave = threadIdx.x >> 6, lane = sort_lane();
  for (int i = threadIdx.x; i < NW * 256; i += kSortThreads) (&s_cnt[0][0])[i] = 0;
  {  // exclusive scan of the 256 digit totals: thread d holds digit d
    const u32 v = totals[threadIdx.x];
    u32 incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 o = __shfl_up(incl, off);
      if ((int) lane >= off) incl += o;
    }
    s_dig[threadIdx.x] = incl - v;   // within the wave; the earlier waves' sums are added below
    __syncthreads();
    u32 before = 0;
    for (u32 w = 0; w < wave; w++) before += s_dig[w * 64 + 63] + totals[w * 64 + 63];
    __syncthreads();
    s_dig[threadIdx.x] += before;
  }
  __syncthreads();
  // wave w takes records [t0 + w * (tile / 4), + tile / 4) in rounds of 64 consecutive records
  const u32 w0 = tile * kSortTile + wave * (kSortTile / NW);
  K key[kSortRounds];
  V val[kSortRounds];
  u32 rank[kSortRounds];  // digit in the low 8 bits, rank among the wave's records of that digit above
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const u32 i = w0 + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : (K) 0;
    if (valid) val[r] = vals_in[i];
    const u32 d = (u32) (key[r] >> shift) & 255u;
    u64 peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (d >> b) & 1u;
      const u64 bal = __ballot(valid && bit);
      peers &= bit ? bal : ~bal;
    }
    u32 base = 0;
    if (valid) {
      const int leader = __ffsll((long long) peers) - 1;
      if ((int) lane == leader) {  // one lane per digit present in this round: its row of the table is this wave's alone
        base = s_cnt[wave][d];
        s_cnt[wave][d] = base + (u32) __popcll(peers);
      }
      base = __shfl(base, leader);
    }
    rank[r] = ((base + (u32) __popcll(peers & ((1ull << lane) - 1ull))) << 8) | d;
  }
  __syncthreads();
  {  // thread d: first output position of digit d for each wave = scanned histogram entry + the earlier waves' counts
    const u32 d = threadIdx.x;
    u32 run = s_dig[d] + hist_scanned[d * ntiles + tile];
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const u32 c = s_cnt[w][d];
      s_cnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const u32 i = w0 + r * 64 + lane;
    if (i < n) {
      const u32 pos = s_cnt[wave][rank[r] & 255u] + (rank[r] >> 8);
      keys_out[pos] = key[r];
      vals_out[pos] = val[r];
    }
  }
}

// Exclusive scan of n 64-bit words (the marks of a quad-tree, mrh_splat.h: two 32-bit counters in one word) in two launches of one
// workgroup per tile of 4 096 words: the tiles' sums, then every tile adds up the sums in front of it (a few dozen words) and scans
// itself through LDS (four neighbours a thread, loads and stores coalesced).  (A first version chained the tiles through one
// workgroup: 22 dependent round trips for a 640 x 480 image, 90 us — the seeding call went from 110 to 194 us.)
constexpr int kChainTile = 4096;
__device__ __forceinline__ u64 block_sum_u64(const u64 mine, u64* s_w) {  // 1024 threads; every thread gets the sum
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 v = mine;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();  // s_w may still be read from an earlier call
  if (lane == 0) s_w[wave] = v;
  __syncthreads();
  u64 all = 0;
  for (u32 w = 0; w < 16; w++) all += s_w[w];
  return all;
}
__global__ __launch_bounds__(1024) void k_tile_sums_u64(const u64* __restrict__ in, const u32 n, u64* __restrict__ sums) {
  __shared__ u64 s_w[16];
  const u32 base = blockIdx.x * kChainTile;
  u64 mine = 0;
#pragma unroll
  for (int k = 0; k < kChainTile / 1024; k++) {
    const u32 i = base + k * 1024 + threadIdx.x;
    mine += i < n ? in[i] : 0ull;
  }
  const u64 all = block_sum_u64(mine, s_w);
  if (threadIdx.x == 0) sums[blockIdx.x] = all;
}
__global__ __launch_bounds__(1024) void k_tile_scan_u64(const u64* __restrict__ in, const u32 n, const u64* __restrict__ sums, u64* __restrict__ out) {
  __shared__ u64 s_v[kChainTile];
  __shared__ u64 s_w[16];
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int PER = kChainTile / 1024;
  const u32 base = blockIdx.x * kChainTile, m = min((u32) kChainTile, n - base);
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const u32 i = k * 1024 + threadIdx.x;
    s_v[i] = i < m ? in[base + i] : 0ull;
  }
  u64 before = 0;
  for (u32 j = threadIdx.x; j < blockIdx.x; j += 1024) before += sums[j];
  const u64 carry = block_sum_u64(before, s_w);  // (its barriers also complete s_v)
  u64 v[PER], mine = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) { v[k] = s_v[threadIdx.x * PER + k]; mine += v[k]; }
  u64 incl = mine;
  for (int off = 1; off < 64; off <<= 1) {
    const u64 o = __shfl_up(incl, off);
    if ((int) lane >= off) incl += o;
  }
  __syncthreads();
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  u64 run = carry + (incl - mine);
  for (u32 w = 0; w < wave; w++) run += s_w[w];
#pragma unroll
  for (int k = 0; k < PER; k++) { s_v[threadIdx.x * PER + k] = run; run += v[k]; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const u32 i = k * 1024 + threadIdx.x;
    if (i < m) out[base + i] = s_v[i];
  }
}

}  // namespace mrh
