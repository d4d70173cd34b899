// mrh_sort.h — stable LSD radix sort of the (voxel id, sdf) records of one LiDAR scan (mrh_lidar.h).
//
// Why not rocPRIM here: a scan has ~10^5 .. 10^6 records and ~21 significant key bits.  rocPRIM's onesweep sort is built for
// 10^7+ items: at this size each of its three passes is a latency-bound chain (decoupled look-back over ~300 tiles, 26 us
// measured) and every call adds 5-7 hipMemsetAsync of look-back state (5 us each): 130 us of a 200 us scan
// (profiles/r03/lidar_kernel_stats_rocprim_sort.csv).  At this size the classic three-kernel pass is the better fit:
//   k_sort_hist      per tile (4096 records, one workgroup): 256-bin histogram of the pass's digit -> hist[digit][tile]
//   k_sort_scan      exclusive scan of hist in (digit, tile) order — one workgroup, the table has 256 x ~160 entries
//   k_sort_scatter   per tile: every record's rank among the records of its digit that precede it — inside its wave by a
//                    ballot match (8 ballots give the lanes holding the same digit; rank = popcount of the lower ones), across
//                    the rounds of a wave and the waves of a tile by per-wave digit counters in LDS — plus the scanned
//                    histogram entry gives its final position.  Order of equal digits = tile, wave, round, lane = input
//                    order: the pass is STABLE, which is what keeps a voxel's records in ascending point index (D6).
// No atomics on global memory, no look-back, nothing to clear between calls except the LDS the kernels own.
#pragma once

#include "mrh_device.h"

namespace mrh {

constexpr int kSortTile = 4096;     // records per workgroup
constexpr int kSortThreads = 256;   // 4 waves x 16 rounds x 64 lanes
constexpr int kSortRounds = kSortTile / kSortThreads;
constexpr u32 kSortScanMax = 1u << 17;  // histogram entries the one-workgroup scan takes (512 tiles = 2 M records); beyond: rocPRIM

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

// exclusive scan of hist[0 .. total) in place, one workgroup of 1024 threads (total <= kSortScanMax)
__global__ __launch_bounds__(1024) void k_sort_scan(u32* __restrict__ hist, const u32 total) {
  __shared__ u32 s_part[1024];
  const u32 per = (total + 1023) / 1024;
  const u32 lo = threadIdx.x * per, hi = min(total, lo + per);
  u32 sum = 0;
  for (u32 i = lo; i < hi; i++) sum += hist[i];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  // Hillis-Steele over the 1024 partial sums
  for (u32 off = 1; off < 1024; off <<= 1) {
    const u32 v = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  u32 run = s_part[threadIdx.x] - sum;  // exclusive prefix of this thread's segment
  for (u32 i = lo; i < hi; i++) {
    const u32 v = hist[i];
    hist[i] = run;
    run += v;
  }
}

template <typename K>
__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const K* __restrict__ keys_in, const float* __restrict__ vals_in, K* __restrict__ keys_out,
                                                               float* __restrict__ vals_out, const u32 n, const int shift,
                                                               const u32* __restrict__ hist_scanned, const u32 ntiles) {
  constexpr int NW = kSortThreads / 64;
  __shared__ u32 s_cnt[NW][256];   // records of each digit seen so far by each wave; afterwards: the wave's first output position per digit
  const u32 tile = blockIdx.x;
  const u32 wave = threadIdx.x >> 6, lane = sort_lane();
  for (int i = threadIdx.x; i < NW * 256; i += kSortThreads) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  // wave w takes records [t0 + w * 1024, + 1024) in 16 rounds of 64 consecutive records
  const u32 w0 = tile * kSortTile + wave * (kSortTile / NW);
  K key[kSortRounds];
  float val[kSortRounds];
  u32 rank[kSortRounds];  // digit in the low 8 bits, rank among the wave's records of that digit above
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const u32 i = w0 + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i] : (K) 0;
    val[r] = valid ? vals_in[i] : 0.f;
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
    u32 run = hist_scanned[d * ntiles + tile];
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

}  // namespace mrh
