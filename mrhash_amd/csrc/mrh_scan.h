// mrh_scan.h — a LiDAR scan without a sort: the records of integrate3DKernel (vds.cu:1215-1379) go straight into per-voxel
// buckets, each bucket is put into point order inside LDS, and folded (combineVoxel, vhu.cuh:167-181) as mrh_lidar.h does.
//
// Why: the reference updates a voxel with a non-atomic read-modify-write per point; the deterministic restatement folds a
// voxel's records in ascending point index (D6).  mrh_lidar.h gets that order from a stable radix sort of ALL records of the
// scan by voxel id — nine launches, 66 of the 137 us of a 128 x 1024 scan, the record buffers written and read three times.
// But the order is only needed INSIDE a voxel, and a voxel's run is short (mean 6 records, a few hundred next to the sensor):
//
//   k_scan_walk     one beam per lane (walk_beam, the one copy of the arithmetic).  The records of the workgroup's 256 beams stay
//                   in LDS and are grouped by voxel in an LDS set: ONE global atomic per distinct voxel and workgroup adds the
//                   group's size to the voxel's counter (neighbouring beams hit the same voxels: the hot voxels next to the
//                   sensor take 10-30 x fewer atomics than records), marks the voxel's block as touched by this scan, and the
//                   records + groups are parked in a stash (one allocation per workgroup, order irrelevant).
//   k_scan_offsets  one wave per touched block: its 512 counters become offsets (block base from one atomic on the record
//                   cursor — the order of the blocks in the buffer does not matter), and its voxels are cut into chunks of
//                   bounded work for the last kernel (<= 4096 records and a bounded sum of squared run lengths; a run longer
//                   than kScanLongRun gets a chunk of its own).
//   k_scan_place    per walk workgroup: one atomic per group reserves the group's slots behind the voxel's offset, the records
//                   land there tagged with (point index, ordinal along the beam) — unordered inside the run.
//   k_scan_apply    per chunk: records -> LDS, every record finds its rank inside its run (counting the smaller tags: runs are
//                   short; a long run is sorted by a bitonic network instead, a run beyond the LDS by windows over the tag
//                   range), then one lane per voxel folds the ordered run exactly as k_points_apply does, and the counters are
//                   zero again for the next scan.
//
// Nothing is sorted globally, every record is written twice and read twice, and the host needs nothing back from the device
// before the last launch is enqueued (the sorted path waits for the record count to size its sort).
//
// Scratch: one u32 counter per voxel slot of the pool (+ 1/3 of the pool's bytes) and a stamp per block, allocated with the first
// scan; maps beyond 2^31 voxel ids, beams beyond kScanMaxSlots voxels and contexts where the scratch does not fit stay on
// the sorted path (mrh_lidar.h), which remains the cross-check of this one (MRH_LIDAR_BUCKETS=0).
#pragma once

#include "mrh_lidar.h"

namespace mrh {

constexpr u32 kScanSetSize = 4096;        // LDS set of a walk workgroup (distinct voxels of 256 beams: typically 600-1000)
constexpr int kScanSetProbe = 16;
constexpr int kScanMaxSlots = 32;         // records per beam this path accepts (LDS of the walk: slots * 2 KB + 32 KB)
constexpr u32 kScanChunkRecs = 2048;      // records an apply workgroup holds in LDS
constexpr u32 kScanChunkWeight = 1u << 15;  // work bound of a chunk: sum of cnt * max(cnt, 32) stays below twice this
constexpr u32 kScanLongRun = 128;         // a run longer than this is a chunk of its own (bitonic network / windows); its square must not exceed the weight
constexpr u32 kScanEmpty = 0xFFFFFFFFu;
constexpr u32 kScanCoarse = 0x80000000u;  // voxel id of a coarse unit
enum ScanCtr : int { SC_TOUCHED = 0, SC_PLACED = 1, SC_CHUNKS = 2, SC_N = 4 };

struct Scan {
  u32* vcnt;      // [pool blocks * 512] all zero between scans
  u32* bstamp;    // [pool blocks] 2 * sequence number of the last scan that touched the block + 1 if it holds coarse units
  u32* touched;   // [touched_cap] block | kScanCoarse
  u32* ctr;       // this scan's counters [SC_N]
  u32* ctr_next;  // the next scan's (zeroed by k_scan_collect)
  uint2* st_meta; // stash of walk workgroup w at w * 256 * slots: {group | rank in group << 13, lane << 5 | ordinal along the beam}
  float* st_sdf;
  uint4* st_grp;  // stash: {voxel id, records, records of the voxel that arrived before this group, 0}
  uint2* wgdesc;  // per walk workgroup {records, groups}
  u32* rp;        // placed records: point index << ord_shift | ordinal
  float* rs;
  uint4* chunks;  // {block | coarse, v0 | v1 << 16, r0, r1}
  u32 rec_cap, chunk_cap, touched_cap, seq;
  int ord_shift;  // 5 on variance-adaptive maps (a beam can cross several fine cells of one coarse voxel), else 0
};

// Same-address atomics with a return value cost ~15 ns each on this chip whoever issues them (measured: 3 000 appends to one
// list counter = 45 us of a kernel), so nothing here takes one per block or per workgroup: the stash is addressed by workgroup,
// the touched blocks are found by their stamps, block bases and chunk slots are reserved 16 blocks at a time.

__global__ __launch_bounds__(256) void k_scan_walk(const Cam c, const Map m, const Tab t, const float* __restrict__ pts,
                                                   const float* __restrict__ normals, const u32 n, const Scan sc, const int slots) {
  extern __shared__ u32 s_dyn[];
  u32* s_id = s_dyn;                                  // [slots][256] voxel id; after the grouping: set slot | rank << 12
  float* s_sdf = (float*) (s_dyn + slots * 256);      // [slots][256]
  u32* s_key = s_dyn + 2 * slots * 256;               // [kScanSetSize]
  u32* s_cnt = s_key + kScanSetSize;                  // [kScanSetSize] records of the group; after phase B: the group's index
  __shared__ u32 s_part[4];
  __shared__ u32 s_ng;
  const u32 tid = threadIdx.x;
  for (u32 i = tid; i < kScanSetSize; i += 256) { s_key[i] = kScanEmpty; s_cnt[i] = 0; }
  if (tid == 0) s_ng = 0;
  const u32 i = blockIdx.x * 256 + tid;
  u32 cnt = 0;
  bool over = false;
  walk_beam(c, m, t, pts, normals, i, i < n, [&](const u32 val, const int res, const u32 li, const float sdf) {
    u32 id;
    if (res) { const u32 u = val & ~kValCoarseBit; id = ((u >> 3) * 512u + (u & 7u) * 64u + li) | kScanCoarse; }
    else id = val * 512u + li;
    if ((int) cnt < slots) { s_id[cnt * 256 + tid] = id; s_sdf[cnt * 256 + tid] = sdf; cnt++; }
    else over = true;
  });
  if (over) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_SCAN);  // the host's bound on the voxels of a beam did not hold: the call fails
  // this beam's place among the workgroup's records
  u32 incl = cnt;
  const u32 lane = tid & 63, wave = tid >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const u32 o = __shfl_up(incl, off);
    if ((int) lane >= off) incl += o;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();  // the set is initialised, the wave totals are there
  u32 lane_off = incl - cnt, R = 0;
  for (u32 w = 0; w < 4; w++) { if (w < wave) lane_off += s_part[w]; R += s_part[w]; }
  // A: group by voxel
  u32 ovf = 0;
  for (u32 j = 0; j < cnt; j++) {
    const u32 key = s_id[j * 256 + tid];
    u32 h = (key * 0x9E3779B1u) >> 20;
    bool ok = false;
#pragma unroll 1
    for (int p = 0; p < kScanSetProbe; p++) {
      const u32 old = atomicCAS(&s_key[h], kScanEmpty, key);
      if (old == kScanEmpty || old == key) { ok = true; break; }
      h = (h + 1) & (kScanSetSize - 1);
    }
    if (ok) s_id[j * 256 + tid] = h | (atomicAdd(&s_cnt[h], 1u) << 12);
    else ovf |= 1u << j;  // set saturated: the record becomes a group of its own
  }
  __syncthreads();
  const u32 off = blockIdx.x * 256u * (u32) slots;  // this workgroup's part of the stash (the host sizes it by the bound: points * slots)
  // B: one group per occupied set slot.  The atomic that adds the group to its voxel's counter returns what was there: the group's
  // place inside the voxel's run (arrival order — the run is put into point order later, in LDS).
  {
    constexpr int PER = kScanSetSize / 256;
    u32 key[PER], prev[PER], cv[PER], g[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const u32 sl = tid + 256u * k;
      key[k] = s_key[sl];
      if (key[k] != kScanEmpty) {
        cv[k] = s_cnt[sl];
        g[k] = atomicAdd(&s_ng, 1u);
        s_cnt[sl] = g[k];
        prev[k] = atomicAdd(&sc.vcnt[key[k] & ~kScanCoarse], cv[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < PER; k++)
      if (key[k] != kScanEmpty) {
        sc.st_grp[off + g[k]] = make_uint4(key[k], cv[k], prev[k], 0u);
        const u32 H = (key[k] & ~kScanCoarse) >> 9, stamp = sc.seq * 2u + (key[k] >> 31);
        if (sc.bstamp[H] != stamp) sc.bstamp[H] = stamp;  // every writer stores the same value
      }
  }
  __syncthreads();
  // C: the records
  for (u32 j = 0; j < cnt; j++) {
    const u32 v = s_id[j * 256 + tid];
    u32 meta;
    if ((ovf >> j) & 1u) {
      const u32 g = atomicAdd(&s_ng, 1u);
      const u32 prev = atomicAdd(&sc.vcnt[v & ~kScanCoarse], 1u);
      sc.st_grp[off + g] = make_uint4(v, 1u, prev, 0u);
      sc.bstamp[(v & ~kScanCoarse) >> 9] = sc.seq * 2u + (v >> 31);
      meta = g;
    } else {
      meta = s_cnt[v & 0xFFFu] | ((v >> 12) << 13);
    }
    sc.st_meta[off + lane_off + j] = make_uint2(meta, (tid << 5) | j);
    sc.st_sdf[off + lane_off + j] = s_sdf[j * 256 + tid];
  }
  __syncthreads();
  if (tid == 0) sc.wgdesc[blockIdx.x] = make_uint2(R, s_ng);
}

// The blocks this scan touched, from their stamps (no list is kept while the beams walk): 1024 blocks per workgroup, one
// append per workgroup that found any.
__global__ __launch_bounds__(1024) void k_scan_collect(const Tab t, const Scan sc, const u32 n_blocks_or_0) {
  __shared__ u32 s_w[16];
  __shared__ u32 s_base;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x == 0 && tid < (u32) SC_N) sc.ctr_next[tid] = 0;
  const u32 nb = n_blocks_or_0 ? n_blocks_or_0 : (u32) t.ctr[CTR_HWM_FINE];
  for (u32 h0 = blockIdx.x * 1024u; h0 < nb; h0 += gridDim.x * 1024u) {
    const u32 H = h0 + tid;
    const u32 st = H < nb ? sc.bstamp[H] : 0u;
    const bool hit = (st >> 1) == sc.seq;
    const u64 bal = __ballot(hit);
    if (lane == 0) s_w[wave] = (u32) __popcll(bal);
    __syncthreads();
    u32 before = 0, all = 0;
    for (u32 w = 0; w < 16; w++) { if (w < wave) before += s_w[w]; all += s_w[w]; }
    if (tid == 0 && all) s_base = atomicAdd(&sc.ctr[SC_TOUCHED], all);
    __syncthreads();
    if (hit) {
      const u32 ti = s_base + before + (u32) __popcll(bal & lanemask_lt());
      if (ti < sc.touched_cap) sc.touched[ti] = H | ((st & 1u) << 31);
    }
    __syncthreads();
  }
}

// One wave per touched block, 16 blocks per workgroup: counters -> offsets, chunks for k_scan_apply.  The 16 blocks reserve their
// records and their chunk slots with ONE atomic each per workgroup.
__global__ __launch_bounds__(1024) void k_scan_offsets(const Tab t, const Scan sc) {
  __shared__ u32 s_excl[16][513];
  __shared__ unsigned short s_start[16][514];
  __shared__ u32 s_tot[16], s_nch[16], s_rbase, s_cbase;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 nt = min(sc.ctr[SC_TOUCHED], sc.touched_cap);
  for (u32 t0 = blockIdx.x * 16u; t0 < nt; t0 += gridDim.x * 16u) {
    const u32 ti = t0 + wave;
    const bool have = ti < nt;
    const u32 Hc = have ? sc.touched[ti] : 0u, H = Hc & ~kScanCoarse;
    uint4* p = (uint4*) (sc.vcnt + (size_t) H * 512 + lane * 8);
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (have) { a = p[0]; b = p[1]; }
    const u32 cv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32 w[8], sumc = 0, sumw = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32 cc = min(cv[k], kScanChunkRecs);
      w[k] = min(cc * max(cc, 32u), kScanChunkWeight);
      sumc += cv[k];
      sumw += w[k];
    }
    u32 ic = sumc, iw = sumw;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 oc = __shfl_up(ic, off), ow = __shfl_up(iw, off);
      if ((int) lane >= off) { ic += oc; iw += ow; }
    }
    const u32 total = __shfl(ic, 63);  // 0: a block looked up by a beam that left no record in it (or no block for this wave)
    const u32 pw7 = __shfl_up(w[7], 1), pc7 = __shfl_up(cv[7], 1);
    u32 run_c = ic - sumc, run_w = iw - sumw;
    u32 pq = lane ? (run_w - pw7) / kScanChunkWeight : 0u;
    bool plong = lane ? pc7 > kScanLongRun : false;
    u32 st[8], flags = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32 v = lane * 8 + k, q = run_w / kScanChunkWeight;
      const bool lng = cv[k] > kScanLongRun;
      if (v == 0 || q != pq || lng || plong) flags |= 1u << k;
      s_excl[wave][v] = run_c;
      st[k] = run_c;
      run_c += cv[k];
      run_w += w[k];
      pq = q;
      plong = lng;
    }
    if (lane == 63) s_excl[wave][512] = total;
    const u32 nf = (u32) __popc(flags);
    u32 inf = nf;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 o = __shfl_up(inf, off);
      if ((int) lane >= off) inf += o;
    }
    const u32 nch_all = __shfl(inf, 63), nch = total ? nch_all : 0u;
    u32 idx = inf - nf;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if ((flags >> k) & 1u) s_start[wave][idx++] = (unsigned short) (lane * 8 + k);
    if (lane == 0) { s_start[wave][nch_all] = 512; s_tot[wave] = total; s_nch[wave] = nch; }
    __syncthreads();
    if (tid == 0) {
      u32 rt = 0, ct = 0;
      for (int k = 0; k < 16; k++) { rt += s_tot[k]; ct += s_nch[k]; }
      s_rbase = rt ? atomicAdd(&sc.ctr[SC_PLACED], rt) : 0u;
      s_cbase = ct ? atomicAdd(&sc.ctr[SC_CHUNKS], ct) : 0u;
    }
    __syncthreads();
    if (total) {
      u32 base = s_rbase, cbase = s_cbase;
      for (u32 k = 0; k < wave; k++) { base += s_tot[k]; cbase += s_nch[k]; }
      p[0] = make_uint4(base + st[0], base + st[1], base + st[2], base + st[3]);
      p[1] = make_uint4(base + st[4], base + st[5], base + st[6], base + st[7]);
      if (cbase + nch > sc.chunk_cap) {
        if (lane == 0) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_SCAN);
      } else {
        for (u32 k = lane; k < nch; k += 64) {
          const u32 v0 = s_start[wave][k], v1 = s_start[wave][k + 1];
          sc.chunks[cbase + k] = make_uint4(Hc, v0 | (v1 << 16), base + s_excl[wave][v0], base + s_excl[wave][v1]);
        }
      }
    }
    __syncthreads();
  }
}

// Records of one walk workgroup -> their voxels' runs: slot = start of the run (k_scan_offsets) + the records that had arrived
// before the group (k_scan_walk's atomic) + rank inside the group.  No atomics.
__global__ __launch_bounds__(256) void k_scan_place(const Scan sc, const int slots) {
  __shared__ u32 s_base[256 * kScanMaxSlots];
  const uint2 d = sc.wgdesc[blockIdx.x];
  const u32 off = blockIdx.x * 256u * (u32) slots, R = d.x, G = d.y;
  for (u32 g = threadIdx.x; g < G; g += 256) {
    const uint4 grp = sc.st_grp[off + g];
    s_base[g] = sc.vcnt[grp.x & ~kScanCoarse] + grp.z;
  }
  __syncthreads();
  for (u32 r = threadIdx.x; r < R; r += 256) {
    const uint2 meta = sc.st_meta[off + r];
    const float sdf = sc.st_sdf[off + r];
    const u32 slot = s_base[meta.x & 0x1FFFu] + (meta.x >> 13);
    if (slot < sc.rec_cap) {
      const u32 pidx = blockIdx.x * 256 + (meta.y >> 5);
      sc.rp[slot] = sc.ord_shift ? (pidx << 5) | (meta.y & 31u) : pidx;
      sc.rs[slot] = sdf;
    }
  }
}

// The fold of one voxel's run, record by record in the order they are pushed: k_points_apply's chain (see there for why each
// piece sits where it does), as a state that survives between the windows of a run beyond the LDS.
struct VoxFold {
  float *p_sdf, *p_ss;
  u32* p_rgbw;
  float s0, w0f, den, rden, w1f, pend, half_vs;
  u32 w0, wsum, w1, wmax, rgb0, nfold;
  bool two;
  __device__ __forceinline__ void begin(const Map& m, const Tab& t, const u32 H, const bool coarse, const u32 vi) {
    char* base = t.pool + (size_t) H * kFineBytes;
    if (coarse) {
      base += (size_t) (vi >> 6) * kCoarseBytes;
      const u32 li = vi & 63u;
      p_sdf = (float*) base + li; p_ss = (float*) (base + 256) + li; p_rgbw = (u32*) (base + 512) + li;
    } else {
      p_sdf = (float*) base + vi; p_ss = (float*) (base + 2048) + vi; p_rgbw = (u32*) (base + 4096) + vi;
    }
    s0 = *p_sdf;
    rgb0 = *p_rgbw;
    w0 = rgb0 >> 24;
    w1 = (u32) (m.weight_sample & 0xFF); wmax = (u32) (m.weight_max & 0xFF);
    w1f = (float) w1;
    wsum = w0 + w1;
    w0f = (float) w0; den = (float) (int) wsum; rden = rcp_refined(den);
    half_vs = m.vs / 2;
    two = m.wsum_two_steps != 0;
    nfold = 0;
    pend = 0.f;
  }
  __device__ __forceinline__ void step(const float sdf) {
    const float num = s0 * w0f + sdf * w1f;
    const u32 w0n = wsum < wmax ? wsum : wmax, wsum_n = w0n + w1;
    const float w0f_n = (float) w0n, den_n = (float) (int) wsum_n, rden_n = rcp_refined(den_n);
    s0 = two ? num / den : div_cr(num, den, rden);
    w0 = w0n; wsum = wsum_n; w0f = w0f_n; den = den_n; rden = rden_n;
  }
  __device__ __forceinline__ void push(const float sdf) {  // a record is folded once its successor is known to belong to the run
    if (nfold) step(pend);
    pend = sdf;
    nfold++;
  }
  __device__ __forceinline__ void run(const float* vals, const u32 n) {  // a whole run from LDS
    pend = vals[0];
    u32 i = 1;
    for (; i + 4 <= n; i += 4) {
      const float v0 = vals[i], v1 = vals[i + 1], v2 = vals[i + 2], v3 = vals[i + 3];
      step(pend); step(v0); step(v1); step(v2);
      pend = v3;
    }
    for (; i < n; i++) { step(pend); pend = vals[i]; }
    nfold = n;
  }
  __device__ __forceinline__ void end() {
    const float s_prev = s0, sdf_last = pend;  // the variance term of the LAST update (vds.cu:1352-1366) needs the state before it
    const u32 w_prev = w0;
    step(pend);
    u32 r0 = rgb0 & 0xFF, g0 = (rgb0 >> 8) & 0xFF, b0 = (rgb0 >> 16) & 0xFF;
    const u32 sh = nfold < 8u ? nfold : 8u, add = (1u << sh) - 1u;  // (c + 1) >> 1 per record, n times = ceil(c / 2^n)
    r0 = (r0 + add) >> sh; g0 = (g0 + add) >> sh; b0 = (b0 + add) >> sh;
    const float curr_mean = w_prev > 0 ? s_prev : 0.f;
    const float delta = (sdf_last - curr_mean) / half_vs;
    const float delta2 = (sdf_last - s0) / half_vs;
    *p_sdf = s0;
    *p_ss = 0.f + delta * delta2;
    *p_rgbw = r0 | (g0 << 8) | (b0 << 16) | (w0 << 24);
  }
};

__global__ __launch_bounds__(256) void k_scan_apply(const Map m, const Tab t, const Scan sc, const u32 tag_space, const int count_updates) {
  constexpr int PER = kScanChunkRecs / 256;
  __shared__ u32 s_tag[kScanChunkRecs];
  __shared__ float s_sdf[kScanChunkRecs];
  __shared__ float s_sorted[kScanChunkRecs];
  __shared__ u32 s_end[513];          // s_end[u] = first record of run u, s_end[u + 1] = one past its last
  __shared__ unsigned short s_order[512];
  __shared__ u32 s_bcnt[16];
  __shared__ u32 s_part[4];
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 nch = min(sc.ctr[SC_CHUNKS], sc.chunk_cap);
  u32 updated = 0;
  for (u32 ci = blockIdx.x; ci < nch; ci += gridDim.x) {
    const uint4 ch = sc.chunks[ci];
    const u32 H = ch.x & ~kScanCoarse, v0 = ch.y & 0xFFFFu, v1 = ch.y >> 16, r0 = ch.z, nr = ch.w - ch.z, nv = v1 - v0;
    const bool coarse = (ch.x & kScanCoarse) != 0;
    for (u32 u = tid; u < nv; u += 256) {  // where the runs start; the counters are zero again for the next scan
      u32* p = sc.vcnt + (size_t) H * 512 + v0 + u;
      s_end[u] = *p - r0;
      *p = 0;
    }
    if (tid == 0) s_end[nv] = nr;
    if (tid < 16) s_bcnt[tid] = 0;
    if (nr == 0) { __syncthreads(); continue; }
    if (nv == 1 && nr > kScanLongRun) {
      VoxFold f;
      if (tid == 0) f.begin(m, t, H, coarse, v0);
      if (nr <= kScanChunkRecs) {  // one long run: bitonic network over the tags
        u32 N = 256;
        while (N < nr) N <<= 1;
        for (u32 q = tid; q < N; q += 256) {
          s_tag[q] = q < nr ? sc.rp[r0 + q] : kScanEmpty;
          s_sdf[q] = q < nr ? sc.rs[r0 + q] : 0.f;
        }
        __syncthreads();
        for (u32 k = 2; k <= N; k <<= 1) {
          for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 i = tid; i < N / 2; i += 256) {
              const u32 lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
              const bool asc = (lo & k) == 0;
              const u32 a = s_tag[lo], b = s_tag[hi];
              if ((a > b) == asc) {
                s_tag[lo] = b; s_tag[hi] = a;
                const float x = s_sdf[lo];
                s_sdf[lo] = s_sdf[hi]; s_sdf[hi] = x;
              }
            }
            __syncthreads();
          }
        }
        if (tid == 0) f.run(s_sdf, nr);
      } else {  // beyond the LDS: windows over the tag range, every tag is unique inside a voxel -> direct addressing
        for (u32 w0 = 0; w0 < tag_space; w0 += kScanChunkRecs) {
          for (u32 q = tid; q < kScanChunkRecs; q += 256) s_tag[q] = 0;
          __syncthreads();
          for (u32 q = tid; q < nr; q += 256) {
            const u32 d = sc.rp[r0 + q] - w0;
            if (d < kScanChunkRecs) { s_sdf[d] = sc.rs[r0 + q]; s_tag[d] = 1; }
          }
          __syncthreads();
          // in-order compaction: thread t owns slots [PER t, PER t + PER)
          u32 mine = 0;
#pragma unroll
          for (int k = 0; k < PER; k++) mine += s_tag[tid * PER + k];
          u32 incl = mine;
          for (int off = 1; off < 64; off <<= 1) {
            const u32 o = __shfl_up(incl, off);
            if ((int) lane >= off) incl += o;
          }
          if (lane == 63) s_part[wave] = incl;
          __syncthreads();
          u32 pos = incl - mine, tot = 0;
          for (u32 w = 0; w < 4; w++) { if (w < wave) pos += s_part[w]; tot += s_part[w]; }
#pragma unroll
          for (int k = 0; k < PER; k++)
            if (s_tag[tid * PER + k]) s_sorted[pos++] = s_sdf[tid * PER + k];
          __syncthreads();
          if (tid == 0)
            for (u32 q = 0; q < tot; q++) f.push(s_sorted[q]);
          __syncthreads();
        }
      }
      if (tid == 0) { f.end(); updated++; }
      __syncthreads();
      continue;
    }
    for (u32 q = tid; q < nr; q += 256) { s_tag[q] = sc.rp[r0 + q]; s_sdf[q] = sc.rs[r0 + q]; }
    __syncthreads();
    for (u32 q = tid; q < nr; q += 256) {  // rank of record q inside its run
      u32 lo = 0, hi = nv;  // largest u with s_end[u] <= q (empty runs share a start: the last of them is the run that holds q)
      while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (s_end[mid] <= q) lo = mid; else hi = mid;
      }
      const u32 a = s_end[lo], b = s_end[lo + 1];
      u32 rank = 0;
      if (b - a > 1) {
        const u32 tag = s_tag[q];
        for (u32 j = a; j < b; j++) rank += s_tag[j] < tag ? 1u : 0u;
      }
      s_sorted[a + rank] = s_sdf[q];
    }
    // the voxels in order of falling run length (by power of two): the lanes of a wave then fold runs of similar length, and a
    // wave is as slow as its longest run
    u32 myb[2], mypos[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const u32 u = tid + 256u * r;
      myb[r] = 16;
      if (u < nv) {
        const u32 len = s_end[u + 1] - s_end[u];
        if (len) { myb[r] = 31u - (u32) __clz(len); mypos[r] = atomicAdd(&s_bcnt[myb[r]], 1u); }
      }
    }
    __syncthreads();
    u32 nnz = 0;
    {
      u32 boff[16], run = 0;
#pragma unroll
      for (int b = 15; b >= 0; b--) { boff[b] = run; run += s_bcnt[b]; }
      nnz = run;
#pragma unroll
      for (int r = 0; r < 2; r++)
        if (myb[r] < 16) {
          u32 o = 0;
#pragma unroll
          for (int b = 0; b < 16; b++) o = myb[r] == (u32) b ? boff[b] : o;
          s_order[o + mypos[r]] = (unsigned short) (tid + 256u * r);
        }
    }
    __syncthreads();
    for (u32 i = tid; i < nnz; i += 256) {
      const u32 u = s_order[i];
      const u32 a = s_end[u], b = s_end[u + 1];
      VoxFold f;
      f.begin(m, t, H, coarse, v0 + u);
      f.run(s_sorted + a, b - a);
      f.end();
      updated++;
    }
    __syncthreads();
  }
  if (count_updates && updated) atomicAdd(&t.prof[PROF_UPDATED], (u64) updated);  // profile mode: voxels this scan updated
}

}  // namespace mrh
