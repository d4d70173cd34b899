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
//   k_scan_offsets  the blocks this scan touched, from their stamps (windows of kScanWindow = 16 blocks per workgroup); one wave per touched
//                   block: its 512 counters become offsets (the bases of 16 blocks from one atomic on the
//                   record cursor — the order of the blocks in the buffer does not matter), and its voxels are cut into chunks
//                   of bounded work for the last kernel (< 512 records and a bounded sum of squared run lengths; a run longer
//                   than kScanLongRun gets a chunk of its own).
//   k_scan_place    per walk workgroup: the records land in their voxel's run — start of the run + what the walk's atomic
//                   returned + rank in the group — tagged with (point index, ordinal along the beam); unordered inside the run.
//   k_scan_apply    per chunk, ONE WAVE: records -> LDS, every record finds its rank inside its run (counting the smaller tags:
//                   runs are short; a long run is sorted by a bitonic network instead, a run beyond the LDS by windows over
//                   the tag range), then one lane per voxel folds the ordered run exactly as k_points_apply does — the voxels
//                   dealt to the lanes in order of falling run length —, and the counters are zero again for the next scan.
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

#ifndef MRH_SCAN_SET
#define MRH_SCAN_SET 4096
#endif
constexpr u32 kScanSetSize = MRH_SCAN_SET;        // LDS set of a walk workgroup (distinct voxels of 256 beams: typically 600-1000)
constexpr int kScanSetProbe = 16;
constexpr int kScanSetShift = kScanSetSize == 4096 ? 20 : kScanSetSize == 2048 ? 21 : 22;  // 32 - log2(size): the hash's top bits
static_assert(kScanSetSize >= 1024 && kScanSetSize <= 4096 && (kScanSetSize & (kScanSetSize - 1)) == 0,
              "MRH_SCAN_SET: 1024, 2048 or 4096 (a set slot is kept in 12 bits; the shift above knows these three sizes)");
constexpr int kScanMaxSlots = 32;         // records per beam this path accepts (LDS of the walk: slots * 2 KB + 32 KB)
constexpr u32 kScanWaveRecs = 512;        // records of a chunk: one WAVE of k_scan_apply holds them in its slice of the LDS
constexpr u32 kScanChunkWeight = 1u << 13;  // work bound of a chunk: sum of cnt * max(cnt, 32) stays below twice this (so: < 512 records)
constexpr u32 kScanLongRun = 64;          // a run longer than this is a chunk of its own (bitonic network); its square must not exceed the weight
constexpr u32 kScanBigRecs = 2048;        // a run beyond a wave's slice is sorted by a whole workgroup (<= this: bitonic; beyond: windows)
constexpr u32 kScanEmpty = 0xFFFFFFFFu;
constexpr u32 kScanCoarse = 0x80000000u;  // voxel id of a coarse unit
enum ScanCtr : int { SC_PLACED = 0, SC_CHUNKS = 1, SC_BIG = 2, SC_UNUSED = 3, SC_N = 4 };  // PLACED | CHUNKS: one 64-bit word (k_scan_offsets reserves both with ONE atomic)

struct Scan {
  u32* vcnt;      // [pool blocks * 512] all zero between scans
  u32* bstamp;    // [pool blocks] 2 * sequence number of the last scan that touched the block + 1 if it holds coarse units
  u32* ctr;       // this scan's counters [SC_N]
  u32* ctr_next;  // the next scan's (zeroed by k_scan_offsets)
  uint2* st_meta; // stash of walk workgroup w at w * 256 * slots: {group | rank in group << 13, lane << 5 | ordinal along the beam}; on fine
                  // maps (ord_shift 0) ONE word a record: group | rank << 13 | lane << 21
  float* st_sdf;
  uint2* st_grp;  // stash: {voxel id, records of the voxel that arrived before this group}
  uint2* wgdesc;  // per walk workgroup {records, groups}
  uint4* rec;     // placed records: {point index << ord_shift | ordinal, sdf, the voxel's index in its block, 0} — or, when the tag
                  // leaves room for the voxel's 9 bits (narrow: tags below 2^23, every scan up to 8 M points; 256 k on variance-adaptive
                  // maps), 8 bytes: {tag << 9 | index, sdf}: the records cross HBM twice, as k_scan_place's output and k_scan_apply's input
  uint4* chunks;  // {block | coarse, v0 | v1 << 16, r0, r1}; runs beyond kScanWaveRecs from the END of the array downwards
  u32 rec_cap, chunk_cap, seq;
  int ord_shift;  // 5 on variance-adaptive maps (a beam can cross several fine cells of one coarse voxel), else 0
  int narrow;     // 8-byte records (see rec)
  BeamOrder order;  // which 256 beams a walk workgroup takes (mrh_lidar.h)
};

// Same-address atomics with a return value cost ~15 ns each on this chip whoever issues them (measured: 3 000 appends to one
// list counter = 45 us of a kernel), so nothing here takes one per block or per workgroup: the stash is addressed by workgroup,
// the touched blocks are found by their stamps, block bases and chunk slots are reserved 16 blocks at a time.

__device__ __forceinline__ uint4 scan_load_rec(const Scan& sc, const u32 i) {  // -> {tag, sdf bits, voxel index in the block, 0}
  if (sc.narrow) {
    const uint2 v = ((const uint2*) sc.rec)[i];
    return make_uint4(v.x >> 9, v.y, v.x & 511u, 0u);
  }
  return sc.rec[i];
}
__device__ __forceinline__ void scan_store_rec(const Scan& sc, const u32 i, const u32 tag, const float sdf, const u32 li) {
  if (sc.narrow) ((uint2*) sc.rec)[i] = make_uint2((tag << 9) | li, __float_as_uint(sdf));
  else sc.rec[i] = make_uint4(tag, __float_as_uint(sdf), li, 0u);
}

#ifdef MRH_SCAN_TRACE
// tuning builds only (tools/trace_scan.sh): wall-clock stamps (100 MHz) of thread 0 of every workgroup at its phase boundaries;
// [kernel][workgroup][slot], kernels: 0 walk, 1 offsets, 2 place, 3 apply
constexpr int kScanTraceWgs = 4096;
__device__ unsigned long long d_scan_trace[4][kScanTraceWgs][8];
#define MRH_SC_TS(kern, slot) do { if (threadIdx.x == 0 && blockIdx.x < kScanTraceWgs) d_scan_trace[kern][blockIdx.x][slot] = wall_clock64(); } while (0)
// per wave: lane 0 of every wave, the stamp's low 44 bits + 20 bits of what the wave worked on
#define MRH_SC_TSW(kern, slot, info) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < kScanTraceWgs) d_scan_trace[kern][blockIdx.x][slot] = (wall_clock64() & ((1ull << 44) - 1)) | ((unsigned long long) (info) << 44); } while (0)
#else
#define MRH_SC_TS(kern, slot) do { } while (0)
#define MRH_SC_TSW(kern, slot, info) do { } while (0)
#endif

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
  MRH_SC_TS(0, 0);
  for (u32 i = tid; i < kScanSetSize; i += 256) { s_key[i] = kScanEmpty; s_cnt[i] = 0; }
  if (tid == 0) s_ng = 0;
  const u32 i = sc.order.point(blockIdx.x, tid);
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
  MRH_SC_TS(0, 1);
  // this beam's place among the workgroup's records
  u32 incl = cnt;
  const u32 lane = tid & 63, wave = tid >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const u32 o = __shfl_up(incl, off);
    if ((int) lane >= off) incl += o;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();  // the set is initialised, the wave totals are there
  MRH_SC_TS(0, 2);
  u32 lane_off = incl - cnt, R = 0;
  for (u32 w = 0; w < 4; w++) { if (w < wave) lane_off += s_part[w]; R += s_part[w]; }
  // A: group by voxel
  u32 ovf = 0;
  for (u32 j = 0; j < cnt; j++) {
    const u32 key = s_id[j * 256 + tid];
    u32 h = (key * 0x9E3779B1u) >> kScanSetShift;
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
  MRH_SC_TS(0, 3);
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
        sc.st_grp[off + g[k]] = make_uint2(key[k], prev[k]);
        const u32 H = (key[k] & ~kScanCoarse) >> 9, stamp = sc.seq * 2u + (key[k] >> 31);
        sc.bstamp[H] = stamp;  // every writer stores the same value (a test first would cost a dependent load per group)
      }
  }
  __syncthreads();
  MRH_SC_TS(0, 4);
  // C: the records, each wave's in its own part of the workgroup's stash, ordinal-major (the lanes that have a j-th record
  // store it side by side: coalesced)
  {
    u32 pos = lane_off - (incl - cnt);  // records of the earlier waves
    for (u32 j = 0;; j++) {
      const bool has = j < cnt;
      const u64 bal = __ballot(has);
      if (!bal) break;
      if (has) {
        const u32 v = s_id[j * 256 + tid];
        u32 meta;
        if ((ovf >> j) & 1u) {
          const u32 g = atomicAdd(&s_ng, 1u);
          const u32 prev = atomicAdd(&sc.vcnt[v & ~kScanCoarse], 1u);
          sc.st_grp[off + g] = make_uint2(v, prev);
          sc.bstamp[(v & ~kScanCoarse) >> 9] = sc.seq * 2u + (v >> 31);
          meta = g;
        } else {
          meta = s_cnt[v & 0xFFFu] | ((v >> 12) << 13);  // v = set slot (< 4096) | rank << 12
        }
        const u32 at = off + pos + (u32) __popcll(bal & lanemask_lt());
        if (sc.ord_shift) sc.st_meta[at] = make_uint2(meta, (tid << 5) | j);
        else ((u32*) sc.st_meta)[at] = meta | (tid << 21);  // fine maps: a beam meets a voxel once — the rank fits 8 bits, the ordinal is not needed
        sc.st_sdf[at] = s_sdf[j * 256 + tid];
      }
      pos += (u32) __popcll(bal);
    }
  }
  __syncthreads();
  if (tid == 0) sc.wgdesc[blockIdx.x] = make_uint2(R, s_ng);
  MRH_SC_TS(0, 5);
}

// The blocks this scan touched are found by their stamps (no list is kept while the beams walk), by the kernel that needs them:
// a workgroup of k_scan_offsets reads a window of kScanWindow stamps, lists the hits in LDS and turns their counters into offsets,
// one wave per touched block, 16 blocks per round: counters -> offsets, chunks for k_scan_apply.  The 16 blocks reserve their
// records and their chunk slots with ONE atomic each per round.  (Until round 4 a launch of its own collected the touched blocks
// into a global list first: 5 us of a 98 us scan and one more crossing of HBM.)
constexpr u32 kScanWindow = 16;  // = the blocks of one round: a scan touches most of the blocks near the sensor, so a window is
                                 // mostly hits and is done in ONE round (with 128-block windows a workgroup walked six rounds in a row)
__global__ __launch_bounds__(1024) void k_scan_offsets(const Tab t, const Scan sc, const u32 n_blocks_or_0) {
  __shared__ u32 s_excl[16][513];
  __shared__ unsigned short s_start[16][514];
  __shared__ u32 s_tot[16], s_nch[16], s_nbig[16], s_rbase, s_cbase, s_bbase;
  __shared__ u32 s_hit[kScanWindow], s_nhit;
  static_assert(kScanWindow <= 64, "one wave reads the window's stamps");
  static_assert(SC_PLACED == 0 && SC_CHUNKS == 1, "one 64-bit word");
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x == 0 && tid < (u32) SC_N) sc.ctr_next[tid] = 0;  // the next scan's counters (this scan's walk is complete)
  MRH_SC_TS(1, 0);
  const u32 nb = n_blocks_or_0 ? n_blocks_or_0 : (u32) t.ctr[CTR_HWM_FINE];
  for (u32 h0 = blockIdx.x * kScanWindow; h0 < nb; h0 += gridDim.x * kScanWindow) {
    // ---- the window's touched blocks, in block order (wave 0 reads the stamps)
    if (wave == 0) {
      const u32 Hs = h0 + lane;
      const u32 st = (lane < kScanWindow && Hs < nb) ? sc.bstamp[Hs] : 0u;
      const bool hit = lane < kScanWindow && (st >> 1) == sc.seq;
      const u64 bal = __ballot(hit);
      if (hit) s_hit[(u32) __popcll(bal & lanemask_lt())] = Hs | ((st & 1u) << 31);
      if (lane == 0) s_nhit = (u32) __popcll(bal);
    }
    __syncthreads();
    const u32 nt = s_nhit;
  for (u32 t0 = 0; t0 < nt; t0 += 16u) {
    const u32 ti = t0 + wave;
    const bool have = ti < nt;
    const u32 Hc = have ? s_hit[ti] : 0u, H = Hc & ~kScanCoarse;
    uint4* p = (uint4*) (sc.vcnt + (size_t) H * 512 + lane * 8);
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (have) { a = p[0]; b = p[1]; }
    const u32 cv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32 w[8], sumc = 0, sumw = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32 cc = min(cv[k], 1024u);
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
    // chunks of this block: small ones (a wave's work) from the front of the list, runs beyond a wave's slice from its end
    if (lane == 0) s_start[wave][nch_all] = 512;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32 nbig = 0;
    for (u32 k0 = 0; k0 < nch; k0 += 64) {
      const u32 k = k0 + lane;
      bool big = false;
      if (k < nch) {
        const u32 v0 = s_start[wave][k], v1 = s_start[wave][k + 1];
        big = v1 - v0 == 1 && s_excl[wave][v1] - s_excl[wave][v0] > kScanWaveRecs;
      }
      nbig += (u32) __popcll(__ballot(big));
    }
    if (lane == 0) { s_tot[wave] = total; s_nch[wave] = nch - nbig; s_nbig[wave] = nbig; }
    __syncthreads();
    // Every workgroup with touched blocks reserves here, at the same time, from the same counters: returning atomics on one
    // address are served one after the other (~15 ns each), 250 workgroups x 2 counters were 7 of this launch's 10 us.  Records
    // and chunks are reserved with ONE 64-bit add (neither half can carry: the host bounds both below 2^32); the runs beyond a
    // wave's slice — rare — keep their own, on a second lane so that the round trips overlap.
    if (tid == 0) {
      u32 rec_sum = 0, ch_sum = 0;
      for (int k = 0; k < 16; k++) { rec_sum += s_tot[k]; ch_sum += s_nch[k]; }
      const u64 add = (u64) rec_sum | ((u64) ch_sum << 32);
      const u64 got = add ? atomicAdd((unsigned long long*) &sc.ctr[SC_PLACED], (unsigned long long) add) : 0ull;
      s_rbase = (u32) got; s_cbase = (u32) (got >> 32);
    } else if (tid == 64) {
      u32 sum = 0;
      for (int k = 0; k < 16; k++) sum += s_nbig[k];
      s_bbase = sum ? atomicAdd(&sc.ctr[SC_BIG], sum) : 0u;
    }
    __syncthreads();
    if (total) {
      u32 base = s_rbase, cbase = s_cbase, bbase = s_bbase, call = 0, ball = 0;
      for (u32 k = 0; k < 16; k++) {
        if (k < wave) { base += s_tot[k]; cbase += s_nch[k]; bbase += s_nbig[k]; }
        call += s_nch[k]; ball += s_nbig[k];
      }
      p[0] = make_uint4(base + st[0], base + st[1], base + st[2], base + st[3]);
      p[1] = make_uint4(base + st[4], base + st[5], base + st[6], base + st[7]);
      if (s_cbase + call + s_bbase + ball > sc.chunk_cap) {
        if (lane == 0) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_SCAN);
      } else {
        for (u32 k0 = 0; k0 < nch; k0 += 64) {
          const u32 k = k0 + lane;
          bool big = false, small = false;
          uint4 d = make_uint4(0, 0, 0, 0);
          if (k < nch) {
            const u32 v0 = s_start[wave][k], v1 = s_start[wave][k + 1];
            d = make_uint4(Hc, v0 | (v1 << 16), base + s_excl[wave][v0], base + s_excl[wave][v1]);
            big = v1 - v0 == 1 && d.w - d.z > kScanWaveRecs;
            small = !big;
          }
          const u64 bs = __ballot(small), bb = __ballot(big);
          if (small) sc.chunks[cbase + (u32) __popcll(bs & lanemask_lt())] = d;
          if (big) sc.chunks[sc.chunk_cap - 1u - (bbase + (u32) __popcll(bb & lanemask_lt()))] = d;
          cbase += (u32) __popcll(bs);
          bbase += (u32) __popcll(bb);
        }
      }
    }
    __syncthreads();
  }
    __syncthreads();  // s_hit is rewritten by the next window
  }
  MRH_SC_TS(1, 1);
}

// Records of one walk workgroup -> their voxels' runs: slot = start of the run (k_scan_offsets) + the records that had arrived
// before the group (k_scan_walk's atomic) + rank inside the group.  No atomics.
__global__ __launch_bounds__(256) void k_scan_place(const Scan sc, const int slots) {
  __shared__ u32 s_base[256 * kScanMaxSlots];
  __shared__ unsigned short s_li[256 * kScanMaxSlots];
  MRH_SC_TS(2, 0);
  const uint2 d = sc.wgdesc[blockIdx.x];
  const u32 off = blockIdx.x * 256u * (u32) slots, R = d.x, G = d.y;
  for (u32 g = threadIdx.x; g < G; g += 256) {
    const uint2 grp = sc.st_grp[off + g];
    s_base[g] = sc.vcnt[grp.x & ~kScanCoarse] + grp.y;
    s_li[g] = (unsigned short) (grp.x & 511u);
  }
  __syncthreads();
  for (u32 r = threadIdx.x; r < R; r += 256) {
    uint2 meta;
    if (sc.ord_shift) meta = sc.st_meta[off + r];
    else { const u32 w = ((const u32*) sc.st_meta)[off + r]; meta = make_uint2(w & 0x1FFFFFu, (w >> 21) << 5); }
    const float sdf = sc.st_sdf[off + r];
    const u32 slot = s_base[meta.x & 0x1FFFu] + (meta.x >> 13);
    if (slot < sc.rec_cap) {
      const u32 pidx = sc.order.point(blockIdx.x, meta.y >> 5);
      scan_store_rec(sc, slot, sc.ord_shift ? (pidx << 5) | (meta.y & 31u) : pidx, sdf, s_li[meta.x & 0x1FFFu]);
    }
  }
  MRH_SC_TS(2, 1);
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
  // A LONG run: the step's state apart from the running mean — the weight before step i, min(W + i w1, wmax), and the refined
  // reciprocal of the weight sum — does not depend on the records.  Until the weight has reached its clamp (at most 255 steps)
  // the other lanes lay that state out in LDS (fill_tables), after that it is constant: the one lane that folds is left with
  // the chain itself — multiply, add, and the three operations of div_cr per record, a fifth of the instructions of step().
  // Same operations on the same values: same bits.
  static constexpr u32 kTable = 256;
  __device__ __forceinline__ u32 steps_to_clamp() const {  // first step whose weight-before is wmax (w1 >= 1); 0 when W >= wmax
    const u32 W = rgb0 >> 24;
    if (W >= wmax || w1 == 0) return W == wmax ? 0u : 0xFFFFFFFFu;
    return (wmax - W + w1 - 1) / w1;
  }
  template <int NT>
  __device__ __forceinline__ void fill_tables(const u32 me, float* x, float* a, float* r, const u32 n) const {
    const u32 W = rgb0 >> 24;
    const u32 nt = min(min(steps_to_clamp(), kTable), n - 1);
    for (u32 i = me; i < nt; i += NT) {
      const u32 wi = i ? min(W + i * w1, wmax) : W;
      a[i] = (float) wi;
      r[i] = rcp_refined((float) (int) (wi + w1));
    }
    for (u32 i = me; i + 1 < n; i += NT) x[i] = x[i] * w1f;  // the last record keeps its value: end() needs it
  }
  __device__ __forceinline__ void run_tables(const float* x, const float* a, const float* r, const u32 n) {
    const u32 W = rgb0 >> 24, clamp_at = steps_to_clamp();
    float s = s0;
    u32 i = 0;
    if (!two && clamp_at <= kTable) {
      const u32 nt = min(clamp_at, n - 1);
      // until the clamp the weight before step i is W + i w1 (< wmax: a small integer, exact in fp32 — the table's a[i], counted
      // up here instead of loaded); the records and the reciprocals of four steps are in registers while the four before them
      // are folded, so the chain of a record is its arithmetic, not an LDS round trip per step (a new voxel's run of ~150 records
      // is what the launch waited for: 24 of its 25 us, tools/trace_scan.py)
      if (i + 4 <= nt) {
        float aw_i = (float) W, x0 = x[i], x1 = x[i + 1], x2 = x[i + 2], x3 = x[i + 3], q0 = r[i], q1 = r[i + 1], q2 = r[i + 2], q3 = r[i + 3];
        for (; i + 8 <= nt; i += 4) {
          const float y0 = x[i + 4], y1 = x[i + 5], y2 = x[i + 6], y3 = x[i + 7], p0 = r[i + 4], p1 = r[i + 5], p2 = r[i + 6], p3 = r[i + 7];
          float an = aw_i + w1f;
          s = div_cr(s * aw_i + x0, an, q0); aw_i = an; an += w1f;
          s = div_cr(s * aw_i + x1, an, q1); aw_i = an; an += w1f;
          s = div_cr(s * aw_i + x2, an, q2); aw_i = an; an += w1f;
          s = div_cr(s * aw_i + x3, an, q3); aw_i = an;
          x0 = y0; x1 = y1; x2 = y2; x3 = y3; q0 = p0; q1 = p1; q2 = p2; q3 = p3;
        }
        float an = aw_i + w1f;
        s = div_cr(s * aw_i + x0, an, q0); aw_i = an; an += w1f;
        s = div_cr(s * aw_i + x1, an, q1); aw_i = an; an += w1f;
        s = div_cr(s * aw_i + x2, an, q2); aw_i = an; an += w1f;
        s = div_cr(s * aw_i + x3, an, q3);
        i += 4;
      }
      for (; i < nt; i++) s = div_cr(s * a[i] + x[i], a[i] + w1f, r[i]);
      const float aw = (float) wmax, dw = (float) (int) (wmax + w1), rw = rcp_refined(dw);
      // The clamped weight sum of the shipped configurations is a power of two (255 + 1): dividing by it is a multiplication by
      // an exact reciprocal, ONE rounding — the IEEE quotient in every case, subnormal results included — so the chain of a
      // record is multiply, add, multiply instead of multiply, add and the three operations of div_cr (which return that
      // same value: q is exact, the residual is zero).  The longest run of a scan is what its last launch waits for.
      const u32 dwi = wmax + w1;
      if ((dwi & (dwi - 1u)) == 0u && i + 4 < n) {
        float x0 = x[i], x1 = x[i + 1], x2 = x[i + 2], x3 = x[i + 3];
        for (; i + 8 < n; i += 4) {
          const float y0 = x[i + 4], y1 = x[i + 5], y2 = x[i + 6], y3 = x[i + 7];
          s = (s * aw + x0) * rw;
          s = (s * aw + x1) * rw;
          s = (s * aw + x2) * rw;
          s = (s * aw + x3) * rw;
          x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        }
        s = (s * aw + x0) * rw;
        s = (s * aw + x1) * rw;
        s = (s * aw + x2) * rw;
        s = (s * aw + x3) * rw;
        i += 4;
      }
      if (i + 4 < n) {  // four records in registers while the four before them are folded: no step waits for the LDS
        float x0 = x[i], x1 = x[i + 1], x2 = x[i + 2], x3 = x[i + 3];
        for (; i + 8 < n; i += 4) {
          const float y0 = x[i + 4], y1 = x[i + 5], y2 = x[i + 6], y3 = x[i + 7];
          s = div_cr(s * aw + x0, dw, rw);
          s = div_cr(s * aw + x1, dw, rw);
          s = div_cr(s * aw + x2, dw, rw);
          s = div_cr(s * aw + x3, dw, rw);
          x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        }
        s = div_cr(s * aw + x0, dw, rw);
        s = div_cr(s * aw + x1, dw, rw);
        s = div_cr(s * aw + x2, dw, rw);
        s = div_cr(s * aw + x3, dw, rw);
        i += 4;
      }
      for (; i + 1 < n; i++) s = div_cr(s * aw + x[i], dw, rw);
    } else {  // a divisor that needs the full division, or a weight above its clamp: the general step (x is scaled by w1 already)
      for (; i + 1 < n; i++) {
        const u32 wi = i ? min(W + i * w1, wmax) : W;
        const float num = s * (float) wi + x[i], den_i = (float) (int) (wi + w1);
        s = two ? num / den_i : div_cr(num, den_i, rcp_refined(den_i));
      }
    }
    // the state before the last record, as step() would have left it
    const u32 last = n - 1;
    s0 = s;
    w0 = last ? min(W + last * w1, wmax) : W;
    wsum = w0 + w1;
    w0f = (float) w0; den = (float) (int) wsum; rden = rcp_refined(den);
    pend = x[last];
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

__device__ __forceinline__ void wave_sync() {  // LDS written by the lanes of this wave is visible to all of them
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Chunks -> voxels.  A chunk (< 512 records: a typical block has ~300 in ~50 voxels) is the work of ONE wave, in its slice of
// the LDS, without workgroup barriers: 6 000 waves are resident, so all chunks of a scan are in flight at once and the kernel
// lasts as long as its longest chain (a long run: sort + fold), not as long as a queue of chunks.  The few runs beyond a
// wave's slice follow, a workgroup each.
struct alignas(16) ApplyWaveLds {
  u32 tag[kScanWaveRecs];
  float sdf[kScanWaveRecs];
  union {
    struct { unsigned short st[514]; unsigned short order[512]; };  // st[u] = first record of run u, st[u + 1] = one past its last
    struct { float a[VoxFold::kTable]; float r[VoxFold::kTable]; };  // a long run: the weights / reciprocals until the clamp
  };
  u32 bcnt[16];
};

// bitonic network over (tag, value) pairs in LDS; NT lanes, every stage's pairs read before any is written
template <int NT, typename Sync>
__device__ __forceinline__ void bitonic_lds(u32* tag, float* val, const u32 N, const u32 me, Sync&& sync) {
  for (u32 k = 2; k <= N; k <<= 1) {
    for (u32 j = k >> 1; j > 0; j >>= 1) {
      for (u32 i0 = 0; i0 < N / 2; i0 += 4 * NT) {
        u32 lo[4], ta[4], tb[4];
        float va[4], vb[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const u32 i = i0 + q * NT + me;
          lo[q] = ((i & ~(j - 1)) << 1) | (i & (j - 1));
          if (i < N / 2) { ta[q] = tag[lo[q]]; tb[q] = tag[lo[q] | j]; va[q] = val[lo[q]]; vb[q] = val[lo[q] | j]; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const u32 i = i0 + q * NT + me;
          if (i < N / 2 && (ta[q] > tb[q]) == ((lo[q] & k) == 0)) {
            tag[lo[q]] = tb[q]; tag[lo[q] | j] = ta[q];
            val[lo[q]] = vb[q]; val[lo[q] | j] = va[q];
          }
        }
      }
      sync();
    }
  }
}

__global__ __launch_bounds__(256, 6) void k_scan_apply(const Map m, const Tab t, const Scan sc, const u32 tag_space, const int count_updates) {
  __shared__ ApplyWaveLds s_w[4];
  __shared__ u32 s_part[4];
  static_assert(sizeof(ApplyWaveLds) * 4 >= kScanBigRecs * 12, "the workgroup pass reuses the waves' slices");
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 n_small = min(sc.ctr[SC_CHUNKS], sc.chunk_cap);
  const u32 n_big = min(sc.ctr[SC_BIG], sc.chunk_cap - n_small);
  // k_scan_offsets checks each workgroup's reservation against the capacity, but the small chunks grow from the front of the
  // array and the long runs from its end: two late reservations can each pass and still overlap.  The final totals decide.
  if (blockIdx.x == 0 && tid == 0 && (u64) sc.ctr[SC_CHUNKS] + (u64) sc.ctr[SC_BIG] > (u64) sc.chunk_cap) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_SCAN);
  u32 updated = 0;
  MRH_SC_TS(3, 0);
  ApplyWaveLds& L = s_w[wave];
  // Runs beyond a wave's slice are the longest chains of the launch (a sort by the whole workgroup, then ONE lane folds up to a
  // few thousand records, ~17 ns each): the workgroups [0, nb_wg) take them, and nothing else, from the first cycle on; the chunks
  // of a wave's size go to the other workgroups.  (Until round 5 every workgroup walked its chunks first and the long runs started
  // 5-10 us into the launch: the launch lasted 26 us with 90 % of its workgroups done after 15, tools/trace_scan.py.)
  const u32 nb_wg = n_big < gridDim.x / 2 ? n_big : 0u;  // uniform; 0: more long runs than spare workgroups — everybody does both
  const bool big_only = blockIdx.x < nb_wg;
  const u32 small_wgs = gridDim.x - nb_wg;
  for (u32 ci = big_only ? n_small : (blockIdx.x - nb_wg) * 4u + wave; ci < n_small; ci += small_wgs * 4u) {
    const uint4 ch = sc.chunks[ci];
    const u32 H = ch.x & ~kScanCoarse, v0 = ch.y & 0xFFFFu, v1 = ch.y >> 16, r0 = ch.z, nr = ch.w - ch.z, nv = v1 - v0;
    const bool coarse = (ch.x & kScanCoarse) != 0;
    const bool long_run = nv == 1 && nr > kScanLongRun;
    u32* pc = sc.vcnt + (size_t) H * 512 + v0 + lane;
    if (nr == 0 || long_run) {  // nothing to order by voxel: the counters are zero again for the next scan
#pragma unroll
      for (int r = 0; r < 8; r++)
        if (lane + 64u * r < nv) pc[64 * r] = 0;
      if (nr == 0) continue;
    }
    if (long_run) {  // one long run: its records into point order (tags are distinct), then one lane folds
      VoxFold f;
      if (nr <= 256) {
        // up to four records a lane: a record's place is the number of smaller tags, counted over the run's tags in LDS four at
        // a time (every lane reads the same words: broadcasts) — a third of the instructions of the network's 28-36 stages
        u32 tg[4], rank[4];
        float sv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const u32 q = lane + 64u * r;
          const uint4 rc = q < nr ? scan_load_rec(sc, r0 + q) : make_uint4(kScanEmpty, 0u, 0u, 0u);
          tg[r] = rc.x; sv[r] = __uint_as_float(rc.y); rank[r] = 0;
          L.tag[q] = rc.x;  // the padding (all ones) is larger than any tag: it counts for nobody
        }
        f.begin(m, t, H, coarse, v0);  // every lane (one address): the tables start from the voxel's weight
        wave_sync();
        const u32 n4 = (nr + 3u) & ~3u;
        for (u32 j = 0; j < n4; j += 4) {
          const uint4 o = *(const uint4*) &L.tag[j];
#pragma unroll
          for (int r = 0; r < 4; r++) rank[r] += (o.x < tg[r] ? 1u : 0u) + (o.y < tg[r] ? 1u : 0u) + (o.z < tg[r] ? 1u : 0u) + (o.w < tg[r] ? 1u : 0u);
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (lane + 64u * r < nr) L.sdf[rank[r]] = sv[r];
        wave_sync();
      } else {
        u32 N = 512;
        for (u32 q = lane; q < N; q += 64) {
          const uint4 rc = q < nr ? scan_load_rec(sc, r0 + q) : make_uint4(kScanEmpty, 0u, 0u, 0u);
          L.tag[q] = rc.x;
          L.sdf[q] = __uint_as_float(rc.y);
        }
        f.begin(m, t, H, coarse, v0);
        wave_sync();
        bitonic_lds<64>(L.tag, L.sdf, N, lane, [] { wave_sync(); });
      }
      f.fill_tables<64>(lane, L.sdf, L.a, L.r, nr);
      wave_sync();
      if (lane == 0) {
        f.run_tables(L.sdf, L.a, L.r, nr);
        f.end();
        updated++;
      }
      wave_sync();
      MRH_SC_TSW(3, 3 + wave, min(nr, 1023u) | (1u << 19));
      continue;
    }
    // all loads of the chunk first (a store in between would order them): where the runs start, the records
    u32 stv[8], tg[8], li[8];
    float sv[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      stv[r] = lane + 64u * r < nv ? pc[64 * r] : 0u;
      const u32 q = lane + 64u * r;
      tg[r] = 0; li[r] = 0; sv[r] = 0.f;
      if (q < nr) { const uint4 rc = scan_load_rec(sc, r0 + q); tg[r] = rc.x; sv[r] = __uint_as_float(rc.y); li[r] = rc.z; }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (lane + 64u * r < nv) { L.st[lane + 64u * r] = (unsigned short) (stv[r] - r0); pc[64 * r] = 0; }
      if (lane + 64u * r < nr) L.tag[lane + 64u * r] = tg[r];
    }
    if (lane == 0) L.st[nv] = (unsigned short) nr;
    if (lane < 16) L.bcnt[lane] = 0;
    wave_sync();
    // rank of every record inside its run (the tags smaller than its own), then the values into run order
    u32 pos[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      pos[r] = 0;
      if (lane + 64u * r < nr) {
        const u32 u = li[r] - v0, a = L.st[u], b = L.st[u + 1];
        u32 rank = 0;
        for (u32 j = a; j < b; j++) rank += L.tag[j] < tg[r] ? 1u : 0u;
        pos[r] = a + rank;
      }
    }
#pragma unroll
    for (int r = 0; r < 8; r++)
      if (lane + 64u * r < nr) L.sdf[pos[r]] = sv[r];
    // the voxels in order of falling run length (by power of two): the lanes then fold runs of similar length at the same
    // time, and a wave is as slow as its longest run
    u32 myb[8], mypos[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const u32 u = lane + 64u * r;
      myb[r] = 16;
      if (u < nv) {
        const u32 len = (u32) L.st[u + 1] - (u32) L.st[u];
        if (len) { myb[r] = 31u - (u32) __clz(len); mypos[r] = atomicAdd(&L.bcnt[myb[r]], 1u); }
      }
    }
    wave_sync();
    u32 nnz = 0;
    {
      u32 boff[16], run = 0;
#pragma unroll
      for (int b = 15; b >= 0; b--) { boff[b] = run; run += L.bcnt[b]; }
      nnz = run;
#pragma unroll
      for (int r = 0; r < 8; r++)
        if (myb[r] < 16) {
          u32 o = 0;
#pragma unroll
          for (int b = 0; b < 16; b++) o = myb[r] == (u32) b ? boff[b] : o;
          L.order[o + mypos[r]] = (unsigned short) (lane + 64u * r);
        }
    }
    wave_sync();
    for (u32 i = lane; i < nnz; i += 64) {
      const u32 u = L.order[i];
      const u32 a = L.st[u], b = L.st[u + 1];
      VoxFold f;
      f.begin(m, t, H, coarse, v0 + u);
      f.run(L.sdf + a, b - a);
      f.end();
      updated++;
    }
    wave_sync();
    MRH_SC_TSW(3, 3 + wave, min(nr, 1023u) | (min(nv, 511u) << 10));
  }
  // runs beyond a wave's slice: one workgroup each, over the same LDS
  __syncthreads();
  MRH_SC_TS(3, 1);
  u32* s_tag = (u32*) &s_w[0];
  float* s_sdf = (float*) (s_tag + kScanBigRecs);
  float* s_sorted = s_sdf + kScanBigRecs;  // windows only; the tables of the bitonic path lie here
  constexpr int PER = kScanBigRecs / 256;
  for (u32 bi = nb_wg ? (big_only ? blockIdx.x : n_big) : blockIdx.x; bi < n_big; bi += nb_wg ? nb_wg : gridDim.x) {
    const uint4 ch = sc.chunks[sc.chunk_cap - 1u - bi];
    const u32 H = ch.x & ~kScanCoarse, v0 = ch.y & 0xFFFFu, r0 = ch.z, nr = ch.w - ch.z;
    const bool coarse = (ch.x & kScanCoarse) != 0;
    if (tid == 0) sc.vcnt[(size_t) H * 512 + v0] = 0;
    VoxFold f;
    f.begin(m, t, H, coarse, v0);
    if (nr <= kScanBigRecs) {
      u32 N = 1024;
      while (N < nr) N <<= 1;
      for (u32 q = tid; q < N; q += 256) {
        const uint4 rc = q < nr ? scan_load_rec(sc, r0 + q) : make_uint4(kScanEmpty, 0u, 0u, 0u);
        s_tag[q] = rc.x;
        s_sdf[q] = __uint_as_float(rc.y);
      }
      __syncthreads();
      bitonic_lds<256>(s_tag, s_sdf, N, tid, [] { __syncthreads(); });
      f.fill_tables<256>(tid, s_sdf, s_sorted, s_sorted + VoxFold::kTable, nr);
      __syncthreads();
      if (tid == 0) f.run_tables(s_sdf, s_sorted, s_sorted + VoxFold::kTable, nr);
    } else {  // beyond the LDS: windows over the tag range, every tag is unique inside a voxel -> direct addressing
      for (u32 w0 = 0; w0 < tag_space; w0 += kScanBigRecs) {
        for (u32 q = tid; q < kScanBigRecs; q += 256) s_tag[q] = 0;
        __syncthreads();
        for (u32 q = tid; q < nr; q += 256) {
          const uint4 rc = scan_load_rec(sc, r0 + q);
          const u32 d = rc.x - w0;
          if (d < kScanBigRecs) { s_sdf[d] = __uint_as_float(rc.y); s_tag[d] = 1; }
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
  }
  if (count_updates && updated) atomicAdd(&t.prof[PROF_UPDATED], (u64) updated);  // profile mode: voxels this scan updated
  MRH_SC_TS(3, 2);
}

}  // namespace mrh
