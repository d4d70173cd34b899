// mrh_lidar.h — LiDAR scans: VoxelContainer::integrate(point_cloud, normals, weights, camera, max_num_frames)
// (voxel_data_structures.cpp:112-135): projective or normal-direction SDF (one caller-supplied normal per point), fine and
// coarse blocks (variance-adaptive maps re-integrate the scan after coarsening, vds.cu:1561-1580); garbage collection
// and the starve step run through the general kernels (mrh_kernels.h) on the list of all live blocks.
//
//   k_alloc3d        allocBlocks3DKernel vds.cu:925-1033 + the host retry loop :1036-1092.  256 points per workgroup:
//                    block-level DDA over the segment range -+ truncation along the beam, keys de-duplicated in an LDS
//                    set, then one probe per distinct key and the lock-free insert of mrh_fast2.h (no mutex, no retry).
//   k_points_walk    integrate3DKernel vds.cu:1215-1379, split in two so that the result does not depend on a race:
//                    the reference updates a voxel with a non-atomic read-modify-write per point.  Here a point's
//                    VOXEL-level DDA only produces (voxel id, clamped sdf) records — counted, prefix-summed and written at
//                    exact offsets in point-major order, no atomics.  The records are sorted by voxel id with a STABLE radix
//                    sort (mrh_sort.h, over the id's significant bits), which keeps every voxel's records in ascending
//                    point index, and
//   k_points_apply   folds each voxel's run in that order (combineVoxel, vhu.cuh:167-181, and the variance term) — the
//                    oracle's sequential order (D6).  A workgroup stages a chunk of the sorted records in LDS; the lane at
//                    the head of a run walks it there instead of through dependent global loads (round 2: 57 us per scan).
#pragma once


#include "mrh_fast2.h"

namespace mrh {

__device__ __forceinline__ float norm3(const f3 p) { return sqrtf((p.x * p.x + p.y * p.y) + p.z * p.z); }  // norm3df restated
__device__ __forceinline__ f3 normalize3(const f3 p) {
  const float inv = 1.0f / sqrtf((p.x * p.x + p.y * p.y) + p.z * p.z);
  return mk3(p.x * inv, p.y * inv, p.z * inv);
}
__device__ __forceinline__ i3 voxel_to_block_fast(const Map& m, const i3 v) {
  const int ax = v.x < 0 ? -v.x : v.x, ay = v.y < 0 ? -v.y : v.y, az = v.z < 0 ? -v.z : v.z;
  if ((u32) (ax | ay | az) < (u32) m.block_shift_limit) return mki3(v.x >> 3, v.y >> 3, v.z >> 3);
  return voxel_to_block(v, m.vs);
}

// record key = voxel id: fine block index * 512 + local index; on a coarse unit u = 8 H + k: (H * 512 + k * 64 + local index) | coarse bit,
// coarse bit = 1 << (9 + bits of the pool capacity) — the host picks the key width (32 or 64 bits) from it.

// Which 256 beams a workgroup takes.  An ORGANISED scan (rows of row_len points, row-major: what a LiDAR driver delivers) is taken
// in patches of (256 >> patch_log2) rows x (1 << patch_log2) columns instead of 256 consecutive points of one row: beams that leave
// the sensor side by side — in BOTH directions of the scan image — end in the same blocks and voxels, so a workgroup has several
// times fewer distinct ones to insert, group and count.  patch_log2 = 8: 256 consecutive points (any other cloud).  The order is a
// matter of speed only: a record carries its point index, and a voxel's records are put into point order before they are folded.
struct BeamOrder {
  u32 patch_log2, patches_per_row, row_len;
  __device__ __forceinline__ u32 point(const u32 wg, const u32 tid) const {
    if (patch_log2 >= 8u) return wg * 256u + tid;
    const u32 pr = wg / patches_per_row, pc = wg - pr * patches_per_row;
    const u32 r = (pr << (8u - patch_log2)) + (tid >> patch_log2), col = (pc << patch_log2) + (tid & ((1u << patch_log2) - 1u));
    return r * row_len + col;
  }
};

__global__ __launch_bounds__(256) void k_alloc3d(const Cam c, const Map m, const Tab t, const Fast f, const float* __restrict__ pts,
                                                 const float* __restrict__ normals, const u32 n, const u32 stamp, const BeamOrder order) {
  __shared__ FrontShared sh;
  constexpr int NT = 256;
  const int tid = threadIdx.x;
  const int hwm0 = t.ctr[CTR_HWM_FINE];
  for (int i = tid; i < kRayCap; i += NT) sh.set[i] = kKeyEmpty;
  if (tid == 0) sh.count = 0;
  __syncthreads();
  auto insert_direct = [&](const i3 cur, const u64 key) {
    const int slot = hash_insert(t, key);
    if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
    if (slot < 0) return;
    (void) commit_block(t, f, slot, atomicSub(&t.ctr[CTR_HEAP_FINE], 1), cur, stamp, hwm0);
  };
  const u32 i = order.point(blockIdx.x, (u32) tid);
  if (i < n) {
    const f3 pcam = mk3(pts[3 * (size_t) i], pts[3 * (size_t) i + 1], pts[3 * (size_t) i + 2]);
    const float range = norm3(pcam);
    const float tr = get_truncation(range, m.trunc, m.trunc_scale);
    const float dmin = fminf(c.max_int_dist, range - tr), dmax = fminf(c.max_int_dist, range + tr);
    if (range != 0.f && !(dmin >= dmax)) {
      // vds.cu:957-962: along the beam (projective SDF) or along the point's normal
      const f3 dir = normals ? normalize3(mk3(normals[3 * (size_t) i], normals[3 * (size_t) i + 1], normals[3 * (size_t) i + 2])) : normalize3(pcam);
      const float a = dmin - range, b = dmax - range;
      const f3 pw_min = se3_apply(c.R, c.t, mk3(pcam.x + dir.x * a, pcam.y + dir.y * a, pcam.z + dir.z * a));
      const f3 pw_max = se3_apply(c.R, c.t, mk3(pcam.x + dir.x * b, pcam.y + dir.y * b, pcam.z + dir.z * b));
      const RayState ray = ray_from_segment(m, pw_min, pw_max);
      bool slow = !ray_keys_in_range(ray);
      if (!slow) {
        slow = !walk_ray_lean(m, ray, [&](const i3 cur, const u64 key) {
          u32 s = (u32) __mul24(cur.z, 5851) + (u32) __mul24(cur.y, 73) + (u32) cur.x;
          s = (s ^ (s >> 7)) & (kRayCap - 1);
#pragma unroll 1
          for (int p = 0; p < kSetProbe; p++) {
            const u64 old = atomicCAS(&sh.set[s], kKeyEmpty, key);
            if (old == kKeyEmpty) { sh.list[atomicAdd(&sh.count, 1u)] = key; return true; }
            if (old == key) return true;
            s = (s + 1) & (kRayCap - 1);
          }
          return false;
        });
      }
      if (slow) {  // LDS set saturated or keys out of range: the literal walk with direct inserts
        RayState r = ray;
#pragma unroll 1
        for (u32 iter = 0; iter < kMaxDdaIter; iter++) {
          u64 key;
          if (!pack_key(r.cur, key)) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_RANGE);
          else if (owns_block(m, r.cur)) insert_direct(r.cur, key);
          const bool ax = r.t_max.x < r.t_max.y && r.t_max.x < r.t_max.z;
          const bool az = !ax && (r.t_max.z < r.t_max.y);
          const bool ay = !ax && !az;
          r.cur.x += ax ? r.step.x : 0; r.cur.y += ay ? r.step.y : 0; r.cur.z += az ? r.step.z : 0;
          if ((ax && r.cur.x == r.bound.x) || (ay && r.cur.y == r.bound.y) || (az && r.cur.z == r.bound.z)) break;
          r.t_max.x = ax ? r.t_max.x + r.t_delta.x : r.t_max.x;
          r.t_max.y = ay ? r.t_max.y + r.t_delta.y : r.t_max.y;
          r.t_max.z = az ? r.t_max.z + r.t_delta.z : r.t_max.z;
        }
      }
    }
  }
  __syncthreads();
  const int nk = (int) sh.count;
#pragma unroll 1
  for (int base = 0; base < nk; base += NT) {
    const int j = base + tid;
    const bool active = j < nk;
    const u64 key = active ? sh.list[j] : kKeyEmpty;
    const i3 b = active ? unpack_key(key) : mki3(0, 0, 0);
    bool won = false;
    int slot = -1, claim;
    u64 claim_val;
    if (active && hash_find_claim(t, key, claim, claim_val) < 0) {  // no frustum test on this path (vds.cu:1010)
      slot = hash_insert_at(t, key, claim, claim_val);
      if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
      won = slot >= 0;
    }
    const u64 ballot = __ballot(won);
    if (ballot) {
      const int leader = __ffsll((long long) ballot) - 1;
      int hb = 0;
      if ((int) lane_id() == leader) hb = atomicSub(&t.ctr[CTR_HEAP_FINE], __popcll(ballot));
      hb = __shfl(hb, leader);
      if (won) (void) commit_block(t, f, slot, hb - __popcll(ballot & lanemask_lt()), b, stamp, hwm0);
    }
  }
}

// One record per (point, traversed voxel of an allocated block) up to the first voxel with sdf <= -truncation.
// EMIT = false: counts[i] = number of records of point i.  EMIT = true: records written at offsets[i] ...: key = voxel id,
// value = the clamped sdf.
// The exact offsets need no scan kernel and no offsets array (round 3b): the count pass leaves counts[i] and one total per
// workgroup (256 consecutive points); an emit workgroup adds up the totals of the workgroups before it (at most a few hundred
// words, one coalesced read) and scans its own 256 counts in LDS.  The LAST emit workgroup thereby knows the grand total before
// its own walk starts and writes the scan's one report into pinned host memory: {high-water mark, records, 0}, then the
// sequence mark.  (A ticket in the count pass — the last workgroup to finish adds up the totals — cost that pass 9 us: 512
// same-address atomics behind a fence each, and the reduction on the launch's critical path.)
struct ScanState {
  u32* wg_totals;   // [grid]
  u32* host_rec;    // pinned [4]
  u32 seq;
};
__device__ __forceinline__ u32 wg_sum_256(const u32 v, u32* s_part) {  // sum over the 256 threads of the workgroup, to every thread
  u32 x = v;
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = x;
  __syncthreads();
  const u32 r = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  __syncthreads();
  return r;
}
// One beam of integrate3DKernel (vds.cu:1215-1379): the voxel-level DDA over the segment range -+ truncation; rec(val, res, li,
// sdf) is called for every traversed voxel of an allocated block (val = its table value, res = 1 on a coarse unit, li = the
// voxel's index in the block, sdf clamped) up to the first voxel with sdf <= -truncation.  Both record paths (sorted records
// below, voxel buckets in mrh_scan.h) walk through this one function: the arithmetic exists once.
template <typename Rec>
__device__ __forceinline__ void walk_beam(const Cam& c, const Map& m, const Tab& t, const float* __restrict__ pts,
                                          const float* __restrict__ normals, const u32 i, const bool live, Rec&& rec) {
  const f3 pcam = live ? mk3(pts[3 * (size_t) i], pts[3 * (size_t) i + 1], pts[3 * (size_t) i + 2]) : mk3(0.f, 0.f, 0.f);
  const float range = norm3(pcam);
  const float tr = get_truncation(range, m.trunc, m.trunc_scale);
  const float dmin = fminf(c.max_int_dist, range - tr), dmax = fminf(c.max_int_dist, range + tr);
  if (live && !((double) range < 1e-6 || range > c.max_int_dist) && !(dmin >= dmax)) {
    const f3 dir0 = normalize3(pcam);
    f3 norm_dir = mk3(0.f, 0.f, 0.f);
    f3 pc_min, pc_max;
    if (!normals) {  // projective (vds.cu:1245-1247)
      pc_min = mk3(pcam.x - dir0.x * tr, pcam.y - dir0.y * tr, pcam.z - dir0.z * tr);
      pc_max = mk3(pcam.x + dir0.x * tr, pcam.y + dir0.y * tr, pcam.z + dir0.z * tr);
    } else {         // along the normal (vds.cu:1248-1251)
      norm_dir = normalize3(mk3(normals[3 * (size_t) i], normals[3 * (size_t) i + 1], normals[3 * (size_t) i + 2]));
      const float a = dmin - range, b = dmax - range;
      pc_min = mk3(pcam.x + norm_dir.x * a, pcam.y + norm_dir.y * a, pcam.z + norm_dir.z * a);
      pc_max = mk3(pcam.x + norm_dir.x * b, pcam.y + norm_dir.y * b, pcam.z + norm_dir.z * b);
    }
    const f3 pw_min = se3_apply(c.R, c.t, pc_min);
    const f3 pw_max = se3_apply(c.R, c.t, pc_max);
    // voxel-level DDA (vds.cu:1257-1296)
    const f3 dir = normalize3(mk3(pw_max.x - pw_min.x, pw_max.y - pw_min.y, pw_max.z - pw_min.z));
    i3 cur = world_to_voxel(m.vs, pw_min);
    const i3 end = world_to_voxel(m.vs, pw_max);
    const f3 step = mk3((float) signi(dir.x), (float) signi(dir.y), (float) signi(dir.z));
    const i3 istep = mki3(signi(dir.x), signi(dir.y), signi(dir.z));
    const i3 nb = mki3(cur.x + f2i(clampf(step.x, 0.0f, 1.f)), cur.y + f2i(clampf(step.y, 0.0f, 1.f)), cur.z + f2i(clampf(step.z, 0.0f, 1.f)));
    const f3 bw = voxel_to_world(m.vs, nb);
    const f3 boundary = mk3(bw.x - 0.5f * m.vs, bw.y - 0.5f * m.vs, bw.z - 0.5f * m.vs);
    f3 t_max = mk3((boundary.x - pw_min.x) / dir.x, (boundary.y - pw_min.y) / dir.y, (boundary.z - pw_min.z) / dir.z);
    f3 t_delta = mk3((step.x * m.vs) / dir.x, (step.y * m.vs) / dir.y, (step.z * m.vs) / dir.z);
    const i3 bound = mki3(f2i((float) end.x + step.x), f2i((float) end.y + step.y), f2i((float) end.z + step.z));
    const bool gx = (fabsf(dir.x) < kFloatEps) || (fabsf(boundary.x - dir.x) < kFloatEps);
    const bool gy = (fabsf(dir.y) < kFloatEps) || (fabsf(boundary.y - dir.y) < kFloatEps);
    const bool gz = (fabsf(dir.z) < kFloatEps) || (fabsf(boundary.z - dir.z) < kFloatEps);
    t_max.x = gx ? kFltMax : t_max.x; t_delta.x = gx ? kFltMax : t_delta.x;
    t_max.y = gy ? kFltMax : t_max.y; t_delta.y = gy ? kFltMax : t_delta.y;
    t_max.z = gz ? kFltMax : t_max.z; t_delta.z = gz ? kFltMax : t_delta.z;
    // consecutive voxels of a beam mostly lie in one block: the last lookup is kept
    i3 last_block = mki3(0x7FFFFFFF, 0, 0);
    u32 last_val = kValNone;
#pragma unroll 1
    for (u32 iter = 0; iter < kMaxDdaIter; iter++) {
      const i3 block = voxel_to_block_fast(m, cur);
      u32 val = last_val;
      if (block.x != last_block.x || block.y != last_block.y || block.z != last_block.z) {
        u64 bkey;
        int slot = -1;
        if (pack_key(block, bkey)) slot = hash_find(t, bkey);
        val = slot >= 0 ? t.vals[slot] : kValNone;
        last_block = block;
        last_val = val;
      }
      if (val != kValNone) {
        const int res = (val & kValCoarseBit) ? 1 : 0;
        const int scale = 1 << res;
        // vds.cu:1303-1309: the voxel the SDF is measured at — on a coarse block the fine coordinate divided by 2 with C's
        // truncation toward zero, times the coarse voxel size (kept literally, negative coordinates included)
        const i3 aprox = mki3(cur.x / scale, cur.y / scale, cur.z / scale);
        const f3 pc = se3_apply(c.Ri, c.ti, voxel_to_world(m.vs * (float) scale, aprox));
        float sdf;
        if (!normals) sdf = range - norm3(pc);
        else sdf = ((pc.x - pcam.x) * norm_dir.x + (pc.y - pcam.y) * norm_dir.y) + (pc.z - pcam.z) * norm_dir.z;
        if (sdf <= -tr) break;
        if (sdf >= 0.f) sdf = fminf(tr, sdf);
        else sdf = fmaxf(-tr, sdf);
        rec(val, res, voxel_local_index(cur, res), sdf);
      }
      const bool ax = t_max.x < t_max.y && t_max.x < t_max.z;
      const bool az = !ax && (t_max.z < t_max.y);
      const bool ay = !ax && !az;
      cur.x += ax ? istep.x : 0; cur.y += ay ? istep.y : 0; cur.z += az ? istep.z : 0;
      if ((ax && cur.x == bound.x) || (ay && cur.y == bound.y) || (az && cur.z == bound.z)) break;
      t_max.x = ax ? t_max.x + t_delta.x : t_max.x;
      t_max.y = ay ? t_max.y + t_delta.y : t_max.y;
      t_max.z = az ? t_max.z + t_delta.z : t_max.z;
    }
  }
}

template <bool EMIT, typename K>
__global__ __launch_bounds__(256) void k_points_walk(const Cam c, const Map m, const Tab t, const float* __restrict__ pts,
                                                     const float* __restrict__ normals, const u32 n, u32* __restrict__ counts,
                                                     const ScanState ss, K* __restrict__ keys, float* __restrict__ vals,
                                                     const int coarse_bit, const u32 rec_cap) {
  // rec_cap: records the buffers hold.  The host sizes them by a bound on the voxels a beam can cross, but the bound is derived,
  // not enforced by the walk (kMaxDdaIter is its only limit): a record beyond the capacity is not written, the host sees the
  // total in the scan's report and fails the call cleanly
  __shared__ u32 s_part[4];
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  u32 cnt = 0;
  u32 out = 0;
  if (EMIT) {
    // records of the points before this one: the earlier workgroups' totals + the earlier points of this workgroup
    u32 before = 0;
    for (u32 j = threadIdx.x; j < blockIdx.x; j += 256) before += ss.wg_totals[j];
    const u32 base = wg_sum_256(before, s_part);
    const u32 mine = live ? counts[i] : 0u;
    u32 incl = mine;
    const u32 lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 o = __shfl_up(incl, off);
      if ((int) lane >= off) incl += o;
    }
    if (lane == 63) s_part[threadIdx.x >> 6] = incl;
    __syncthreads();
    u32 woff = 0;
    for (u32 w = 0; w < (threadIdx.x >> 6); w++) woff += s_part[w];
    out = base + woff + incl - mine;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
      // the scan's one report: the last workgroup knows the grand total before it starts its own walk; the host reads it (and
      // sizes the sort) while the emit pass runs
      ss.host_rec[0] = (u32) t.ctr[CTR_HWM_FINE];
      ss.host_rec[1] = out + mine;
      ss.host_rec[2] = 0;
      __threadfence_system();
      __hip_atomic_store(&ss.host_rec[3], ss.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  walk_beam(c, m, t, pts, normals, i, live, [&](const u32 val, const int res, const u32 li, const float sdf) {
    if (EMIT) {
      u64 vid;
      if (res) { const u32 u = val & ~kValCoarseBit; vid = ((u64) (u >> 3) * 512u + (u64) (u & 7u) * 64u + li) | (1ull << coarse_bit); }
      else vid = (u64) val * 512u + li;
      if (out + cnt < rec_cap) {
        keys[out + cnt] = (K) vid;
        vals[out + cnt] = sdf;
      }
    }
    cnt++;
  });
  if (!EMIT) {
    if (live) counts[i] = cnt;
    const u32 total = wg_sum_256(cnt, s_part);
    if (threadIdx.x == 0) ss.wg_totals[blockIdx.x] = total;
  }
}

// Sorted records -> voxels.  A workgroup stages kApplyChunk consecutive records (keys + values) in LDS, lists the heads of the
// runs that begin in its chunk (key differs from the predecessor's), and hands them out one per lane: the lane loads the voxel,
// folds the run in order — from LDS, from global memory once the run leaves the chunk — and stores the voxel.  The fold is a
// serial chain by nature (a running mean in fp32), so what the kernel can do is keep an iteration short: the sdf update (one
// division, through the refined reciprocal of the weight sum), the weight, and the three colour channels as integers — u8(0.5 c + 0.5 * 0 + 0.5) of combineVoxel (vhu.cuh:170-176)
// is (c + 1) >> 1 for every c in 0 .. 255; the variance term delta * delta2 (vds.cu:1352-1366) is overwritten by every
// update, so only the LAST record of a run computes it.
constexpr int kApplyChunk = 1024;
template <typename K>
__global__ __launch_bounds__(256) void k_points_apply(const Map m, const Tab t, const K* __restrict__ keys, const float* __restrict__ vals,
                                                      const u32 n_rec, const int coarse_bit, const int count_updates) {
  __shared__ K s_key[kApplyChunk];
  __shared__ float s_val[kApplyChunk];
  __shared__ unsigned short s_head[kApplyChunk];
  __shared__ u32 s_nhead;
  const u32 c0 = blockIdx.x * kApplyChunk;
  if (c0 >= n_rec) return;
  const u32 cn = min((u32) kApplyChunk, n_rec - c0);
  for (u32 j = threadIdx.x; j < cn; j += 256) { s_key[j] = keys[c0 + j]; s_val[j] = vals[c0 + j]; }
  __syncthreads();
  // heads of the runs that begin in this chunk, compacted IN ORDER (round r, wave w, lane = ascending record index): the run of
  // head h then ends where head h + 1 begins — a lane knows its trip count and reads no key while it folds
  constexpr int kRounds = kApplyChunk / 256;
  __shared__ u32 s_cnt[kRounds * 4];
  const u32 wave = threadIdx.x >> 6;
  u32 hflag = 0, hpre[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; r++) {
    const u32 j = (u32) r * 256 + threadIdx.x;
    bool head = false;
    if (j < cn) {
      const K key = s_key[j];
      head = !(j > 0 ? s_key[j - 1] == key : (c0 > 0 && keys[c0 - 1] == key));
    }
    const u64 ballot = __ballot(head);
    hpre[r] = (u32) __popcll(ballot & lanemask_lt());
    hflag |= head ? (1u << r) : 0u;
    if ((threadIdx.x & 63) == 0) s_cnt[r * 4 + wave] = (u32) __popcll(ballot);
  }
  __syncthreads();
  {
    u32 run = 0;
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
#pragma unroll
      for (int w = 0; w < 4; w++) {
        if ((u32) w == wave && ((hflag >> r) & 1u)) s_head[run + hpre[r]] = (unsigned short) (r * 256 + threadIdx.x);
        run += s_cnt[r * 4 + w];
      }
    }
    if (threadIdx.x == 0) s_nhead = run;
  }
  __syncthreads();
  const u32 nhead = s_nhead;
  const u32 w1 = (u32) (m.weight_sample & 0xFF), wmax = (u32) (m.weight_max & 0xFF);
  const float half_vs = m.vs / 2;
  const u64 cmask = 1ull << coarse_bit;
  for (u32 h = threadIdx.x; h < nhead; h += 256) {
    const u32 j = s_head[h];
    const K key = s_key[j];
    const u64 vid = (u64) key;
    const u64 id = vid & (cmask - 1);
    const u32 H = (u32) (id >> 9);
    char* base = t.pool + (size_t) H * kFineBytes;
    float *p_sdf, *p_ss;
    u32* p_rgbw;
    if (vid & cmask) {
      const u32 k = (u32) (id >> 6) & 7u, li = (u32) (id & 63u);
      base += (size_t) k * kCoarseBytes;
      p_sdf = (float*) base + li; p_ss = (float*) (base + 256) + li; p_rgbw = (u32*) (base + 512) + li;
    } else {
      const u32 li = (u32) (id & 511u);
      p_sdf = (float*) base + li; p_ss = (float*) (base + 2048) + li; p_rgbw = (u32*) (base + 4096) + li;
    }
    float s0 = *p_sdf;
    const u32 rgbw0 = *p_rgbw;
    u32 w0 = rgbw0 >> 24, r0 = rgbw0 & 0xFF, g0 = (rgbw0 >> 8) & 0xFF, b0 = (rgbw0 >> 16) & 0xFF;
    // combineVoxel with curr = {sdf, weight_update, rgb (0, 0, 0)} (vhu.cuh:167-181), one record.  What a wave pays for is the
    // LONGEST of its 64 runs (mean 6 records, one in a hundred above 50, the voxels next to the sensor ~200) times the
    // latency of one step, so the step carries nothing that can wait for the end of the run: the colours — (c + 1) >> 1 per record
    // = ceil(c / 2), n times = ceil(c / 2^n) — and the state before the LAST update (the variance term's) are settled after the
    // loop (a record is folded once its successor is known to belong to the run), and no step waits for memory.
    // One step, s0 <- (s0 w0 + sdf w1) / (w0 + w1), is a chain of five dependent operations on s0 (multiply, add, and the three of
    // div_cr); everything else of the step — the weight, its conversion, the reciprocal of the NEXT weight sum (rcp_refined: the
    // values of the table, no memory access), the next record — is computed one step ahead, off that chain.
    const float w1f = (float) w1;
    u32 wsum = w0 + w1;
    float w0f = (float) w0, den = (float) (int) wsum, rden = rcp_refined(den);
    auto fold = [&](const float sdf) {
      const float num = s0 * w0f + sdf * w1f;
      // the state of the step after this one (independent of s0)
      const u32 w0n = wsum < wmax ? wsum : wmax, wsum_n = w0n + w1;
      const float w0f_n = (float) w0n, den_n = (float) (int) wsum_n, rden_n = rcp_refined(den_n);
      s0 = m.wsum_two_steps ? num / den : div_cr(num, den, rden);
      w0 = w0n; wsum = wsum_n; w0f = w0f_n; den = den_n; rden = rden_n;
    };
    float pend = s_val[j];  // the head's own record
    const u32 end = h + 1 < nhead ? (u32) s_head[h + 1] : cn;  // the chunk's last run may go on in the next chunk
    u32 nfold = end - j;
    {
      u32 i = j + 1;
      for (; i + 4 <= end; i += 4) {  // four records fetched together, folded one after the other
        const float v0 = s_val[i], v1 = s_val[i + 1], v2 = s_val[i + 2], v3 = s_val[i + 3];
        fold(pend); fold(v0); fold(v1); fold(v2);
        pend = v3;
      }
      for (; i < end; i++) {
        fold(pend);
        pend = s_val[i];
      }
    }
    if (h + 1 == nhead && c0 + cn < n_rec) {  // one lane per workgroup: the rest of its run, from global memory, by key
      bool more = true;
      for (u32 k = c0 + cn; more; k += 4) {
        K kk[4];
        float vv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const u32 idx = k + q;
          if (idx < n_rec) { kk[q] = keys[idx]; vv[q] = vals[idx]; }
          else { kk[q] = ~key; vv[q] = 0.f; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (!more || kk[q] != key) { more = false; continue; }
          fold(pend);
          pend = vv[q];
          nfold++;
        }
      }
    }
    const float s_prev = s0, sdf_last = pend;  // state BEFORE the last update, and the last record's sdf: the variance term needs them
    const u32 w_prev = w0;
    fold(pend);
    {
      const u32 sh = nfold < 8u ? nfold : 8u, add = (1u << sh) - 1u;  // c <= 255: eight halvings leave 1 (or 0 for c == 0), as do more
      r0 = (r0 + add) >> sh; g0 = (g0 + add) >> sh; b0 = (b0 + add) >> sh;
    }
    // vds.cu:1352-1366 for the last update: delta against the mean before it (0 for a voxel without weight), delta2 against the mean after
    const float curr_mean = w_prev > 0 ? s_prev : 0.f;
    const float delta = (sdf_last - curr_mean) / half_vs;
    const float delta2 = (sdf_last - s0) / half_vs;
    *p_sdf = s0;
    *p_ss = 0.f + delta * delta2;
    *p_rgbw = r0 | (g0 << 8) | (b0 << 16) | (w0 << 24);
  }
  if (count_updates && threadIdx.x == 0 && nhead) atomicAdd(&t.prof[PROF_UPDATED], (u64) nhead);  // profile mode: voxels this scan updated
}

}  // namespace mrh
