// mrh_kernels.h — the per-frame fusion kernels (gfx950, wave64).
//
// One frame = k_alloc -> k_compact -> k_integrate [-> variance stage] [-> starve] -> k_gc_identify -> k_gc_free,
// all enqueued on one stream with NO host round trip (every count lives in Tab::ctr and is consumed by
// grid-stride loops).  The reference needs >= 4 device->host scalar reads and >= 10 cudaDeviceSynchronize
// per frame for the same work.
#pragma once

#include "mrh_device.h"

namespace mrh {

constexpr int kWave = 64;

__device__ __forceinline__ u32 lane_id() { return __lane_id(); }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// =====================================================================================================
// K1  block allocation along pixel rays                        (reference: allocBlocksKernel vds.cu:758-857
//     + allocBlock :502-624 + host retry loop :874-922)
//
// One workgroup = one 16x16 pixel tile.  Every pixel walks its ray segment [d-t, d+t] with the
// reference's DDA, but instead of probing the global table at every step (thousands of threads fighting
// over the same few buckets) the visited block keys are first de-duplicated in an LDS hash set: a tile at
// 2 m depth touches ~10-30 distinct blocks for ~900 visits.  Only distinct keys then pay the 8-corner
// frustum test and the global lock-free insert; heap pops are wave-aggregated (one atomic per wave).
// =====================================================================================================

constexpr int kTile = 16;
constexpr int kSetCap = 1024;   // LDS key set capacity (8 KiB)
constexpr int kSetProbe = 24;

__device__ __forceinline__ void alloc_commit(const Tab& t, const Map& m, bool won, int slot, i3 b) {
  // wave-aggregated pop of the fine free list
  const u64 ballot = __ballot(won);
  if (ballot == 0) return;
  const int n = __popcll(ballot);
  const int leader = __ffsll((long long) ballot) - 1;
  int base = 0;
  if ((int) lane_id() == leader) base = atomicSub(&t.ctr[CTR_HEAP_FINE], n);
  base = __shfl(base, leader);
  if (!won) return;
  const int idx = base - __popcll(ballot & lanemask_lt());
  if (idx < 0) {
    publish_without_storage(t, slot, CTR_HEAP_FINE);  // pool exhausted (vds.cu:566-569 prints and skips the block)
    return;
  }
  const u32 H = t.heap_fine[idx];
  t.vals[slot] = H;
  t.desc_fine[H] = make_int4(b.x, b.y, b.z, 1);
  atomicMax(&t.ctr[CTR_HWM_FINE], (int) H + 1);
}

template <bool PROFILE>
__global__ __launch_bounds__(256) void k_alloc(const Cam c, const Map m, const Tab t, const float* __restrict__ depth) {
  __shared__ u64 set[kSetCap];
  __shared__ u32 s_inserted;
  const int tid = threadIdx.y * kTile + threadIdx.x;
  for (int i = tid; i < kSetCap; i += 256) set[i] = kKeyEmpty;
  if (tid == 0) {
    s_inserted = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0) t.ctr[CTR_COMPACT] = 0;  // consumed by k_compact (next kernel)
  }
  __syncthreads();

  const int row = blockIdx.y * kTile + threadIdx.y;
  const int col = blockIdx.x * kTile + threadIdx.x;
  u32 my_inserted = 0;

  // what to do with one visited block
  auto visit_global = [&](i3 b, u64 key, bool active) {
    bool won = false;
    int slot = -1;
    if (active && block_in_frustum_approx_m(c, m.vs, b)) {
      slot = hash_insert(t, key);
      if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
      won = slot >= 0;
    }
    alloc_commit(t, m, won, slot, b);
    if (PROFILE && won) my_inserted++;
  };

  bool in_img = row < c.rows && col < c.cols;
  float d = 0.f;
  if (in_img) {
    d = depth[(size_t) row * c.cols + col];
    // camera.cu:13-18: cloud stays 0 outside (min_depth, max_depth]; pinhole cloud.z == d.  Spherical: `depth` is the
    // pre-computed getDepth(cloud) image (k_cloud_depth), already 0 where the raw depth is out of range
    if (c.model == 0 && (d <= c.min_depth || d > c.max_depth)) d = 0.f;
  }
  // vds.cu:771-781
  const float tr = get_truncation(d, m.trunc, m.trunc_scale);
  const float dmin = fminf(c.max_int_dist, d - tr);
  const float dmax = fminf(c.max_int_dist, d + tr);
  const bool walk = in_img && !(d == 0.f) && !(dmin >= dmax);
  if (walk) {
    const f3 pw_min = se3_apply(c.R, c.t, inverse_projection_m(c, (u32) row, (u32) col, dmin));
    const f3 pw_max = se3_apply(c.R, c.t, inverse_projection_m(c, (u32) row, (u32) col, dmax));
    const f3 dd = mk3(pw_max.x - pw_min.x, pw_max.y - pw_min.y, pw_max.z - pw_min.z);
    const float inv_len = 1.0f / sqrtf(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z);  // normalize, cuda_math.cuh:1075-1078
    const f3 dir = mk3(dd.x * inv_len, dd.y * inv_len, dd.z * inv_len);
    i3 cur = world_to_block(m.vs, pw_min);
    const i3 end = world_to_block(m.vs, pw_max);
    const f3 step = mk3((float) signi(dir.x), (float) signi(dir.y), (float) signi(dir.z));
    const i3 nb = mki3(cur.x + f2i(clampf(step.x, 0.0f, 1.f)), cur.y + f2i(clampf(step.y, 0.0f, 1.f)), cur.z + f2i(clampf(step.z, 0.0f, 1.f)));
    const f3 bw = voxel_to_world(m.vs, mki3(nb.x * kBlockSide, nb.y * kBlockSide, nb.z * kBlockSide));
    const f3 boundary = mk3(bw.x - 0.5f * m.vs, bw.y - 0.5f * m.vs, bw.z - 0.5f * m.vs);
    f3 t_max = mk3((boundary.x - pw_min.x) / dir.x, (boundary.y - pw_min.y) / dir.y, (boundary.z - pw_min.z) / dir.z);
    f3 t_delta = mk3((step.x * (float) kBlockSide * m.vs) / dir.x, (step.y * (float) kBlockSide * m.vs) / dir.y, (step.z * (float) kBlockSide * m.vs) / dir.z);
    const i3 bound = mki3(f2i((float) end.x + step.x), f2i((float) end.y + step.y), f2i((float) end.z + step.z));
    const float fmx = 3.402823466e+38f;
    // vds.cu:801-827 (second test of each pair compares a position with a direction; kept literally)
    if (fabsf(dir.x) < kFloatEps) { t_max.x = fmx; t_delta.x = fmx; }
    if (fabsf(boundary.x - dir.x) < kFloatEps) { t_max.x = fmx; t_delta.x = fmx; }
    if (fabsf(dir.y) < kFloatEps) { t_max.y = fmx; t_delta.y = fmx; }
    if (fabsf(boundary.y - dir.y) < kFloatEps) { t_max.y = fmx; t_delta.y = fmx; }
    if (fabsf(dir.z) < kFloatEps) { t_max.z = fmx; t_delta.z = fmx; }
    if (fabsf(boundary.z - dir.z) < kFloatEps) { t_max.z = fmx; t_delta.z = fmx; }

    u64 last_key = kKeyEmpty;
#pragma unroll 1
    for (u32 iter = 0; iter < kMaxDdaIter; iter++) {
      u64 key;
      if (!pack_key(cur, key)) {
        atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_RANGE);
      } else if (key != last_key && owns_block(m, cur)) {
        last_key = key;
        // LDS set insert
        u32 s = hash_key(key) & (kSetCap - 1);
        bool placed = false;
#pragma unroll 1
        for (int p = 0; p < kSetProbe; p++) {
          const u64 old = atomicCAS(&set[s], kKeyEmpty, key);
          if (old == kKeyEmpty || old == key) { placed = true; break; }
          s = (s + 1) & (kSetCap - 1);
        }
        if (!placed) {
          // set saturated (far, sparse rays): go to the global table directly, un-aggregated
          if (block_in_frustum_approx_m(c, m.vs, cur)) {
            const int slot = hash_insert(t, key);
            if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
            if (slot >= 0) {
              const int idx = atomicSub(&t.ctr[CTR_HEAP_FINE], 1);
              if (idx < 0) {
                publish_without_storage(t, slot, CTR_HEAP_FINE);
              } else {
                const u32 H = t.heap_fine[idx];
                t.vals[slot] = H;
                t.desc_fine[H] = make_int4(cur.x, cur.y, cur.z, 1);
                atomicMax(&t.ctr[CTR_HWM_FINE], (int) H + 1);
                if (PROFILE) my_inserted++;
              }
            }
          }
        }
      }
      // vds.cu:836-853
      if (t_max.x < t_max.y && t_max.x < t_max.z) {
        cur.x = f2i((float) cur.x + step.x);
        if (cur.x == bound.x) break;
        t_max.x += t_delta.x;
      } else if (t_max.z < t_max.y) {
        cur.z = f2i((float) cur.z + step.z);
        if (cur.z == bound.z) break;
        t_max.z += t_delta.z;
      } else {
        cur.y = f2i((float) cur.y + step.y);
        if (cur.y == bound.y) break;
        t_max.y += t_delta.y;
      }
    }
  }
  __syncthreads();
  // distinct keys of this tile -> frustum test + global insert (all 256 threads take part in the ballots)
#pragma unroll 1
  for (int i = tid; i < kSetCap; i += 256) {
    const u64 key = set[i];
    const bool active = key != kKeyEmpty;
    const i3 b = active ? unpack_key(key) : mki3(0, 0, 0);
    visit_global(b, key, active);
  }
  if (PROFILE) {
    if (my_inserted) atomicAdd(&s_inserted, my_inserted);
    __syncthreads();
    if (tid == 0 && s_inserted) atomicAdd(&t.prof[PROF_INSERTED], (u64) s_inserted);
  }
}

// spherical camera: getDepth(cloud) per pixel (camera.cu:5-19 + camera.cuh:120-129) = the norm of the back-projected
// point, 0 where the raw depth is outside (min_depth, max_depth] — what every later kernel of the frame reads as "depth"
__global__ __launch_bounds__(256) void k_cloud_depth(const Cam c, const float* __restrict__ depth, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c.rows * c.cols) return;
  const float d = depth[i];
  float r = 0.f;
  if (!(d <= c.min_depth || d > c.max_depth)) r = get_depth(c, inverse_projection_m(c, (u32) (i / c.cols), (u32) (i % c.cols), d));
  out[i] = r;
}

// =====================================================================================================
// K2  frustum compaction of the live blocks   (reference: resetCompactHashTableKernel vds.cu:9-14 +
//     flatAndReduceHashTableKernel :406-434/:452-480, both O(hash slots) = 2 x 24 B x 10 x buckets per frame)
//
// Sweeps the dense block-descriptor array up to its high-water mark (O(blocks ever live), 16 B each),
// wave ballot + popcount prefix, one atomic per wave.
// =====================================================================================================
__global__ __launch_bounds__(256) void k_compact(const Cam c, const Map m, const Tab t, const int use_camera) {
  const int hwm = t.ctr[CTR_HWM_FINE];
  const int total = t.multi_res ? hwm * 9 : hwm;
  for (int base = blockIdx.x * 256; base < total; base += gridDim.x * 256) {
    const int i = base + threadIdx.x;
    bool keep = false;
    int4 d = make_int4(0, 0, 0, 0);
    u32 val = 0;
    if (i < total) {
      if (i < hwm) { d = t.desc_fine[i]; val = (u32) i; }
      else { const u32 u = (u32) (i - hwm); d = t.desc_coarse[u]; val = u | kValCoarseBit; }
      if (d.w & 1) keep = !use_camera || block_in_frustum_approx_m(c, m.vs, mki3(d.x, d.y, d.z));  // bit 0 = live (upper bits: insertion stamp)
    }
    const u64 ballot = __ballot(keep);
    if (ballot) {
      const int leader = __ffsll((long long) ballot) - 1;
      int wbase = 0;
      if ((int) lane_id() == leader) wbase = atomicAdd(&t.ctr[CTR_COMPACT], __popcll(ballot));
      wbase = __shfl(wbase, leader);
      if (keep) t.compact[wbase + __popcll(ballot & lanemask_lt())] = make_int4(d.x, d.y, d.z, (int) val);
    }
  }
}

// =====================================================================================================
// K3  depth -> TSDF integration            (reference: integrateDepthMapKernel vds.cu:1095-1181,
//     combineVoxel vhu.cuh:167-181, projectPoint camera.cuh:131-147)
//
// One workgroup per block, lane = voxel: the 512 voxels of a block are 3 contiguous 2-KiB planes, so a
// wave reads/writes 256 contiguous bytes per plane (the reference maps threadIdx.x -> entry and
// threadIdx.y -> voxel, so a warp touches 16 blocks x 2 voxels: 12-B accesses 6 KiB apart).
// VARIANCE = false restates reintegrateDepthMapKernel (vds.cu:1942-2018): same body, no sum_squared term.
// =====================================================================================================

struct VoxUpdate {
  bool ok;
  float sdf_new, sumsq_new;
  u32 rgbw_new;
};

// the per-voxel body shared by integrate / reintegrate
template <bool VARIANCE>
__device__ __forceinline__ bool integrate_voxel(const Cam& c, const Map& m, const float* __restrict__ depth,
                                                const uint8_t* __restrict__ rgb, i3 pi, float* p_sdf, float* p_sumsq, u32* p_rgbw) {
  const f3 pf = voxel_to_world(m.vs, pi);
  const f3 pcam = se3_apply(c.Ri, c.ti, pf);
  int row, col;
  if (!project_point_m<false>(c, pcam, row, col)) return false;
  float d = depth[(size_t) row * c.cols + col];
  if (c.model == 0 && (d <= c.min_depth || d > c.max_depth)) d = 0.f;  // camera.cu:13-18 (cloud z); spherical: see k_cloud_depth
  if (d == 0.f || d > c.max_int_dist) return false;
  float sdf = d - get_depth(c, pcam);
  const float truncation = get_truncation(d, m.trunc, m.trunc_scale);
  if (sdf <= -truncation) return false;
  if (sdf >= 0.f) sdf = fminf(truncation, sdf);
  else sdf = fmaxf(-truncation, sdf);

  const u32 w1 = (u32) (m.weight_sample & 0xFF);
  const uint8_t* px = rgb + ((size_t) row * c.cols + col) * 3;
  const u32 r1 = px[0], g1 = px[1], b1 = px[2];

  const float s0 = *p_sdf;
  const u32 old = *p_rgbw;
  const u32 w0 = old >> 24;
  u32 r0 = old & 0xFF, g0 = (old >> 8) & 0xFF, b0 = (old >> 16) & 0xFF;
  const float curr_mean = (w0 > 0) ? s0 : sdf;
  const float delta = (sdf - curr_mean) / (m.vs / 2);
  if (w0 == 0) { r0 = r1; g0 = g1; b0 = b1; }
  const u32 rn = (u32) f2i((0.5f * (float) r0 + 0.5f * (float) r1) + 0.5f) & 0xFF;
  const u32 gn = (u32) f2i((0.5f * (float) g0 + 0.5f * (float) g1) + 0.5f) & 0xFF;
  const u32 bn = (u32) f2i((0.5f * (float) b0 + 0.5f * (float) b1) + 0.5f) & 0xFF;
  const float sn = (s0 * (float) w0 + sdf * (float) w1) / (float) (int) (w0 + w1);
  const u32 wmax = (u32) (m.weight_max & 0xFF);
  const u32 wn = (w0 + w1) < wmax ? (w0 + w1) : wmax;
  *p_sdf = sn;
  *p_rgbw = rn | (gn << 8) | (bn << 16) | (wn << 24);
  if (VARIANCE) {
    const float delta2 = (sdf - sn) / (m.vs / 2);
    *p_sumsq = 0.f + delta * delta2;  // whole-voxel store of a default Voxel then atomicAdd (vds.cu:1178-1180)
  } else {
    *p_sumsq = 0.f;
  }
  return true;
}

template <bool PROFILE>
__global__ __launch_bounds__(512) void k_integrate(const Cam c, const Map m, const Tab t, const float* __restrict__ depth,
                                                   const uint8_t* __restrict__ rgb, u64* __restrict__ updated_partials) {
  const int count = t.ctr[CTR_COMPACT];
  const int v = threadIdx.x;
  u32 my_updates = 0;
  for (int e = blockIdx.x; e < count; e += gridDim.x) {
    const int4 ent = t.compact[e];
    const u32 val = (u32) ent.w;
    const bool coarse = (val & kValCoarseBit) != 0;
    if (coarse && v >= kCoarseVoxels) continue;
    const VoxPtr vp = vox_ptr(t, val);
    i3 pi;
    if (!coarse) pi = mki3(ent.x * kBlockSide + (v & 7), ent.y * kBlockSide + ((v >> 3) & 7), ent.z * kBlockSide + (v >> 6));
    else pi = mki3(ent.x * kBlockSide + 2 * (v & 3), ent.y * kBlockSide + 2 * ((v >> 2) & 3), ent.z * kBlockSide + 2 * (v >> 4));
    const bool ok = integrate_voxel<true>(c, m, depth, rgb, pi, vp.sdf + v, vp.sumsq + v, vp.rgbw + v);
    if (PROFILE) my_updates += ok ? 1u : 0u;
  }
  if (PROFILE) {
    __shared__ u32 s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const u64 b = __ballot(true);
    (void) b;
    // wave reduce then one LDS atomic per wave
    u32 w = my_updates;
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off);
    if (lane_id() == 0 && w) atomicAdd(&s_sum, w);
    __syncthreads();
    if (threadIdx.x == 0) {
      updated_partials[blockIdx.x] += (u64) s_sum;  // private slot, no atomics
      if (blockIdx.x == 0) t.prof[PROF_COMPACT] += (u64) count;
    }
  }
}

// =====================================================================================================
// K4  garbage collection                     (reference: garbageCollectIdentifyKernel vds.cu:1674-1713,
//     garbageCollectFreeKernel :1827-1844, deleteHashEntryElement :1727-1824)
// identify: workgroup per block, wave64 shuffle reduction of min |sdf| (weighted voxels) and max weight.
// free: thread per compact entry; heap pushes wave-aggregated; the wave then zeroes each freed block
// cooperatively with 16-byte stores (the reference zeroes 6 KiB in a single-thread loop).
// =====================================================================================================
__global__ __launch_bounds__(512) void k_gc_identify(const Tab t, const float trunc_threshold, u32* __restrict__ decision) {
  __shared__ float s_min[8];
  __shared__ u32 s_max[8];
  const int count = t.ctr[CTR_COMPACT];
  const int v = threadIdx.x;
  const int wave = v >> 6;
  for (int e = blockIdx.x; e < count; e += gridDim.x) {
    const int4 ent = t.compact[e];
    const u32 val = (u32) ent.w;
    const bool coarse = (val & kValCoarseBit) != 0;
    const VoxPtr vp = vox_ptr(t, val);
    float mn = 3.402823466e+38f;
    u32 mx = 0;
    if (!coarse || v < kCoarseVoxels) {
      const u32 w = vp.rgbw[v] >> 24;
      if (w != 0) mn = fabsf(vp.sdf[v]);
      mx = w;
    }
    for (int off = 32; off > 0; off >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, off));
      const u32 o = __shfl_xor(mx, off);
      mx = o > mx ? o : mx;
    }
    if (lane_id() == 0) { s_min[wave] = mn; s_max[wave] = mx; }
    __syncthreads();
    if (v == 0) {
      for (int i = 1; i < 8; i++) { mn = fminf(mn, s_min[i]); mx = s_max[i] > mx ? s_max[i] : mx; }
      decision[e] = (mn >= trunc_threshold || mx == 0) ? 1u : 0u;
    }
    __syncthreads();
  }
}

// removes one block from the table and returns its storage.  Single writer per key.
__device__ __forceinline__ bool hash_erase(const Tab& t, u64 key) {
  const int s = hash_find(t, key);
  if (s < 0) return false;
  t.keys[s] = kKeyTomb;
  return true;
}

template <bool PROFILE>
__global__ __launch_bounds__(256) void k_gc_free(const Tab t, const u32* __restrict__ decision) {
  const int count = t.ctr[CTR_COMPACT];
  for (int base = blockIdx.x * 256; base < count; base += gridDim.x * 256) {
    const int e = base + threadIdx.x;
    bool fr = false;
    int4 ent = make_int4(0, 0, 0, 0);
    if (e < count && decision[e] != 0) { ent = t.compact[e]; fr = true; }
    u32 val = (u32) ent.w;
    if (fr) {
      u64 key;
      pack_key(mki3(ent.x, ent.y, ent.z), key);
      fr = hash_erase(t, key);
    }
    const bool coarse = (val & kValCoarseBit) != 0;
    // wave-aggregated pushes (fine and coarse lists separately)
    for (int pass = 0; pass < 2; pass++) {
      const bool mine = fr && (coarse == (pass == 1));
      const u64 ballot = __ballot(mine);
      if (!ballot) continue;
      const int n = __popcll(ballot);
      const int leader = __ffsll((long long) ballot) - 1;
      int* ctr = &t.ctr[pass == 0 ? CTR_HEAP_FINE : CTR_HEAP_COARSE];
      int b0 = 0;
      if ((int) lane_id() == leader) b0 = atomicAdd(ctr, n);
      b0 = __shfl(b0, leader);
      if (mine) {
        const int idx = b0 + 1 + __popcll(ballot & lanemask_lt());  // vds.cu:53-62: heap[old + 1] = ptr
        if (pass == 0) { t.heap_fine[idx] = val; t.desc_fine[val].w = 0; }
        else { const u32 u = val & ~kValCoarseBit; t.heap_coarse[idx] = u; t.desc_coarse[u].w = 0; }
      }
    }
    // cooperative zeroing: the wave walks its freed blocks, 16 B per lane per store
    u64 todo = __ballot(fr);
    if (PROFILE && todo && lane_id() == 0) atomicAdd(&t.prof[PROF_FREED], (u64) __popcll(todo));
    while (todo) {
      const int src = __ffsll((long long) todo) - 1;
      todo &= todo - 1;
      const u32 bval = __shfl(val, src);
      const uint4 z = make_uint4(0, 0, 0, 0);
      if (bval & kValCoarseBit) {
        const u32 u = bval & ~kValCoarseBit;
        uint4* p = (uint4*) (t.pool + (size_t) (u >> 3) * kFineBytes + (size_t) (u & 7) * kCoarseBytes);
        if (lane_id() < kCoarseBytes / 16) p[lane_id()] = z;
      } else {
        uint4* p = (uint4*) (t.pool + (size_t) bval * kFineBytes);
#pragma unroll
        for (int k = 0; k < kFineBytes / 16 / kWave; k++) p[k * kWave + lane_id()] = z;
      }
    }
  }
}

// =====================================================================================================
// K5  starve                                     (reference: starveVoxelsKernel vds.cu:1597-1649)
// The reference packs (depth bits << 32) + thread id into one u64 atomicMin, so equal depths are broken by
// the race-ordered compact index.  Here the tie-break is the canonical (block position, voxel index): a
// 72-bit key split over two u64 min-buffers, three passes, no sort and no host round trip.
//   pass 0: zbuf0[pix] = min( depth_bits << 32 | key72 >> 40 )
//   pass 1: candidates equal to zbuf0: zbuf1[pix] = min( key72 & (2^40 - 1) )
//   pass 2: the unique winner decrements its weight.
// =====================================================================================================
template <int PASS>
__global__ __launch_bounds__(512) void k_starve(const Cam c, const Map m, const Tab t, u64* __restrict__ zbuf0, u64* __restrict__ zbuf1) {
  const int count = t.ctr[CTR_COMPACT];
  const int v = threadIdx.x;
  for (int e = blockIdx.x; e < count; e += gridDim.x) {
    const int4 ent = t.compact[e];
    const u32 val = (u32) ent.w;
    const bool coarse = (val & kValCoarseBit) != 0;
    if (coarse && v >= kCoarseVoxels) continue;  // oracle deviation D3
    // fine delinearisation for every block, as the reference does (vds.cu:1606-1607)
    const i3 pi = mki3(ent.x * kBlockSide + (v & 7), ent.y * kBlockSide + ((v >> 3) & 7), ent.z * kBlockSide + (v >> 6));
    const f3 pcam = se3_apply(c.Ri, c.ti, voxel_to_world(m.vs, pi));
    const float dep = get_depth(c, pcam);
    if (dep < c.min_depth) continue;
    int row, col;
    if (!project_point_m<false>(c, pcam, row, col)) continue;
    u64 key;
    pack_key(mki3(ent.x, ent.y, ent.z), key);
    const u64 hi = ((u64) __float_as_uint(dep) << 32) | (key >> 31);                 // top 32 bits of the 72-bit (key63<<9 | v)
    const u64 lo = ((key & 0x7FFFFFFFull) << 9) | (u64) v;                           // low 40 bits
    const size_t pix = (size_t) row * c.cols + col;
    if (PASS == 0) {
      atomicMin(&zbuf0[pix], hi);
    } else if (PASS == 1) {
      if (zbuf0[pix] == hi) atomicMin(&zbuf1[pix], lo);
    } else {
      if (zbuf0[pix] == hi && zbuf1[pix] == lo) {
        const VoxPtr vp = vox_ptr(t, val);
        const u32 old = vp.rgbw[v];
        const u32 w = old >> 24;
        vp.rgbw[v] = (old & 0x00FFFFFFu) | ((w > 0 ? w - 1 : 0) << 24);
      }
    }
  }
}

// =====================================================================================================
// K6-K8  variance-adaptive resolution       (reference: checkVarSDFKernel vds.cu:1857-1939,
//        allocateMemoryLow :860-871, reallocBlocksKernel :2021-2034 / reallocBlock :627-755,
//        reintegrateDepthMapKernel :1942-2018)
// =====================================================================================================

// decides on the device whether the coarse free list needs a refill (vds.cu:885-891 reads it on the host)
__global__ void k_refill_decide(const Tab t, const int low_blocks_to_allocate, int* __restrict__ flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *flag = (t.ctr[CTR_HEAP_COARSE] + 1 < low_blocks_to_allocate) ? 1 : 0;
    t.ctr[CTR_NREINT] = 0;  // reintegrate list (general path) / deferred coarse frees (fused path) of this frame
  }
}
// one thread per fine block converted: pops H from the fine list, pushes 8H+7 .. 8H (vds.cu:860-871)
__global__ __launch_bounds__(256) void k_refill(const Tab t, const int low_blocks_to_allocate, const int* __restrict__ flag) {
  if (*flag == 0) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= low_blocks_to_allocate) return;
  const int addr_high = atomicSub(&t.ctr[CTR_HEAP_FINE], 1);
  if (addr_high < 0) { atomicAdd(&t.ctr[CTR_HEAP_FINE], 1); atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_POOL); return; }
  const u32 H = t.heap_fine[addr_high];
  atomicMax(&t.ctr[CTR_HWM_FINE], (int) H + 1);
  const int addr_low = atomicAdd(&t.ctr[CTR_HEAP_COARSE], 8);
  for (int idx = 1; idx <= 8; idx++) t.heap_coarse[addr_low + idx] = H * 8 + 8 - idx;
}

// one 64-thread workgroup per fine compact block; thread t sums its 2x2x2 voxels, then the reference's
// shared-memory tree (stride 32..1) restated as wave shuffles: lane l += lane l+stride, identical order.
__global__ __launch_bounds__(64) void k_check_var(const Map m, const Tab t, int4* __restrict__ realloc_list) {
  const int count = t.ctr[CTR_COMPACT];
  const int tid = threadIdx.x;
  for (int e = blockIdx.x; e < count; e += gridDim.x) {
    const int4 ent = t.compact[e];
    const u32 val = (u32) ent.w;
    if (val & kValCoarseBit) continue;
    const VoxPtr vp = vox_ptr(t, val);
    float local_sum_sq = 0.f, local_weight = 0.f;
    const int gx = (tid % 4) * 2, gy = ((tid / 4) % 4) * 2, gz = (tid / 16) * 2;
    for (int dz = 0; dz < 2; ++dz)
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
          const int li = (gz + dz) * 64 + (gy + dy) * 8 + (gx + dx);
          const u32 w = vp.rgbw[li] >> 24;
          if (w > 0) { local_sum_sq += vp.sumsq[li]; local_weight += (float) w; }
        }
    for (int stride = 32; stride > 0; stride >>= 1) {
      const float os = __shfl_down(local_sum_sq, stride);
      const float ow = __shfl_down(local_weight, stride);
      if (tid < stride) { local_sum_sq += os; local_weight += ow; }
    }
    int coarsen = 0;
    if (tid == 0 && !(local_weight < 2)) {
      const double avg_var = (double) (local_sum_sq / (local_weight - 1));
      if ((local_weight - 1) > 1e-6f && avg_var > 0.f && avg_var < (double) m.var_threshold) coarsen = 1;
    }
    coarsen = __shfl(coarsen, 0);
    if (coarsen) {
      if (tid == 0) {
        u64 key;
        pack_key(mki3(ent.x, ent.y, ent.z), key);
        hash_erase(t, key);
        const int idx = atomicAdd(&t.ctr[CTR_HEAP_FINE], 1);
        t.heap_fine[idx + 1] = val;
        t.desc_fine[val].w = 0;
        const int r = atomicAdd(&t.ctr[CTR_NREALLOC], 1);
        realloc_list[r] = make_int4(ent.x, ent.y, ent.z, 1);
      }
      uint4* p = (uint4*) (t.pool + (size_t) val * kFineBytes);
      const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < kFineBytes / 16 / kWave; k++) p[k * kWave + tid] = z;
    }
  }
}

// re-inserts every coarsened position as a 4^3 block and records it for re-integration
__global__ __launch_bounds__(256) void k_realloc(const Tab t, const int4* __restrict__ realloc_list, int4* __restrict__ reint_list) {
  const int n = t.ctr[CTR_NREALLOC];
  if (blockIdx.x == 0 && threadIdx.x == 0) t.ctr[CTR_COMPACT] = 0;  // for the second k_compact of this frame
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int4 r = realloc_list[i];
    u64 key;
    pack_key(mki3(r.x, r.y, r.z), key);
    const int slot = hash_insert(t, key);
    if (slot < 0) { if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE); continue; }
    const int idx = atomicSub(&t.ctr[CTR_HEAP_COARSE], 1);
    if (idx < 0) {
      publish_without_storage(t, slot, CTR_HEAP_COARSE);
      continue;
    }
    const u32 u = t.heap_coarse[idx];
    t.vals[slot] = u | kValCoarseBit;
    t.desc_coarse[u] = make_int4(r.x, r.y, r.z, 1);
    const int q = atomicAdd(&t.ctr[CTR_NREINT], 1);
    reint_list[q] = make_int4(r.x, r.y, r.z, (int) (u | kValCoarseBit));
  }
}

// the reference launches this with blockDim = (16,1,1) and gridDim.y = 32 (vds.cu:2097): voxel index 0..31 only
__global__ __launch_bounds__(64) void k_reintegrate(const Cam c, const Map m, const Tab t, const float* __restrict__ depth,
                                                    const uint8_t* __restrict__ rgb, const int4* __restrict__ reint_list) {
  const int n = t.ctr[CTR_NREINT];
  const int v = threadIdx.x;
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    if (v >= 32) continue;
    const int4 ent = reint_list[e];
    const u32 val = (u32) ent.w;
    const VoxPtr vp = vox_ptr(t, val);
    const i3 pi = mki3(ent.x * kBlockSide + 2 * (v & 3), ent.y * kBlockSide + 2 * ((v >> 2) & 3), ent.z * kBlockSide + 2 * (v >> 4));
    integrate_voxel<false>(c, m, depth, rgb, pi, vp.sdf + v, vp.sumsq + v, vp.rgbw + v);
  }
}

// =====================================================================================================
// bookkeeping kernels
// =====================================================================================================
__global__ __launch_bounds__(256) void k_init_table(u64* keys, const size_t slots) {
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < slots; i += (size_t) gridDim.x * 256) keys[i] = kKeyEmpty;
}
// the per-frame report of mrh_peek_free_blocks / mrh_peek_error_flags: ctr[0 .. 4] into a pinned host record (one lane; a
// 20-byte hipMemcpyAsync would cost the host as much as the two launches of the frame together)
__global__ void k_report(const int* __restrict__ ctr, int* __restrict__ host_rec) {
  if (threadIdx.x < 5) host_rec[threadIdx.x] = ctr[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_fill_u64(u64* p, const size_t n, const u64 v) {
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) p[i] = v;
}
// heap[i] = N - 1 - i (voxel_data_structures.cpp:60-66)
__global__ __launch_bounds__(256) void k_fill_u32(u32* p, const size_t n, const u32 v) {
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) p[i] = v;
}
// free list of fine block indices, popped from the top: index 0 first (heap[i] = N - 1 - i, vds.cu:33-40) — or, for the
// addressing test (MRH_DEBUG_HEAP_DESCENDING=1), the HIGHEST index first, so that a small scene lands at the far end of a
// pool sized to HBM and every byte offset it touches needs more than 32 bits
__global__ __launch_bounds__(256) void k_init_heap(u32* heap, const u32 n, const int highest_first) {
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) heap[i] = highest_first ? (u32) i : n - 1 - (u32) i;
}
__global__ __launch_bounds__(256) void k_count_live(const Tab t) {
  const int hwm = t.ctr[CTR_HWM_FINE];
  const int total = t.multi_res ? hwm * 9 : hwm;
  for (int base = blockIdx.x * 256; base < total; base += gridDim.x * 256) {
    const int i = base + threadIdx.x;
    bool lf = false, lc = false;
    int4 d = make_int4(0, 0, 0, 0);
    if (i < total) {
      d = i < hwm ? t.desc_fine[i] : t.desc_coarse[i - hwm];
      if (i < hwm) lf = (d.w & 1) != 0;
      else lc = (d.w & 1) != 0;
    }
    // probe-path length of every live key (mrh_stats.max_probe_length)
    u32 steps = 0;
    if (lf || lc) {
      u64 key;
      pack_key(mki3(d.x, d.y, d.z), key);
      u32 s = hash_key(key) & t.slot_mask;
      for (steps = 1; steps < t.max_probe && t.keys[s] != key; steps++) s = (s + 1) & t.slot_mask;
    }
    steps = wave_max_u32(steps);
    const u64 bf = __ballot(lf), bc = __ballot(lc);
    if (lane_id() == 0) {
      if (bf) atomicAdd(&t.ctr[CTR_LIVE_FINE], __popcll(bf));
      if (bc) atomicAdd(&t.ctr[CTR_LIVE_COARSE], __popcll(bc));
      if (steps) atomicMax(&t.ctr[CTR_MAXPROBE], (int) steps);
    }
  }
}

// =====================================================================================================
// table maintenance.  Erasing from an open-address table leaves a tombstone (the reference's buckets return the slot to
// FREE, vds.cu:1727-1824), and GC frees and re-creates the free-space blocks of the truncation band every frame: without
// upkeep the never-used slots only get fewer, and a lookup of an absent key — 27 per block in k_mc, one per tile key in
// k_front — walks ever longer runs.  Every few dozen frames (mrh_capi.hip: maintain_table) a census counts the tombstones;
// above a quarter of the slots (or when a key was left without storage) the key array is cleared and rebuilt from the dense
// block descriptors: O(live blocks), between two frames, no host round trip — the decision stays on the device.
// =====================================================================================================
__global__ __launch_bounds__(256) void k_table_census(const Tab t, const size_t slots) {
  u32 n = 0;
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < slots; i += (size_t) gridDim.x * 256) n += t.keys[i] == kKeyTomb ? 1u : 0u;
  for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
  if (lane_id() == 0 && n) atomicAdd(&t.ctr[CTR_TOMBS_NOW], (int) n);
}
__global__ void k_rehash_decide(const Tab t, const u32 tomb_limit, const int force) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int tombs = t.ctr[CTR_TOMBS_NOW];
  t.ctr[CTR_TOMBS_NOW] = 0;
  const int go = (force || (u32) tombs > tomb_limit || t.ctr[CTR_ORPHANS] > 0) ? 1 : 0;
  t.ctr[CTR_REHASH] = go;
  t.ctr[CTR_TOMBS] = go ? 0 : tombs;
  if (go) { t.ctr[CTR_NREHASH]++; t.ctr[CTR_ORPHANS] = 0; }
}
__global__ __launch_bounds__(256) void k_rehash_clear(const Tab t, const size_t slots) {
  if (t.ctr[CTR_REHASH] == 0) return;
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < slots; i += (size_t) gridDim.x * 256) t.keys[i] = kKeyEmpty;
}
__global__ __launch_bounds__(256) void k_rehash_insert(const Tab t) {
  if (t.ctr[CTR_REHASH] == 0) return;
  const int hwm = t.ctr[CTR_HWM_FINE];
  const int total = t.multi_res ? hwm * 9 : hwm;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int4 d = i < hwm ? t.desc_fine[i] : t.desc_coarse[i - hwm];
    if (!(d.w & 1)) continue;
    u64 key;
    pack_key(mki3(d.x, d.y, d.z), key);
    const int slot = hash_insert(t, key);  // distinct keys into a table without tombstones
    if (slot < 0) { atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE); continue; }
    t.vals[slot] = i < hwm ? (u32) i : ((u32) (i - hwm) | kValCoarseBit);
  }
}

// SoA pool -> reference 12-byte Voxel AoS (vhu.cuh:8-22), for dump / tests / streaming
__global__ __launch_bounds__(512) void k_dump(const Tab t, const int first, const int n, int4* __restrict__ descs, char* __restrict__ voxels) {
  const int v = threadIdx.x;
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    const int4 ent = t.compact[first + e];
    const u32 val = (u32) ent.w;
    const bool coarse = (val & kValCoarseBit) != 0;
    if (v == 0) descs[e] = make_int4(ent.x, ent.y, ent.z, coarse ? 1 : 0);
    float sdf = 0.f, ss = 0.f;
    u32 rgbw = 0;
    if (!coarse || v < kCoarseVoxels) {
      const VoxPtr vp = vox_ptr(t, val);
      sdf = vp.sdf[v]; ss = vp.sumsq[v]; rgbw = vp.rgbw[v];
    }
    u32* out = (u32*) (voxels + ((size_t) e * kBlockVoxels + v) * 12);
    out[0] = __float_as_uint(sdf);
    out[1] = __float_as_uint(ss);
    out[2] = rgbw;
  }
}

// reference 12-byte Voxel AoS -> SoA pool, one workgroup per incoming block (inverse of k_dump).  Block e has its
// descriptor at descs + e * desc_stride and its 512 voxels at voxels + e * vox_stride (bytes): separate arrays for
// mrh_import_blocks (16 / 6144), interleaved mrh_block_record for the multi-GPU exchange (6160 / 6160).
//   MODE 0  import: insert or overwrite
//   MODE 1  halo:   only blocks of other shards that are 26-adjacent to a block position this shard owns; newly inserted
//                   ones are remembered in halo_list (mrh_drop_blocks(MRH_DROP_HALO))
//   MODE 2  merge:  voxel-wise weighted merge into what the map holds (combineVoxel, vhu.cuh:167-181)
//           One position at two resolutions (variance-adaptive sub-maps; the reference, single-GPU, has no rule): the COARSE side
//           wins — reallocBlock (vds.cu:627-755) frees the fine block and starts the coarse one from the current frame, nothing of
//           a fine payload survives a coarsening — so a fine record onto a coarse block is dropped, a coarse record onto a fine
//           block replaces it: the fine slot is emptied, its index parked in `released` (released[0] = count; pushed onto the fine
//           free list by k_release_fine behind this launch: a stack cannot take pushes next to this kernel's pops).
constexpr int kImportPlain = 0, kImportHalo = 1, kImportMerge = 2;
__global__ __launch_bounds__(256) void k_release_fine(const Tab t, const u32* __restrict__ released) {
  __shared__ int s_base;
  const int n = (int) released[0];
  if (threadIdx.x == 0) s_base = n ? atomicAdd(&t.ctr[CTR_HEAP_FINE], n) : 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) t.heap_fine[s_base + 1 + i] = released[1 + i];  // vds.cu:53-57
}
// coarse records among `n` (mrh_unpack_blocks sizes the coarse free list by it before a merge)
__global__ __launch_bounds__(256) void k_count_coarse_records(const char* __restrict__ descs, const size_t desc_stride, const int n, u32* __restrict__ out) {
  u32 c = 0;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) c += (*(const int4*) (descs + (size_t) e * desc_stride)).w != 0 ? 1u : 0u;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
template <int MODE>
__global__ __launch_bounds__(512) void k_import(const Map m, const Tab t, uint2* __restrict__ summary, const int n, const char* __restrict__ descs,
                                                const size_t desc_stride, const char* __restrict__ voxels, const size_t vox_stride,
                                                int4* __restrict__ halo_list, u32* __restrict__ taken, u32* __restrict__ released = nullptr) {
  __shared__ u32 s_val, s_released;
  __shared__ float s_min[8];
  __shared__ u32 s_max[8];
  const int v = threadIdx.x;
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    const int4 d = *(const int4*) (descs + (size_t) e * desc_stride);
    const bool coarse = d.w != 0;
    if (v == 0) {
      u32 val = kValNone;
      u32 rel = kValNone;
      u64 key;
      bool wanted = true;
      if (MODE == kImportHalo) {
        const i3 b = mki3(d.x, d.y, d.z);
        wanted = false;
        if (!owns_block(m, b)) {
          for (int k = 0; k < 27 && !wanted; k++) wanted = owns_block(m, mki3(b.x + (k % 3) - 1, b.y + ((k / 3) % 3) - 1, b.z + (k / 9) - 1));
        }
      }
      if (!wanted) {
      } else if (!pack_key(mki3(d.x, d.y, d.z), key)) {
        atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_RANGE);
      } else {
        int slot = hash_find(t, key);
        const u32 have = slot >= 0 ? t.vals[slot] : kValNone;
        if (have != kValNone && ((have & kValCoarseBit) != 0) == coarse) {
          val = have;  // overwrite / merge in place
        } else if (have != kValNone && MODE == kImportMerge) {
          if (coarse) {  // map fine, record coarse: the coarse unit takes the table slot, the fine block goes
            const int idx = atomicSub(&t.ctr[CTR_HEAP_COARSE], 1);
            if (idx < 0) {
              atomicAdd(&t.ctr[CTR_HEAP_COARSE], 1);
              atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_POOL);  // no coarse unit: the fine block stays as it is
            } else {
              const u32 u = t.heap_coarse[idx];
              val = u | kValCoarseBit;
              t.vals[slot] = val;
              t.desc_coarse[u] = make_int4(d.x, d.y, d.z, 1);
              t.desc_fine[have].w = 0;
              rel = have;
              if (released) released[1 + atomicAdd(&released[0], 1u)] = have;
            }
          }  // map coarse, record fine: the record is dropped
        } else if (have != kValNone) {
          atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);  // same position at another resolution: an import cannot hold both
        } else {
          if (slot < 0) slot = hash_insert(t, key);  // (slot >= 0: a key left without storage by an exhausted pool takes the block)
          if (slot < 0) {
            atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
          } else {
            const int idx = atomicSub(&t.ctr[coarse ? CTR_HEAP_COARSE : CTR_HEAP_FINE], 1);
            if (idx < 0) {
              publish_without_storage(t, slot, coarse ? CTR_HEAP_COARSE : CTR_HEAP_FINE);
            } else if (coarse) {
              const u32 u = t.heap_coarse[idx];
              val = u | kValCoarseBit;
              t.vals[slot] = val;
              t.desc_coarse[u] = make_int4(d.x, d.y, d.z, 1);
            } else {
              const u32 H = t.heap_fine[idx];
              val = H;
              t.vals[slot] = val;
              t.desc_fine[H] = make_int4(d.x, d.y, d.z, 1);
              atomicMax(&t.ctr[CTR_HWM_FINE], (int) H + 1);
            }
            if (MODE == kImportHalo && val != kValNone) halo_list[atomicAdd(&t.ctr[CTR_HALO], 1)] = make_int4(d.x, d.y, d.z, (int) val);
          }
        }
      }
      if (val != kValNone && taken) atomicAdd(taken, 1u);
      s_val = val;
      s_released = rel;
    }
    __syncthreads();
    const u32 val = s_val;
    if (MODE == kImportMerge && s_released != kValNone && v < kFineBytes / 16)  // the fine block that made way: zero, as a freed block is
      ((uint4*) (t.pool + (size_t) s_released * kFineBytes))[v] = make_uint4(0, 0, 0, 0);
    float mn = 3.402823466e+38f;
    u32 mx = 0;
    if (val != kValNone && (!coarse || v < kCoarseVoxels)) {
      const u32* in = (const u32*) (voxels + (size_t) e * vox_stride + (size_t) v * 12);
      const VoxPtr vp = vox_ptr(t, val);
      float sdf = __uint_as_float(in[0]), ss = __uint_as_float(in[1]);
      u32 rgbw = in[2];
      bool store = true;
      if (MODE == kImportMerge) {
        const float s0 = vp.sdf[v];
        const u32 old = vp.rgbw[v];
        const u32 w0 = old >> 24, w1 = rgbw >> 24;
        if (w1 == 0) {  // nothing observed on the incoming side
          store = false;
          sdf = s0; rgbw = old;
        } else if (w0 != 0) {
          const u32 c0x = old & 0x00FFFFFFu, c1x = rgbw & 0x00FFFFFFu;
          const u32 rgbn = (c0x | c1x) - (((c0x ^ c1x) >> 1) & 0x007F7F7Fu);  // u8(0.5 c0 + 0.5 c1 + 0.5) per channel (mrh_fast.h: blend4)
          const u32 wmax = (u32) (m.weight_max & 0xFF);
          const u32 wn = (w0 + w1) < wmax ? (w0 + w1) : wmax;
          sdf = (s0 * (float) w0 + sdf * (float) w1) / (float) (int) (w0 + w1);
          rgbw = rgbn | (wn << 24);
        }
      }
      if (store) {
        vp.sdf[v] = sdf;
        vp.sumsq[v] = ss;
        vp.rgbw[v] = rgbw;
      }
      mx = rgbw >> 24;
      if (mx != 0) mn = fabsf(sdf);
    }
    for (int off = 32; off > 0; off >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, off));
      const u32 o = __shfl_xor(mx, off);
      mx = o > mx ? o : mx;
    }
    if (lane_id() == 0) { s_min[v >> 6] = mn; s_max[v >> 6] = mx; }
    __syncthreads();
    if (v == 0 && val != kValNone && !coarse && summary) {
      for (int i = 1; i < 8; i++) { mn = fminf(mn, s_min[i]); mx = s_max[i] > mx ? s_max[i] : mx; }
      summary[val] = make_uint2(__float_as_uint(mn), mx);  // GC summary of the fast path
    }
    __syncthreads();
  }
}

// ---- multi-GPU block exchange (include/mrhash_hip.h: mrh_pack_blocks / mrh_drop_blocks) ---------------------------
// rank that owns a block: cubes of 2^shard_chunk_log2 blocks, hashed (owns_block is `== shard_rank`)
__device__ __forceinline__ int owner_rank(const Map& m, const i3 b) {
  if (m.shard_count <= 1) return 0;
  const int sh = m.shard_chunk_log2;
  const u32 cx = (u32) (b.x >> sh), cy = (u32) (b.y >> sh), cz = (u32) (b.z >> sh);
  const u32 h = (cx * 73856093u) ^ (cy * 19349669u) ^ (cz * 83492791u);
  return (int) ((h ^ (h >> 15)) % (u32) m.shard_count);
}
constexpr int kSelHalo = 0, kSelOwner = 1, kSelForeign = 2, kSelAll = 3;
// live blocks that satisfy the predicate -> Tab::compact[0 .. CTR_COMPACT) (ballot + popcount prefix, one atomic per wave)
__global__ __launch_bounds__(256) void k_select_blocks(const Map m, const Tab t, const int mode, const int rank_arg) {
  const int hwm = t.ctr[CTR_HWM_FINE];
  const int total = t.multi_res ? hwm * 9 : hwm;
  for (int base = blockIdx.x * 256; base < total; base += gridDim.x * 256) {
    const int i = base + threadIdx.x;
    bool keep = false;
    int4 d = make_int4(0, 0, 0, 0);
    u32 val = 0;
    if (i < total) {
      if (i < hwm) { d = t.desc_fine[i]; val = (u32) i; }
      else { const u32 u = (u32) (i - hwm); d = t.desc_coarse[u]; val = u | kValCoarseBit; }
      if (d.w & 1) {
        const i3 b = mki3(d.x, d.y, d.z);
        const int owner = owner_rank(m, b);
        if (mode == kSelHalo) {
          const int side = 1 << m.shard_chunk_log2;
          const int lx = b.x & (side - 1), ly = b.y & (side - 1), lz = b.z & (side - 1);
          keep = owner == m.shard_rank && (lx == 0 || lx == side - 1 || ly == 0 || ly == side - 1 || lz == 0 || lz == side - 1);
        } else if (mode == kSelOwner) {
          keep = owner == rank_arg;
        } else if (mode == kSelForeign) {
          keep = owner != m.shard_rank;
        } else {
          keep = true;
        }
      }
    }
    const u64 ballot = __ballot(keep);
    if (ballot) {
      const int leader = __ffsll((long long) ballot) - 1;
      int wbase = 0;
      if ((int) lane_id() == leader) wbase = atomicAdd(&t.ctr[CTR_COMPACT], __popcll(ballot));
      wbase = __shfl(wbase, leader);
      if (keep) t.compact[wbase + __popcll(ballot & lanemask_lt())] = make_int4(d.x, d.y, d.z, (int) val);
    }
  }
}
// Tab::compact[first .. first + n) -> mrh_block_record[n] (16-byte descriptor + 512 reference-layout voxels, 6160 bytes)
__global__ __launch_bounds__(512) void k_pack_records(const Tab t, const int first, const int n, char* __restrict__ records) {
  const int v = threadIdx.x;
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    const int4 ent = t.compact[first + e];
    const u32 val = (u32) ent.w;
    const bool coarse = (val & kValCoarseBit) != 0;
    char* rec = records + (size_t) e * 6160;
    if (v == 0) *(int4*) rec = make_int4(ent.x, ent.y, ent.z, coarse ? 1 : 0);
    float sdf = 0.f, ss = 0.f;
    u32 rgbw = 0;
    if (!coarse || v < kCoarseVoxels) {
      const VoxPtr vp = vox_ptr(t, val);
      sdf = vp.sdf[v]; ss = vp.sumsq[v]; rgbw = vp.rgbw[v];
    }
    u32* out = (u32*) (rec + 16 + (size_t) v * 12);
    out[0] = __float_as_uint(sdf);
    out[1] = __float_as_uint(ss);
    out[2] = rgbw;
  }
}

__global__ void k_get_voxel(const Map m, const Tab t, const int vx, const int vy, const int vz, u32* __restrict__ out) {
  const i3 v = mki3(vx, vy, vz);
  const i3 b = voxel_to_block(v, m.vs);
  u64 key;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!pack_key(b, key)) return;
  const int s = hash_find(t, key);
  if (s < 0) return;
  const u32 val = t.vals[s];
  if (val == kValNone) return;
  const int res = (val & kValCoarseBit) ? 1 : 0;
  const VoxPtr vp = vox_ptr(t, val);
  const u32 li = voxel_local_index(v, res);
  out[0] = __float_as_uint(vp.sdf[li]);
  out[1] = __float_as_uint(vp.sumsq[li]);
  out[2] = vp.rgbw[li];
  out[3] = 1;
}

// rcp_refined of the weight sums, for the host to compare with the correctly rounded reciprocals (Map::wsum_two_steps)
__global__ void k_rcp_weights(float* __restrict__ out) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < kRcpWeightEntries) out[w] = w ? rcp_refined((float) w) : 0.f;
}

// Exhaustive check behind Map::half_vs_two_steps: div_cr(a, b, r) against the compiler's correctly rounded a / b for every
// dividend in the working range 2^-100 <= |a| <= 2^100 (and a == +0); out[0] = number of mismatches.  4096 x 256 threads.
__global__ __launch_bounds__(256) void k_check_div_cr(const float b, const float r, u32* __restrict__ out) {
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  u32 bad = 0;
  for (u32 k = 0; k < 4096u; k++) {
    const u32 bits = tid * 4096u + k;
    const float a = __uint_as_float(bits), mag = fabsf(a);
    if (!((mag >= 7.888609e-31f && mag <= 1.2676506e30f) || bits == 0u)) continue;
    if (__float_as_uint(div_cr(a, b, r)) != __float_as_uint(a / b)) bad++;
  }
  if (bad) atomicAdd(out, bad);
}

// Self-test: div_rr(a, b, rcp_refined(b)) must equal the compiler's correctly rounded a / b bit for bit on the
// operand domains the fused kernel uses it for (pixel projection, running mean, variance deltas) plus a broad
// log-uniform domain.  Returns the number of mismatching samples.
__device__ __forceinline__ u32 st_rand(u64& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return (u32) (s >> 33) ^ (u32) (s >> 11);
}
// exhaustive check behind Map::block_shift_limit: smallest |v| in [0, 2^23) whose float voxel -> block conversion
// (either sign) differs from the arithmetic shift; out[0] must be initialised to 2^23
__global__ __launch_bounds__(256) void k_block_shift_limit(const float vs, u32* __restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= (1 << 23)) return;
  const i3 bp = voxel_to_block(mki3(v, v, v), vs);
  const i3 bn = voxel_to_block(mki3(-v, -v, -v), vs);
  if (bp.x != (v >> 3) || bn.x != ((-v) >> 3)) atomicMin(out, (u32) v);
}

__global__ __launch_bounds__(256) void k_selftest_division(const u64 seed, const u32 iters, u64* __restrict__ mismatches) {
  u64 st = seed * 0x9E3779B97F4A7C15ull + (u64) (blockIdx.x * 256 + threadIdx.x) * 0xD1B54A32D192ED03ull + 1;
  u32 bad = 0;
  for (u32 it = 0; it < iters; it++) {
    const u32 dom = st_rand(st) & 3;
    float a, b;
    const float ua = (float) (int) (st_rand(st) >> 8) * (1.0f / 16777216.0f);  // [0,1)
    const float ub = (float) (int) (st_rand(st) >> 8) * (1.0f / 16777216.0f);
    if (dom == 0) { a = (ua - 0.5f) * 2.0e5f; b = 0.01f + ub * 30.0f; }              // f * x / z
    else if (dom == 1) { a = (ua - 0.5f) * 40.0f; b = (float) (1 + (st_rand(st) % 510)); }  // weighted mean
    else if (dom == 2) { a = (ua - 0.5f) * 0.5f; b = 0.0005f + ub * 0.2f; }           // delta / (vs / 2)
    else {  // log-uniform magnitudes 2^-40 .. 2^40, random signs
      a = __uint_as_float(((st_rand(st) % 80 + 87) << 23) | (st_rand(st) & 0x7FFFFF) | (st_rand(st) & 0x80000000u));
      b = __uint_as_float(((st_rand(st) % 80 + 87) << 23) | (st_rand(st) & 0x7FFFFF) | (st_rand(st) & 0x80000000u));
    }
    const float q_ref = a / b;
    const float q = div_rr(a, b, rcp_refined(b));
    if (__float_as_uint(q) != __float_as_uint(q_ref)) bad++;
  }
  if (bad) atomicAdd(mismatches, (u64) bad);
}

}  // namespace mrh
