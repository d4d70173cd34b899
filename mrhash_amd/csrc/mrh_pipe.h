// mrh_pipe.h — block allocation split in two so that its ray-marching half can run AHEAD of the map updates.
//
//   k_rays    (input stream)  per pixel: cleaned depth + packed colour, DDA along the truncation segment, block keys
//                             de-duplicated per 16x16 tile in LDS, the tile's distinct keys written to a per-tile list
//                             in HBM.  Reads only the images and the pose: it does not touch the map, so the rays of
//                             frame f+1 are marched on a second HIP stream while frame f is still being integrated.
//   k_insert  (map stream)    one wave per tile: frustum test + lock-free insert of the listed keys (the part of
//                             allocation that needs the table as frame f-1 left it).
//
// Same arithmetic and same inserted set as k_alloc2 (mrh_fast.h); reference: allocBlocksKernel vds.cu:758-857.
#pragma once

#include "mrh_fast.h"

namespace mrh {

constexpr int kRayTile = 16;
constexpr int kRayCap = kRayTile * kRayTile * 4;  // keys per tile list / LDS set (1024)
constexpr u32 kTileOverflow = 0xFFFFFFFFu;        // tile_count value: LDS set saturated, k_insert re-walks the rays

// Amanatides-Woo walk over the blocks of one pixel's segment [d - t, d + t] (vds.cu:771-853); visit(block, key) is
// called for every traversed block this shard owns.  `d` is the cleaned depth (0 = invalid pixel).
template <typename V>
__device__ __forceinline__ void walk_ray(const Cam& c, const Map& m, const Tab& t, const int row, const int col, const float d, V&& visit) {
  const float tr = get_truncation(d, m.trunc, m.trunc_scale);
  const float dmin = fminf(c.max_int_dist, d - tr);
  const float dmax = fminf(c.max_int_dist, d + tr);
  if ((d == 0.f) || (dmin >= dmax)) return;
  const f3 pw_min = se3_apply(c.R, c.t, inverse_projection(c, (u32) row, (u32) col, dmin));
  const f3 pw_max = se3_apply(c.R, c.t, inverse_projection(c, (u32) row, (u32) col, dmax));
  const f3 dd = mk3(pw_max.x - pw_min.x, pw_max.y - pw_min.y, pw_max.z - pw_min.z);
  const float inv_len = 1.0f / sqrtf(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z);  // normalize, cuda_math.cuh:1075-1078
  const f3 dir = mk3(dd.x * inv_len, dd.y * inv_len, dd.z * inv_len);
  const GridRcp grid = make_grid_rcp(m.vs);
  i3 cur = world_to_block_r(grid, pw_min);
  const i3 end = world_to_block_r(grid, pw_max);
  const f3 step = mk3((float) signi(dir.x), (float) signi(dir.y), (float) signi(dir.z));
  const i3 nb = mki3(cur.x + f2i(clampf(step.x, 0.0f, 1.f)), cur.y + f2i(clampf(step.y, 0.0f, 1.f)), cur.z + f2i(clampf(step.z, 0.0f, 1.f)));
  const f3 bw = voxel_to_world(m.vs, mki3(nb.x * kBlockSide, nb.y * kBlockSide, nb.z * kBlockSide));
  const f3 boundary = mk3(bw.x - 0.5f * m.vs, bw.y - 0.5f * m.vs, bw.z - 0.5f * m.vs);
  const f3 rd = mk3(rcp_refined(dir.x), rcp_refined(dir.y), rcp_refined(dir.z));
  f3 t_max = mk3(div_rr(boundary.x - pw_min.x, dir.x, rd.x), div_rr(boundary.y - pw_min.y, dir.y, rd.y), div_rr(boundary.z - pw_min.z, dir.z, rd.z));
  f3 t_delta = mk3(div_rr(step.x * (float) kBlockSide * m.vs, dir.x, rd.x), div_rr(step.y * (float) kBlockSide * m.vs, dir.y, rd.y),
                   div_rr(step.z * (float) kBlockSide * m.vs, dir.z, rd.z));
  const i3 bound = mki3(f2i((float) end.x + step.x), f2i((float) end.y + step.y), f2i((float) end.z + step.z));
  // vds.cu:801-827 (the second test of each pair compares a position with a direction; kept literally)
  if (fabsf(dir.x) < kFloatEps) { t_max.x = kFltMax; t_delta.x = kFltMax; }
  if (fabsf(boundary.x - dir.x) < kFloatEps) { t_max.x = kFltMax; t_delta.x = kFltMax; }
  if (fabsf(dir.y) < kFloatEps) { t_max.y = kFltMax; t_delta.y = kFltMax; }
  if (fabsf(boundary.y - dir.y) < kFloatEps) { t_max.y = kFltMax; t_delta.y = kFltMax; }
  if (fabsf(dir.z) < kFloatEps) { t_max.z = kFltMax; t_delta.z = kFltMax; }
  if (fabsf(boundary.z - dir.z) < kFloatEps) { t_max.z = kFltMax; t_delta.z = kFltMax; }
#pragma unroll 1
  for (u32 iter = 0; iter < kMaxDdaIter; iter++) {
    u64 key;
    if (!pack_key(cur, key)) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_RANGE);
    else if (owns_block(m, cur)) visit(cur, key);
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

// The same walk with the hot loop kept lean for the allocation workgroups of k_front: the per-step three-way branch of
// walk_ray becomes selects (block coordinates are far below 2^24, so the reference's float increment
// `int(float(cur) + step)` is the integer increment), the key-range test is hoisted (every traversed block lies in
// the box spanned by the first block and `bound`), and nothing rare lives inside the loop — `visit(block, key)`
// returns false to stop (the caller then falls back to walk_ray for this pixel).  Visits exactly walk_ray's blocks.
struct RayState {
  i3 cur, bound, step;
  f3 t_max, t_delta;
  bool valid;
};
// block-level DDA state for the world-space segment pw_min -> pw_max (vds.cu:782-827; allocBlocks3DKernel :966-1005)
__device__ __forceinline__ RayState ray_from_segment(const Map& m, const f3 pw_min, const f3 pw_max) {
  RayState r;
  const f3 dd = mk3(pw_max.x - pw_min.x, pw_max.y - pw_min.y, pw_max.z - pw_min.z);
  const float inv_len = 1.0f / sqrtf(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z);  // normalize, cuda_math.cuh:1075-1078
  const f3 dir = mk3(dd.x * inv_len, dd.y * inv_len, dd.z * inv_len);
  const GridRcp grid = make_grid_rcp(m.vs);
  r.cur = world_to_block_fast(grid, pw_min, m.block_shift_limit);
  const i3 end = world_to_block_fast(grid, pw_max, m.block_shift_limit);
  const f3 step = mk3((float) signi(dir.x), (float) signi(dir.y), (float) signi(dir.z));
  r.step = mki3(signi(dir.x), signi(dir.y), signi(dir.z));
  const i3 nb = mki3(r.cur.x + f2i(clampf(step.x, 0.0f, 1.f)), r.cur.y + f2i(clampf(step.y, 0.0f, 1.f)), r.cur.z + f2i(clampf(step.z, 0.0f, 1.f)));
  const f3 bw = voxel_to_world(m.vs, mki3(nb.x * kBlockSide, nb.y * kBlockSide, nb.z * kBlockSide));
  const f3 boundary = mk3(bw.x - 0.5f * m.vs, bw.y - 0.5f * m.vs, bw.z - 0.5f * m.vs);
  const f3 rd = mk3(rcp_refined(dir.x), rcp_refined(dir.y), rcp_refined(dir.z));
  r.t_max = mk3(div_rr(boundary.x - pw_min.x, dir.x, rd.x), div_rr(boundary.y - pw_min.y, dir.y, rd.y), div_rr(boundary.z - pw_min.z, dir.z, rd.z));
  r.t_delta = mk3(div_rr(step.x * (float) kBlockSide * m.vs, dir.x, rd.x), div_rr(step.y * (float) kBlockSide * m.vs, dir.y, rd.y),
                  div_rr(step.z * (float) kBlockSide * m.vs, dir.z, rd.z));
  r.bound = mki3(f2i((float) end.x + step.x), f2i((float) end.y + step.y), f2i((float) end.z + step.z));
  // vds.cu:801-827 (the second test of each pair compares a position with a direction; kept literally)
  const bool gx = (fabsf(dir.x) < kFloatEps) || (fabsf(boundary.x - dir.x) < kFloatEps);
  const bool gy = (fabsf(dir.y) < kFloatEps) || (fabsf(boundary.y - dir.y) < kFloatEps);
  const bool gz = (fabsf(dir.z) < kFloatEps) || (fabsf(boundary.z - dir.z) < kFloatEps);
  r.t_max.x = gx ? kFltMax : r.t_max.x; r.t_delta.x = gx ? kFltMax : r.t_delta.x;
  r.t_max.y = gy ? kFltMax : r.t_max.y; r.t_delta.y = gy ? kFltMax : r.t_delta.y;
  r.t_max.z = gz ? kFltMax : r.t_max.z; r.t_delta.z = gz ? kFltMax : r.t_delta.z;
  r.valid = true;
  return r;
}
__device__ __forceinline__ RayState ray_setup(const Cam& c, const Map& m, const int row, const int col, const float d) {
  RayState r;
  r.valid = false;
  const float tr = get_truncation(d, m.trunc, m.trunc_scale);
  const float dmin = fminf(c.max_int_dist, d - tr);
  const float dmax = fminf(c.max_int_dist, d + tr);
  if ((d == 0.f) || (dmin >= dmax)) return r;
  const f3 pw_min = se3_apply(c.R, c.t, inverse_projection(c, (u32) row, (u32) col, dmin));
  const f3 pw_max = se3_apply(c.R, c.t, inverse_projection(c, (u32) row, (u32) col, dmax));
  return ray_from_segment(m, pw_min, pw_max);
}
// true if every block key of the walk is representable (pack_key cannot fail inside the loop)
__device__ __forceinline__ bool ray_keys_in_range(const RayState& r) {
  u64 k;
  return pack_key(r.cur, k) && pack_key(r.bound, k);
}
template <typename V>
__device__ __forceinline__ bool walk_ray_lean(const Map& m, RayState r, V&& visit) {
#pragma unroll 1
  for (u32 iter = 0; iter < kMaxDdaIter; iter++) {
    u64 key;
    pack_key(r.cur, key);
    if (owns_block(m, r.cur) && !visit(r.cur, key)) return false;
    const bool ax = r.t_max.x < r.t_max.y && r.t_max.x < r.t_max.z;
    const bool az = !ax && (r.t_max.z < r.t_max.y);
    const bool ay = !ax && !az;
    r.cur.x += ax ? r.step.x : 0;
    r.cur.y += ay ? r.step.y : 0;
    r.cur.z += az ? r.step.z : 0;
    if ((ax && r.cur.x == r.bound.x) || (ay && r.cur.y == r.bound.y) || (az && r.cur.z == r.bound.z)) break;
    r.t_max.x = ax ? r.t_max.x + r.t_delta.x : r.t_max.x;
    r.t_max.y = ay ? r.t_max.y + r.t_delta.y : r.t_max.y;
    r.t_max.z = az ? r.t_max.z + r.t_delta.z : r.t_max.z;
  }
  return true;
}

__global__ __launch_bounds__(kRayTile * kRayTile) void k_rays(const Cam c, const Map m, const Tab t, const float* __restrict__ depth,
                                                              const uint8_t* __restrict__ rgb, float* __restrict__ depth_clean,
                                                              u32* __restrict__ rgbx, u64* __restrict__ tile_keys, u32* __restrict__ tile_count) {
  constexpr int NT = kRayTile * kRayTile;
  __shared__ u64 set[kRayCap];
  __shared__ u64 list[kRayCap];
  __shared__ u32 s_count, s_overflow;
  const int tid = threadIdx.y * kRayTile + threadIdx.x;
  for (int i = tid; i < kRayCap; i += NT) set[i] = kKeyEmpty;
  if (tid == 0) { s_count = 0; s_overflow = 0; }
  __syncthreads();
  const int row = blockIdx.y * kRayTile + threadIdx.y;
  const int col = blockIdx.x * kRayTile + threadIdx.x;
  float d = 0.f;
  if (row < c.rows && col < c.cols) {
    const size_t pix = (size_t) row * c.cols + col;
    d = depth[pix];
    if (d <= c.min_depth || d > c.max_depth) d = 0.f;  // camera.cu:13-18
    depth_clean[pix] = d;
    const uint8_t* px = rgb + pix * 3;
    rgbx[pix] = (u32) px[0] | ((u32) px[1] << 8) | ((u32) px[2] << 16);
    walk_ray(c, m, t, row, col, d, [&](const i3 cur, const u64 key) {
      u32 s = (u32) __mul24(cur.z, 5851) + (u32) __mul24(cur.y, 73) + (u32) cur.x;  // cheap tile-local mix
      s = (s ^ (s >> 7)) & (kRayCap - 1);
#pragma unroll 1
      for (int p = 0; p < kSetProbe; p++) {
        const u64 old = atomicCAS(&set[s], kKeyEmpty, key);
        if (old == kKeyEmpty) { list[atomicAdd(&s_count, 1u)] = key; return; }  // first sighting in this tile
        if (old == key) return;
        s = (s + 1) & (kRayCap - 1);
      }
      s_overflow = 1;  // far, sparse rays: more distinct blocks than the set holds
    });
  }
  __syncthreads();
  const int tile = blockIdx.y * gridDim.x + blockIdx.x;
  const u32 n = s_count;
  for (u32 i = tid; i < n; i += NT) tile_keys[(size_t) tile * kRayCap + i] = list[i];
  if (tid == 0) tile_count[tile] = s_overflow ? kTileOverflow : n;
}

// un-aggregated insert of one block (used by the rare overflow path)
template <bool PROFILE>
__device__ __forceinline__ void insert_direct(const Cam& c, const Map& m, const Tab& t, const Fast& f, const i3 b, const u64 key) {
  if (!block_in_frustum_approx(c, m.vs, b)) return;
  const int slot = hash_insert(t, key);
  if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
  if (slot < 0) return;
  const int idx = atomicSub(&t.ctr[CTR_HEAP_FINE], 1);
  if (idx < 0) {
    atomicAdd(&t.ctr[CTR_HEAP_FINE], 1);
    atomicExch(&t.keys[slot], kKeyTomb);
    atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_POOL);
    return;
  }
  const u32 H = t.heap_fine[idx];
  t.vals[slot] = H;
  t.desc_fine[H] = make_int4(b.x, b.y, b.z, 1);
  f.summary[H] = make_uint2(0x7F7FFFFFu, 0u);
  if ((int) H >= __hip_atomic_load(&t.ctr[CTR_HWM_FINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&t.ctr[CTR_HWM_FINE], (int) H + 1);
  if (PROFILE) atomicAdd(&t.prof[PROF_INSERTED], 1ull);
}

template <bool PROFILE>
__global__ __launch_bounds__(64) void k_insert(const Cam c, const Map m, const Tab t, const Fast f, const u64* __restrict__ tile_keys,
                                               const u32* __restrict__ tile_count, const int tiles_x) {
  const int tile = blockIdx.x;
  const int lane = threadIdx.x;
  if (tile == 0 && lane == 0) { t.ctr[CTR_COMPACT] = 0; t.ctr[CTR_CULLED] = 0; t.ctr[CTR_FREED_EARLY] = 0; }
  const u32 n = tile_count[tile];
  if (n == kTileOverflow) {
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    for (int p = lane; p < kRayTile * kRayTile; p += 64) {
      const int row = ty * kRayTile + (p >> 4), col = tx * kRayTile + (p & 15);
      if (row < c.rows && col < c.cols)
        walk_ray(c, m, t, row, col, f.depth_clean[(size_t) row * c.cols + col],
                 [&](const i3 cur, const u64 key) { insert_direct<PROFILE>(c, m, t, f, cur, key); });
    }
    return;
  }
  u32 inserted = 0;
#pragma unroll 1
  for (u32 base = 0; base < n; base += 64) {
    const u32 i = base + lane;
    const bool active = i < n;
    const u64 key = active ? tile_keys[(size_t) tile * kRayCap + i] : kKeyEmpty;
    const i3 b = active ? unpack_key(key) : mki3(0, 0, 0);
    bool won = false;
    int slot = -1;
    if (active && block_in_frustum_approx(c, m.vs, b)) {
      slot = hash_insert(t, key);
      if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
      won = slot >= 0;
    }
    alloc_commit2(t, f, won, slot, b);
    if (PROFILE) inserted += __popcll(__ballot(won));
  }
  if (PROFILE && lane == 0 && inserted) atomicAdd(&t.prof[PROF_INSERTED], (u64) inserted);
}

}  // namespace mrh
