// mrh_pipe.h — the pixel-ray walk of block allocation (allocBlocksKernel vds.cu:758-857): the literal walk
// (walk_ray) and the lean one the allocation workgroups run (ray_setup / walk_ray_lean); mrh_fast2.h and mrh_lidar.h
// build their kernels on them.
#pragma once

#include "mrh_fast.h"

namespace mrh {

constexpr int kRayTile = 16;
constexpr int kRayCap = kRayTile * kRayTile * 4;  // keys per tile list / LDS set (1024)

// Amanatides-Woo walk over the blocks of one pixel's segment [d - t, d + t] (vds.cu:771-853); visit(block, key) is
// called for every traversed block this shard owns.  `d` is the cleaned depth (0 = invalid pixel).
template <bool SPH = false, typename V>
__device__ __forceinline__ void walk_ray(const Cam& c, const Map& m, const Tab& t, const int row, const int col, const float d, V&& visit) {
  const float tr = get_truncation(d, m.trunc, m.trunc_scale);
  const float dmin = fminf(c.max_int_dist, d - tr);
  const float dmax = fminf(c.max_int_dist, d + tr);
  if ((d == 0.f) || (dmin >= dmax)) return;
  const f3 pw_min = se3_apply(c.R, c.t, SPH ? inverse_projection_m(c, (u32) row, (u32) col, dmin) : inverse_projection(c, (u32) row, (u32) col, dmin));
  const f3 pw_max = se3_apply(c.R, c.t, SPH ? inverse_projection_m(c, (u32) row, (u32) col, dmax) : inverse_projection(c, (u32) row, (u32) col, dmax));
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
  // int(clamp(step, 0, 1)) is 1 for a positive step and 0 otherwise (step is -1, 0 or +1)
  const i3 nb = mki3(r.cur.x + (r.step.x > 0 ? 1 : 0), r.cur.y + (r.step.y > 0 ? 1 : 0), r.cur.z + (r.step.z > 0 ? 1 : 0));
  const f3 bw = voxel_to_world(m.vs, mki3(nb.x * kBlockSide, nb.y * kBlockSide, nb.z * kBlockSide));
  const f3 boundary = mk3(bw.x - 0.5f * m.vs, bw.y - 0.5f * m.vs, bw.z - 0.5f * m.vs);
  const f3 rd = mk3(rcp_refined(dir.x), rcp_refined(dir.y), rcp_refined(dir.z));
  r.t_max = mk3(div_rr(boundary.x - pw_min.x, dir.x, rd.x), div_rr(boundary.y - pw_min.y, dir.y, rd.y), div_rr(boundary.z - pw_min.z, dir.z, rd.z));
  r.t_delta = mk3(div_rr(step.x * (float) kBlockSide * m.vs, dir.x, rd.x), div_rr(step.y * (float) kBlockSide * m.vs, dir.y, rd.y),
                  div_rr(step.z * (float) kBlockSide * m.vs, dir.z, rd.z));
  r.bound = mki3(f2i_hw((float) end.x + step.x), f2i_hw((float) end.y + step.y), f2i_hw((float) end.z + step.z));
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
template <bool SPH = false>
__device__ __forceinline__ RayState ray_setup(const Cam& c, const Map& m, const int row, const int col, const float d) {
  RayState r;
  r.valid = false;
  const float tr = get_truncation(d, m.trunc, m.trunc_scale);
  const float dmin = fminf(c.max_int_dist, d - tr);
  const float dmax = fminf(c.max_int_dist, d + tr);
  if ((d == 0.f) || (dmin >= dmax)) return r;
  const f3 pw_min = se3_apply(c.R, c.t, SPH ? inverse_projection_m(c, (u32) row, (u32) col, dmin) : inverse_projection(c, (u32) row, (u32) col, dmin));
  const f3 pw_max = se3_apply(c.R, c.t, SPH ? inverse_projection_m(c, (u32) row, (u32) col, dmax) : inverse_projection(c, (u32) row, (u32) col, dmax));
  return ray_from_segment(m, pw_min, pw_max);
}
// true if every block key of the walk is representable (pack_key cannot fail inside the loop)
__device__ __forceinline__ bool ray_keys_in_range(const RayState& r) {
  u64 k;
  return pack_key(r.cur, k) && pack_key(r.bound, k);
}
template <typename V>
__device__ __forceinline__ bool walk_ray_lean(const Map& m, RayState r, V&& visit) {
  // the packed key follows the walk by addition: one step along an axis changes its 21-bit field by +-1, and every block
  // between the first one and `bound` is representable (ray_keys_in_range), so no field ever carries into its neighbour
  u64 key;
  pack_key(r.cur, key);
  const u64 dkx = (u64) ((long long) r.step.x * (1ll << 42)), dky = (u64) ((long long) r.step.y * (1ll << 21)), dkz = (u64) (long long) r.step.z;
#pragma unroll 1
  for (u32 iter = 0; iter < kMaxDdaIter; iter++) {
    if (owns_block(m, r.cur) && !visit(r.cur, key)) return false;
    const bool ax = r.t_max.x < r.t_max.y && r.t_max.x < r.t_max.z;
    const bool az = !ax && (r.t_max.z < r.t_max.y);
    const bool ay = !ax && !az;
    key += ax ? dkx : (ay ? dky : dkz);
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

}  // namespace mrh
