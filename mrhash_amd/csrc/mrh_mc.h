// mrh_mc.h — marching-cubes extraction kernels (gfx950).
//
// Reference: extractIsoSurfaceKernel / extractIsoSurfaceAtPosition / checkVertexVoxels
// (marching_cubes.cu:7-285), trilinearInterpolation (vds.cu:260-338), vertexInterp / appendTriangle
// (mesh_extractor.cu:6-55).  The reference appends triangles through one global atomic counter, so its
// buffer order is a race; here extraction is count -> exclusive scan -> emit over a position-sorted block
// list, which makes the triangle buffer canonical: (block position, voxel index, triangle number).
#pragma once

#include "mrh_kernels.h"
#include "../../include/mrhash_hip.h"
#include "../../include/mrh_mc_tables.h"

namespace mrh {

__device__ const uint8_t d_mc_tri[256][16] = MRH_MC_TRI_TABLE_INIT;

struct VoxSample {
  float sdf;
  u32 rgbw;
  int res;
  bool found;
};

// The 27 blocks around the block a workgroup is extracting, resolved ONCE per workgroup (table value, or kNbAbsent):
// on a single-resolution map every sample of marching cubes falls into this neighbourhood, so a lookup is a shift, a
// subtraction and one LDS read instead of the float voxel->block detour, a 64-bit hash and a dependent probe of the
// global table per sample (72 samples per voxel that reaches all eight corners).
constexpr u32 kNbAbsent = 0xFFFFFFFFu;
// The voxels marching cubes can sample for a block, staged ONCE per workgroup in LDS: the block's 8^3 fine cells and a
// rim of kHaloRim cells around it, {sdf, rgbw} per FINE cell (a coarse neighbour's voxel fills the 2^3 fine cells it
// covers, which is exactly what the reference's read through `local index >> resolution` returns for each of them).
// Reach: a corner sits <= 1/2 voxel from the sample position; the trilinear stencil adds one voxel of the sampled block's
// size; on a resolution jump the coarser re-sample (vds.cu:296-309) reaches pos - h + {0, 2h}: at most 3 fine cells
// from the voxel when the sampled block is coarse, 1 cell on a single-resolution neighbourhood.
constexpr int kHaloRim = 3;
constexpr int kHaloSide = 8 + 2 * kHaloRim;                      // 14
constexpr int kHaloCells = kHaloSide * kHaloSide * kHaloSide;    // 2744
struct Neigh {
  const u32* vals;  // LDS [27], index (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1); nullptr: no table (lookups outside k_mc)
  i3 base;          // block position of the workgroup's block
  int shift_limit;  // Map::block_shift_limit
  // LDS halo (nullptr: none); valid for cells within `halo_rim` of the block, under the workgroup's guarantee that
  // voxel -> block is the arithmetic shift for every voxel it can reach
  const float* halo_sdf;
  const u32* halo_rgbw;
  int halo_rim;
  float r_vs;       // rcp_refined(voxel size) for the position -> voxel divisions (0: plain IEEE division)
};
__device__ __forceinline__ Neigh neigh_none() {
  Neigh nb;
  nb.vals = nullptr; nb.base = mki3(0, 0, 0); nb.shift_limit = 0; nb.halo_sdf = nullptr; nb.halo_rgbw = nullptr; nb.halo_rim = 0; nb.r_vs = 0.f;
  return nb;
}

// table value of a block (kNbAbsent: not allocated): the workgroup's 27-block neighbourhood answers from LDS, anything
// else through the hash table
__device__ __forceinline__ u32 block_val(const Tab& t, const Neigh& nb, const i3 b) {
  if (nb.vals) {
    const int dx = b.x - nb.base.x, dy = b.y - nb.base.y, dz = b.z - nb.base.z;
    if ((u32) (dx + 1) < 3u && (u32) (dy + 1) < 3u && (u32) (dz + 1) < 3u) return nb.vals[(dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)];
  }
  u64 key;
  if (!pack_key(b, key)) return kNbAbsent;
  const int s = hash_find(t, key);
  return s >= 0 ? t.vals[s] : kNbAbsent;
}

// vds.cu:163-205 getVoxel(int3[, block_res]); coarse blocks are read with the writers' dense index
__device__ __forceinline__ VoxSample get_voxel_i(const Map& m, const Tab& t, const Neigh& nb, i3 v) {
  VoxSample r;
  r.sdf = 0.f; r.rgbw = 0; r.res = 0; r.found = false;
  if (nb.halo_sdf) {
    const int lx = v.x - nb.base.x * kBlockSide, ly = v.y - nb.base.y * kBlockSide, lz = v.z - nb.base.z * kBlockSide;
    const int rim = nb.halo_rim;
    if ((u32) (lx + rim) < (u32) (kBlockSide + 2 * rim) && (u32) (ly + rim) < (u32) (kBlockSide + 2 * rim) && (u32) (lz + rim) < (u32) (kBlockSide + 2 * rim)) {
      const u32 val = nb.vals[((lz >> 3) + 1) * 9 + ((ly >> 3) + 1) * 3 + ((lx >> 3) + 1)];
      if (val == kNbAbsent) return r;
      const int idx = ((lz + kHaloRim) * kHaloSide + (ly + kHaloRim)) * kHaloSide + (lx + kHaloRim);
      r.res = (val & kValCoarseBit) ? 1 : 0;
      r.found = true;
      r.sdf = nb.halo_sdf[idx];
      r.rgbw = nb.halo_rgbw[idx];
      return r;
    }
  }
  const int ax = v.x < 0 ? -v.x : v.x, ay = v.y < 0 ? -v.y : v.y, az = v.z < 0 ? -v.z : v.z;
  // voxel -> block is the shift below the limit (mrh_device.h)
  const i3 b = (u32) (ax | ay | az) < (u32) nb.shift_limit ? mki3(v.x >> 3, v.y >> 3, v.z >> 3) : voxel_to_block(v, m.vs);
  const u32 val = block_val(t, nb, b);
  if (val == kNbAbsent) return r;
  r.res = (val & kValCoarseBit) ? 1 : 0;
  r.found = true;
  const VoxPtr vp = vox_ptr(t, val);
  const u32 li = voxel_local_index(v, r.res);
  r.sdf = vp.sdf[li];
  r.rgbw = vp.rgbw[li];
  return r;
}
// vhu.cuh:143-151 worldPointToVirtualVoxelPos with the three divisions by the voxel size through one shared refined
// reciprocal (div_rr is bit-identical to the IEEE divide for these operands, mrh_device.h; a -0 quotient may come out as
// +0, which both sign(.) and the >= 0 test treat identically)
__device__ __forceinline__ i3 world_to_voxel_nb(const Neigh& nb, const float vs, const f3 pt) {
  if (nb.r_vs == 0.f) return world_to_voxel(vs, pt);
  const f3 p = mk3(div_rr(pt.x, vs, nb.r_vs), div_rr(pt.y, vs, nb.r_vs), div_rr(pt.z, vs, nb.r_vs));
  const float epsilon = 1e-5;
  f3 a = mk3(p.x + (float) signi(p.x) * 0.5f, p.y + (float) signi(p.y) * 0.5f, p.z + (float) signi(p.z) * 0.5f);
  a.x = (a.x >= 0) ? floorf(a.x + epsilon) : ceilf(a.x - epsilon);
  a.y = (a.y >= 0) ? floorf(a.y + epsilon) : ceilf(a.y - epsilon);
  a.z = (a.z >= 0) ? floorf(a.z + epsilon) : ceilf(a.z - epsilon);
  return mki3(f2i_hw(a.x), f2i_hw(a.y), f2i_hw(a.z));
}
__device__ __forceinline__ VoxSample get_voxel_f(const Map& m, const Tab& t, const Neigh& nb, f3 pos) { return get_voxel_i(m, t, nb, world_to_voxel_nb(nb, m.vs, pos)); }

// vds.cu:236-240 getVoxelSize(float3).  With a single resolution every block (and every miss) answers vs.
__device__ __forceinline__ float get_voxel_size_f(const Map& m, const Tab& t, const Neigh& nb, f3 pos) {
  if (!t.multi_res) return m.vs * (float) (1 << 0);
  i3 b;
  if (nb.halo_sdf) {  // staged workgroup: every voxel it can reach converts to its block by the arithmetic shift
    const i3 v = world_to_voxel_nb(nb, m.vs, pos);
    b = mki3(v.x >> 3, v.y >> 3, v.z >> 3);
  } else {
    b = world_to_block(m.vs, pos);
  }
  const u32 val = block_val(t, nb, b);
  const int res = (val != kNbAbsent && (val & kValCoarseBit)) ? 1 : 0;
  return m.vs * (float) (1 << res);
}

// vds.cu:260-338
__device__ __forceinline__ bool trilinear(const Map& m, const Tab& t, const Neigh& nb, f3 pos, float& dist) {
  const float voxel_size = get_voxel_size_f(m, t, nb, pos);
  const f3 pos_dual = mk3(pos.x - voxel_size * 0.5f, pos.y - voxel_size * 0.5f, pos.z - voxel_size * 0.5f);
  int base_resolution = 0;
  if (t.multi_res) {
    const u32 val = block_val(t, nb, world_to_block(voxel_size, pos));  // note: voxel_size, not vs (vds.cu:264)
    if (val != kNbAbsent) base_resolution = (val & kValCoarseBit) ? 1 : 0;
  }
  dist = 0.f;
  float pos_sdf = 0.f;
  if (t.multi_res) pos_sdf = get_voxel_f(m, t, nb, pos_dual).sdf;  // only consumed on resolution jumps
  const float x0 = pos_dual.x, y0 = pos_dual.y, z0 = pos_dual.z;
  float x1 = x0, y1 = y0, z1 = z0;
  float sdf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int dx = i & 1, dy = (i >> 1) & 1, dz = (i >> 2) & 1;
    const f3 vp = mk3(pos_dual.x + (float) dx * voxel_size, pos_dual.y + (float) dy * voxel_size, pos_dual.z + (float) dz * voxel_size);
    const VoxSample v = get_voxel_f(m, t, nb, vp);
    if ((v.rgbw >> 24) == 0) return false;
    if (v.res > base_resolution) {
      const float nvs = voxel_size * 2;
      const f3 np = mk3(pos.x - nvs * 0.5f + (float) dx * nvs, pos.y - nvs * 0.5f + (float) dy * nvs, pos.z - nvs * 0.5f + (float) dz * nvs);
      const float np_sdf = get_voxel_f(m, t, nb, np).sdf;
      const float alpha = 0.5f;
      sdf[i] = (1 - alpha) * pos_sdf + alpha * np_sdf;
    } else {
      sdf[i] = v.sdf;
    }
    if (vp.x > x1) x1 = vp.x;
    if (vp.y > y1) y1 = vp.y;
    if (vp.z > z1) z1 = vp.z;
  }
  const float ddx = (x1 - x0) > 1e-6f ? (pos.x - x0) / (x1 - x0) : 0.5f;
  const float ddy = (y1 - y0) > 1e-6f ? (pos.y - y0) / (y1 - y0) : 0.5f;
  const float ddz = (z1 - z0) > 1e-6f ? (pos.z - z0) / (z1 - z0) : 0.5f;
  const float c0 = sdf[0];
  const float c1 = (sdf[1] - sdf[0]);
  const float c2 = (sdf[2] - sdf[0]);
  const float c3 = (sdf[4] - sdf[0]);
  const float c4 = (sdf[3] - sdf[2] - sdf[1] + sdf[0]);
  const float c5 = (sdf[6] - sdf[4] - sdf[2] + sdf[0]);
  const float c6 = (sdf[5] - sdf[4] - sdf[1] + sdf[0]);
  const float c7 = (sdf[7] - sdf[6] - sdf[5] - sdf[3] + sdf[1] + sdf[4] + sdf[2] - sdf[0]);
  dist = c0 + c1 * ddx + c2 * ddy + c3 * ddz + c4 * ddx * ddy + c5 * ddy * ddz + c6 * ddx * ddz + c7 * ddx * ddy * ddz;
  return true;
}

// trilinearInterpolation (vds.cu:260-338) for corner `k` of the voxel at local coordinates (lx, ly, lz) of a staged FINE
// block, when the voxel's 3^3 cells lie in fine (or absent) blocks only, without converting the eight sample positions to voxels: the
// samples are pos_dual + {0, 1} * vs per axis with pos_dual = (pf +- vs / 2) - vs / 2, i.e. within a few ulp of the voxel
// centres k - 1, k (corner on the low side) or k, k + 1 (high side).  worldPointToVirtualVoxelPos rounds p / vs to the
// nearest integer (ties aside), so an error below 0.49 voxel cannot change the cell: with |voxel coordinate| < 2^18
// (the workgroup checks it) the accumulated rounding error of those few operations stays below 2^18 * 6 * 2^-24 = 0.1.
// The arithmetic on the positions (the interpolation weights) is the reference's, so the value is bit-identical; only
// WHICH cells are read is known beforehand.  The raw sample at the corner itself sits on a half-integer and keeps the
// literal conversion (get_voxel_f).
__device__ __forceinline__ bool trilinear_known(const Map& m, const Neigh& nb, const f3 pos, const int k, const int lx, const int ly, const int lz,
                                                float& dist) {
  const float voxel_size = m.vs * (float) (1 << 0);
  const f3 pos_dual = mk3(pos.x - voxel_size * 0.5f, pos.y - voxel_size * 0.5f, pos.z - voxel_size * 0.5f);
  const int bx = lx - ((k & 1) ? 0 : 1), by = ly - ((k & 2) ? 0 : 1), bz = lz - ((k & 4) ? 0 : 1);
  const int idx0 = ((bz + kHaloRim) * kHaloSide + (by + kHaloRim)) * kHaloSide + (bx + kHaloRim);
  dist = 0.f;
  float sdf[8];
  u32 wmin = 0xFFFFFFFFu;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int o = idx0 + (i & 1) + ((i >> 1) & 1) * kHaloSide + ((i >> 2) & 1) * kHaloSide * kHaloSide;
    sdf[i] = nb.halo_sdf[o];
    wmin = umin_(wmin, nb.halo_rgbw[o] >> 24);
  }
  if (wmin == 0u) return false;  // vds.cu:283-284: a sample without weight
  const float x0 = pos_dual.x, y0 = pos_dual.y, z0 = pos_dual.z;
  float x1 = x0, y1 = y0, z1 = z0;
  {  // the running maximum over the eight sample positions (vds.cu:311-316): the dx = dy = dz = 1 sample dominates
    const float xa = pos_dual.x + 0.f * voxel_size, xb = pos_dual.x + 1.f * voxel_size;
    const float ya = pos_dual.y + 0.f * voxel_size, yb = pos_dual.y + 1.f * voxel_size;
    const float za = pos_dual.z + 0.f * voxel_size, zb = pos_dual.z + 1.f * voxel_size;
    if (xa > x1) x1 = xa;
    if (xb > x1) x1 = xb;
    if (ya > y1) y1 = ya;
    if (yb > y1) y1 = yb;
    if (za > z1) z1 = za;
    if (zb > z1) z1 = zb;
  }
  const float ddx = (x1 - x0) > 1e-6f ? (pos.x - x0) / (x1 - x0) : 0.5f;
  const float ddy = (y1 - y0) > 1e-6f ? (pos.y - y0) / (y1 - y0) : 0.5f;
  const float ddz = (z1 - z0) > 1e-6f ? (pos.z - z0) / (z1 - z0) : 0.5f;
  const float c0 = sdf[0];
  const float c1 = (sdf[1] - sdf[0]);
  const float c2 = (sdf[2] - sdf[0]);
  const float c3 = (sdf[4] - sdf[0]);
  const float c4 = (sdf[3] - sdf[2] - sdf[1] + sdf[0]);
  const float c5 = (sdf[6] - sdf[4] - sdf[2] + sdf[0]);
  const float c6 = (sdf[5] - sdf[4] - sdf[1] + sdf[0]);
  const float c7 = (sdf[7] - sdf[6] - sdf[5] - sdf[3] + sdf[1] + sdf[4] + sdf[2] - sdf[0]);
  dist = c0 + c1 * ddx + c2 * ddy + c3 * ddz + c4 * ddx * ddy + c5 * ddy * ddz + c6 * ddx * ddz + c7 * ddx * ddy * ddz;
  return true;
}

// trilinearInterpolation (vds.cu:260-338) for corner `k` of the COARSE voxel `v` (index in its 4^3 block) of a staged block,
// when the voxel's 3^3 fine-cell neighbourhood lies in coarse blocks only (the caller checks it), with the sample positions as
// INTEGERS.  In fine-voxel units the voxel sits at V (even coordinates), the corner at Pk = V + (+-1, +-1, +-1): the corner's
// block is coarse, so voxel_size = 2 vs and pos_dual = Pk - 1; the eight samples are pos_dual + {0, 2}, the coarser re-sample of a
// resolution jump (vds.cu:296-309: nvs = 4 vs) reaches Pk - 2 + {0, 4} — all lattice points, within three cells of the voxel,
// which worldPointToVirtualVoxelPos (round to nearest) cannot miss while the accumulated rounding error of the position
// arithmetic stays below 0.1 voxel (|coordinate| < 2^18, checked by the workgroup: the argument of trilinear_known).  WHICH
// cells are read is all that is known beforehand: found / resolution / weight / sdf of every sample, the data-dependent branches
// and the float arithmetic of the weights are the reference's.  The base-resolution look-up converts the corner with the
// SAMPLED voxel size (vds.cu:264), which lands on a half-integer of the 2 vs lattice: it keeps the literal conversion.
__device__ __forceinline__ bool trilinear_coarse_known(const Map& m, const Tab& t, const Neigh& nb, const f3 pos, const int k, const int v, float& dist) {
  const i3 P = mki3(nb.base.x * kBlockSide + 2 * (v & 3) + ((k & 1) ? 1 : -1), nb.base.y * kBlockSide + 2 * ((v >> 2) & 3) + ((k & 2) ? 1 : -1),
                    nb.base.z * kBlockSide + 2 * (v >> 4) + ((k & 4) ? 1 : -1));
  const float voxel_size = m.vs * (float) (1 << 1);
  const f3 pos_dual = mk3(pos.x - voxel_size * 0.5f, pos.y - voxel_size * 0.5f, pos.z - voxel_size * 0.5f);
  int base_resolution = 0;
  {
    const u32 val = block_val(t, nb, world_to_block(voxel_size, pos));  // note: voxel_size, not vs (vds.cu:264)
    if (val != kNbAbsent) base_resolution = (val & kValCoarseBit) ? 1 : 0;
  }
  dist = 0.f;
  const float pos_sdf = get_voxel_i(m, t, nb, mki3(P.x - 1, P.y - 1, P.z - 1)).sdf;  // only consumed on resolution jumps
  const float x0 = pos_dual.x, y0 = pos_dual.y, z0 = pos_dual.z;
  float x1 = x0, y1 = y0, z1 = z0;
  float sdf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int dx = i & 1, dy = (i >> 1) & 1, dz = (i >> 2) & 1;
    const f3 vp = mk3(pos_dual.x + (float) dx * voxel_size, pos_dual.y + (float) dy * voxel_size, pos_dual.z + (float) dz * voxel_size);
    const VoxSample s = get_voxel_i(m, t, nb, mki3(P.x - 1 + 2 * dx, P.y - 1 + 2 * dy, P.z - 1 + 2 * dz));
    if ((s.rgbw >> 24) == 0) return false;
    if (s.res > base_resolution) {
      const float np_sdf = get_voxel_i(m, t, nb, mki3(P.x - 2 + 4 * dx, P.y - 2 + 4 * dy, P.z - 2 + 4 * dz)).sdf;
      const float alpha = 0.5f;
      sdf[i] = (1 - alpha) * pos_sdf + alpha * np_sdf;
    } else {
      sdf[i] = s.sdf;
    }
    if (vp.x > x1) x1 = vp.x;
    if (vp.y > y1) y1 = vp.y;
    if (vp.z > z1) z1 = vp.z;
  }
  const float ddx = (x1 - x0) > 1e-6f ? (pos.x - x0) / (x1 - x0) : 0.5f;
  const float ddy = (y1 - y0) > 1e-6f ? (pos.y - y0) / (y1 - y0) : 0.5f;
  const float ddz = (z1 - z0) > 1e-6f ? (pos.z - z0) / (z1 - z0) : 0.5f;
  const float c0 = sdf[0];
  const float c1 = (sdf[1] - sdf[0]);
  const float c2 = (sdf[2] - sdf[0]);
  const float c3 = (sdf[4] - sdf[0]);
  const float c4 = (sdf[3] - sdf[2] - sdf[1] + sdf[0]);
  const float c5 = (sdf[6] - sdf[4] - sdf[2] + sdf[0]);
  const float c6 = (sdf[5] - sdf[4] - sdf[1] + sdf[0]);
  const float c7 = (sdf[7] - sdf[6] - sdf[5] - sdf[3] + sdf[1] + sdf[4] + sdf[2] - sdf[0]);
  dist = c0 + c1 * ddx + c2 * ddy + c3 * ddz + c4 * ddx * ddy + c5 * ddy * ddz + c6 * ddx * ddz + c7 * ddx * ddy * ddz;
  return true;
}

// mesh_extractor.cu:6-36
__device__ __forceinline__ mrh_vertex vertex_interp(f3 p1, f3 p2, float d1, float d2, u32 c1, u32 c2) {
  const float isolevel = 0.f;
  const int r1 = c1 & 0xFF, g1 = (c1 >> 8) & 0xFF, b1 = (c1 >> 16) & 0xFF;
  const int r2 = c2 & 0xFF, g2 = (c2 >> 8) & 0xFF, b2 = (c2 >> 16) & 0xFF;
  mrh_vertex v;
  if (fabsf(isolevel - d1) < 0.00001f || (!(fabsf(isolevel - d2) < 0.00001f) && fabsf(d1 - d2) < 0.00001f)) {
    v.p[0] = p1.x; v.p[1] = p1.y; v.p[2] = p1.z;
    v.c[0] = (float) r1 / 255.f; v.c[1] = (float) g1 / 255.f; v.c[2] = (float) b1 / 255.f;
    return v;
  }
  if (fabsf(isolevel - d2) < 0.00001f) {
    v.p[0] = p2.x; v.p[1] = p2.y; v.p[2] = p2.z;
    v.c[0] = (float) r2 / 255.f; v.c[1] = (float) g2 / 255.f; v.c[2] = (float) b2 / 255.f;
    return v;
  }
  const float mu = (isolevel - d1) / (d2 - d1);
  v.p[0] = p1.x + mu * (p2.x - p1.x);
  v.p[1] = p1.y + mu * (p2.y - p1.y);
  v.p[2] = p1.z + mu * (p2.z - p1.z);
  v.c[0] = (float) r1 + mu * (float) (r2 - r1) / 255.f;
  v.c[1] = (float) g1 + mu * (float) (g2 - g1) / 255.f;
  v.c[2] = (float) b1 + mu * (float) (b2 - b1) / 255.f;
  return v;
}

// marching_cubes.cu:72-261 for one voxel.  Returns the triangle count; with EMIT writes triangles j < max_out to out[j].
template <bool EMIT>
__device__ __forceinline__ int mc_voxel(const Map& m, const Tab& t, const Neigh& nb, f3 pf, mrh_triangle* out, const int max_out = 5) {
  const float vvs = get_voxel_size_f(m, t, nb, pf);
  const float P = vvs * 0.5f;
  const float M = -P;
  f3 sP = mk3(P * 1.f, P * 1.f, P * 1.f);
  f3 sM = mk3(M * 1.f, M * 1.f, M * 1.f);
  if (t.multi_res) {
    // marching_cubes.cu:7-69 checkVertexVoxels
    float vs;
    vs = get_voxel_size_f(m, t, nb, mk3(pf.x + sP.x, pf.y + 0.0f, pf.z + 0.0f));
    if (vs > 0 && vs < 1 && vs != vvs) sP.x *= 0.499f;
    vs = get_voxel_size_f(m, t, nb, mk3(pf.x + sM.x, pf.y + 0.0f, pf.z + 0.0f));
    if (vs > 0 && vs < 1 && vs != vvs) sM.x *= 0.499f;
    vs = get_voxel_size_f(m, t, nb, mk3(pf.x + 0.0f, pf.y + sP.y, pf.z + 0.0f));
    if (vs > 0 && vs < 1 && vs != vvs) sP.y *= 0.499f;
    vs = get_voxel_size_f(m, t, nb, mk3(pf.x + 0.0f, pf.y + sM.y, pf.z + 0.0f));
    if (vs > 0 && vs < 1 && vs != vvs) sM.y *= 0.499f;
    vs = get_voxel_size_f(m, t, nb, mk3(pf.x + 0.0f, pf.y + 0.0f, pf.z + sP.z));
    if (vs > 0 && vs < 1 && vs != vvs) sP.z *= 0.499f;
    vs = get_voxel_size_f(m, t, nb, mk3(pf.x + 0.0f, pf.y + 0.0f, pf.z + sM.z));
    if (vs > 0 && vs < 1 && vs != vvs) sM.z *= 0.499f;
  }
  f3 p[8];
  float dist[8];
  u32 col[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    p[k] = mk3(pf.x + ((k & 1) ? sP.x : sM.x), pf.y + ((k & 2) ? sP.y : sM.y), pf.z + ((k & 4) ? sP.z : sM.z));
    const bool valid = trilinear(m, t, nb, p[k], dist[k]);
    const VoxSample v = get_voxel_f(m, t, nb, p[k]);
    col[k] = v.rgbw;
    if (!valid) {
      if ((int) (v.rgbw >> 24) < m.min_weight_threshold) return 0;
      dist[k] = v.sdf;
    }
  }
  u32 cube = 0;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (dist[k] < 0.f) cube |= (1u << k);
  const float thr = m.mc_threshold;
#pragma unroll
  for (int k = 0; k < 8; k++)
#pragma unroll
    for (int l = 0; l < 8; l++) {
      if (dist[k] * dist[l] < 0.f) {
        if (fabsf(dist[k]) + fabsf(dist[l]) > thr) return 0;
      } else {
        if (fabsf(dist[k] - dist[l]) > thr) return 0;
      }
    }
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (fabsf(dist[k]) > thr) return 0;
  const uint8_t* row = d_mc_tri[cube];
  const int ntri = row[0];
  if (EMIT) {
    for (int j = 0; j < ntri && j < max_out; j++)
      for (int k = 0; k < 3; k++) {
        const int code = row[1 + 3 * j + k];
        const int a = code >> 4, b = code & 0xF;
        out[j].v[k] = vertex_interp(p[a], p[b], dist[a], dist[b], col[a], col[b]);
      }
  }
  return ntri;
}

// marching_cubes.cu:72-261 for one voxel by EIGHT adjacent lanes, lane `k` = corner k (bit 0: +x, bit 1: +y, bit 2: +z; the
// corner numbers of the edge codes), `gb` = first lane of the group.  Each lane evaluates one trilinear corner value and
// one raw sample; the cube index, the early returns (a corner without a usable value, a jump above the threshold) and the
// triangle count are formed with ballots and shuffles inside the group — the same pure function of the same 8 corner
// values as the sequential mc_voxel, so the same result.  With EMIT the up to 15 vertices of the voxel are interpolated
// by the 8 lanes (two each) and stored straight into out[0 .. min(ntri, room)).  Must be called by all 8 lanes of a group
// (inactive groups pass active = false and take part in the ballots with neutral values).
// the up to 15 vertices of one voxel by its 8 lanes (two slots each): lane `k` holds corner k's value and colour, `row` = the
// voxel's triangle-table row, triangles j < room are stored to out[j] (marching_cubes.cu:205-261)
__device__ __forceinline__ void mc_emit_vertices(const f3 pf, const f3 sP, const f3 sM, const float dist, const u32 col, const uint8_t* row,
                                                 const int ntri, const int k, const int gb, mrh_triangle* out, const int room) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int sidx = k + 8 * h;  // vertex slot: triangle sidx / 3, corner sidx % 3
    const bool mine = sidx < 3 * ntri;
    const int code = mine ? (int) row[1 + sidx] : 0;
    const int a = code >> 4, b = code & 0xF;
    const float da = __shfl(dist, gb + a), db = __shfl(dist, gb + b);
    const u32 ca = (u32) __shfl((int) col, gb + a), cb = (u32) __shfl((int) col, gb + b);
    const f3 pa = mk3(pf.x + ((a & 1) ? sP.x : sM.x), pf.y + ((a & 2) ? sP.y : sM.y), pf.z + ((a & 4) ? sP.z : sM.z));
    const f3 pb = mk3(pf.x + ((b & 1) ? sP.x : sM.x), pf.y + ((b & 2) ? sP.y : sM.y), pf.z + ((b & 4) ? sP.z : sM.z));
    if (mine && sidx / 3 < room) out[sidx / 3].v[sidx % 3] = vertex_interp(pa, pb, da, db, ca, cb);
  }
}

// corner record of the count pass (k_mc_emit_records): this lane's corner value and colour, and the six checkVertexVoxels flags
struct McCorner {
  float dist;
  u32 col, flags;
};
constexpr int kMcLiteral = 0, kMcFineKnown = 1, kMcCoarseKnown = 2;
template <bool EMIT>
// `mode` (uniform): kMcLiteral, kMcFineKnown (trilinear_known) or kMcCoarseKnown (trilinear_coarse_known)
__device__ __forceinline__ int mc_group(const Map& m, const Tab& t, const Neigh& nb, const f3 pf, const int v, const int mode,
                                        const int k, const int gb, const bool active, mrh_triangle* out, const int room, McCorner* rec = nullptr) {
  const bool stencil_known = mode != kMcLiteral;
  // stencil_known (a fine voxel whose 3^3 cells lie in fine or absent blocks): the voxel's own size is the fine one, and the six
  // checkVertexVoxels probes — half a voxel along an axis: the voxel itself or its neighbour on that axis — all answer the fine
  // size (a missing block reads resolution 0, vds.cu:236-240), so no flag can be raised: nothing to look up
  // (the same holds for a coarse voxel whose 3^3 fine cells lie in coarse blocks: it and its probes read the coarse size)
  const float vvs = mode == kMcFineKnown ? m.vs * (float) (1 << 0) : (mode == kMcCoarseKnown ? m.vs * (float) (1 << 1) : get_voxel_size_f(m, t, nb, pf));
  const float P = vvs * 0.5f;
  const float M = -P;
  f3 sP = mk3(P * 1.f, P * 1.f, P * 1.f);
  f3 sM = mk3(M * 1.f, M * 1.f, M * 1.f);
  u32 vflags = 0;
  if (t.multi_res && !stencil_known) {
    // marching_cubes.cu:7-69 checkVertexVoxels: six independent tests, lane j < 6 takes test j (+x, -x, +y, -y, +z, -z)
    bool flag = false;
    if (k < 6) {
      const float o = (k & 1) ? M : P;
      const f3 q = mk3(pf.x + ((k >> 1) == 0 ? o : 0.0f), pf.y + ((k >> 1) == 1 ? o : 0.0f), pf.z + ((k >> 1) == 2 ? o : 0.0f));
      const float vs = get_voxel_size_f(m, t, nb, q);
      flag = vs > 0 && vs < 1 && vs != vvs;
    }
    const u32 flags = (u32) (__ballot(flag) >> gb) & 0x3Fu;
    vflags = flags;
    if (flags & 1u) sP.x *= 0.499f;
    if (flags & 2u) sM.x *= 0.499f;
    if (flags & 4u) sP.y *= 0.499f;
    if (flags & 8u) sM.y *= 0.499f;
    if (flags & 16u) sP.z *= 0.499f;
    if (flags & 32u) sM.z *= 0.499f;
  }
  const f3 p = mk3(pf.x + ((k & 1) ? sP.x : sM.x), pf.y + ((k & 2) ? sP.y : sM.y), pf.z + ((k & 4) ? sP.z : sM.z));
  float dist = 0.f;
  const bool valid = mode == kMcFineKnown ? trilinear_known(m, nb, p, k, v & 7, (v >> 3) & 7, v >> 6, dist)
                     : (mode == kMcCoarseKnown ? trilinear_coarse_known(m, t, nb, p, k, v, dist) : trilinear(m, t, nb, p, dist));
  // the raw sample at the corner: a lattice point for a coarse voxel (V +- 1), a half-integer for a fine one (literal conversion)
  const VoxSample vs_ = mode == kMcCoarseKnown
                            ? get_voxel_i(m, t, nb, mki3(nb.base.x * kBlockSide + 2 * (v & 3) + ((k & 1) ? 1 : -1), nb.base.y * kBlockSide + 2 * ((v >> 2) & 3) + ((k & 2) ? 1 : -1),
                                                         nb.base.z * kBlockSide + 2 * (v >> 4) + ((k & 4) ? 1 : -1)))
                            : get_voxel_f(m, t, nb, p);
  const u32 col = vs_.rgbw;
  bool bad = false;
  if (!valid) {
    if ((int) (vs_.rgbw >> 24) < m.min_weight_threshold) bad = true;
    else dist = vs_.sdf;
  }
  const u32 badmask = (u32) (__ballot(bad && active) >> gb) & 0xFFu;
  const u32 cube = (u32) (__ballot(dist < 0.f) >> gb) & 0xFFu;
  const float thr = m.mc_threshold;
  bool fail = fabsf(dist) > thr;
#pragma unroll
  for (int l = 0; l < 8; l++) {
    const float dl = __shfl(dist, gb + l);
    if (dist * dl < 0.f) fail = fail || (fabsf(dist) + fabsf(dl) > thr);
    else fail = fail || (fabsf(dist - dl) > thr);
  }
  const u32 failmask = (u32) (__ballot(fail && active) >> gb) & 0xFFu;
  const uint8_t* row = d_mc_tri[cube];
  const int ntri = (!active || badmask || failmask) ? 0 : (int) row[0];
  if (!EMIT && rec) { rec->dist = dist; rec->col = col; rec->flags = vflags | (vvs != m.vs * (float) (1 << 0) ? 64u : 0u); }
  if (EMIT) mc_emit_vertices(pf, sP, sM, dist, col, row, ntri, k, gb, out, room);
  return ntri;
}

// packed keys of a block list: key order == (x, y, z) lexicographic order, the canonical order of the extraction
__global__ __launch_bounds__(256) void k_list_keys(const int4* __restrict__ list, const int n, u64* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 key = ~0ull;  // a listed block always packs (it came out of the table)
  pack_key(mki3(list[i].x, list[i].y, list[i].z), key);
  keys[i] = key;
}

// The block list in canonical order (packed key order == (x, y, z) order) without a sort library: a list has 10^3 .. 10^4 blocks
// and every key is distinct, so a block's place is the number of keys below its own.  Grid (blocks / 256, slices of
// kRankSlice keys): a workgroup counts, for its 256 blocks, the keys of one slice (staged in LDS, read as broadcasts) that
// are smaller, and adds that to the block's rank; k_block_scatter then moves the entries.  n^2 comparisons — 5 x 10^7 for the
// 7 000 blocks of a 640 x 480 room, a few microseconds of the whole chip — against rocPRIM's radix sort of 64-bit keys, which
// for a list this small is one workgroup working for 74 us (profiles/r04/driver_cmd_mc_kernel_stats.csv).  Lists beyond
// kRankSortMax take the radix sort of mrh_sort.h.
constexpr int kRankSlice = 512;
constexpr int kRankSortMax = 32768;
// the slice's keys are the same for every lane: they come through the scalar cache (uniform index into the packed-key array of
// k_list_keys) and the loop body is a 64-bit compare against a scalar and an add — the first version staged them in LDS and
// was bound by the LDS return path (a 512-byte broadcast per compare: 21 us for 14 k blocks); packing them inside the loop
// instead costs ~20 scalar instructions per key (133 us)
__global__ __launch_bounds__(256) void k_block_rank(const u64* __restrict__ keys, const int n, u32* __restrict__ partial) {
  const int s0 = blockIdx.y * kRankSlice, s1 = min(n, s0 + kRankSlice);
  const int i = blockIdx.x * 256 + threadIdx.x;
  const u64 key = i < n ? keys[i] : 0ull;
  u32 below = 0;
#pragma unroll 16
  for (int j = s0; j < s1; j++) below += keys[j] < key ? 1u : 0u;
  if (i < n) partial[(size_t) blockIdx.y * n + i] = below;  // one row per slice: plain stores, nothing to clear, no atomics
}
__global__ __launch_bounds__(256) void k_block_scatter(const int4* __restrict__ list, const int n, const u32* __restrict__ partial, const int slices,
                                                       int4* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u32 rank = 0;
  for (int sl = 0; sl < slices; sl++) rank += partial[(size_t) sl * n + i];
  sorted[rank] = list[i];
}

// per-block triangle counts -> exact offsets, the total and the record demand for the host: one workgroup.  The counts go
// through LDS a tile at a time (coalesced loads; the first version gave every thread a contiguous piece of the list in global
// memory: 2 x 14 dependent loads per thread, 17.6 us for 14 k blocks), every thread scans eight neighbours of the tile, the
// tiles chain through a running carry (a list of any length: 10^6 blocks are 122 tiles).
constexpr int kScanTile = 8192;
__global__ __launch_bounds__(1024) void k_mc_scan_total(const u32* __restrict__ counts, const int n, u64* __restrict__ offsets,
                                                        const u32* __restrict__ rec_ctr, u64* __restrict__ total) {
  __shared__ u32 s_c[kScanTile];
  __shared__ u32 s_w[16];
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 carry = 0;
  for (int base = 0; base < n; base += kScanTile) {
    const int m = min(kScanTile, n - base);
#pragma unroll
    for (int k = 0; k < kScanTile / 1024; k++) {
      const int i = k * 1024 + (int) threadIdx.x;
      s_c[i] = i < m ? counts[base + i] : 0u;
    }
    __syncthreads();
    u32 v[kScanTile / 1024], mine = 0;  // a block has at most 512 x 5 triangles: a tile's sum fits 32 bits
#pragma unroll
    for (int k = 0; k < kScanTile / 1024; k++) { v[k] = s_c[threadIdx.x * (kScanTile / 1024) + k]; mine += v[k]; }
    u32 incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 o = __shfl_up(incl, off);
      if ((int) lane >= off) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    u64 run = carry + (incl - mine);
    u32 all = 0;
    for (u32 w = 0; w < 16; w++) { if (w < wave) run += s_w[w]; all += s_w[w]; }
#pragma unroll
    for (int k = 0; k < kScanTile / 1024; k++) {
      const int i = base + (int) threadIdx.x * (kScanTile / 1024) + k;
      if (i < n) offsets[i] = run;
      run += v[k];
    }
    carry += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    total[0] = carry;
    total[1] = rec_ctr ? ((u64) rec_ctr[0] | ((u64) (rec_ctr[1] & 1u) << 63)) : (1ull << 63);  // records asked for | did not fit
  }
}

// One workgroup (256 threads) per block of the sorted list.
//   1. the 27 surrounding blocks are resolved once (table value or absent)                              -> s_nb
//   2. every voxel marching cubes can sample for this block is staged in LDS ({sdf, rgbw} per fine cell) -> halo
//   3. COUNT pass: class of every staged cell as a one-hot bit (1: weighted and clearly positive, 2: weighted and clearly
//      negative, 4: never observed, 8: anything else) and an OR over the (2w + 1)^3 window of each voxel: everything marching
//      cubes evaluates for a voxel — the eight trilinear corner values, the coarser re-samples they blend in on a resolution
//      jump, or the raw sample a corner falls back to — is a convex combination of, or a sample from, cells of that window
//      (fp32 evaluation error < 2e-5 x the largest magnitude).
//        w = 1, a fine voxel whose 3^3 cells lie in fine (or absent) blocks: no re-sample can occur, so a cell without weight
//        never contributes a VALUE (it invalidates the stencil, and a raw sample below min_weight_threshold ends the voxel
//        without a triangle).  Without an "anything else" cell and without both signs among the weighted cells every corner
//        that has a value has the same sign: cube index 0 or 255, no triangle — the voxel is not evaluated.  This also drops
//        the two shells where the observed band ends (weighted cells of one sign next to unseen ones), which made up most
//        of what the one-class rule of the first version let through (4.03 M candidates for 1.47 M triangles).
//        w = 3, a coarse voxel or a fine one next to a coarse block (Neigh's comment derives the reach): a re-sample blends
//        the sdf of a possibly UNWEIGHTED cell in (vds.cu:268, :296-309); such a cell reads +-0 unless its weight was
//        starved away (class bit 16), and a zero cannot turn the sign of the clearly-signed value it is averaged with:
//        the same rule holds with bit 16 counted as "anything else" (argument at the test in the kernel).
//      "Clearly" = 1e-3 x sdf_bound <= |sdf| <= 1.001 x sdf_bound, sdf_bound = the largest truncation a sample can carry;
//      anything outside (or NaN) is class 8.
//      EMIT pass: the candidates are the voxels the count pass found non-empty (per_voxel).
//   4. the candidates are COMPACTED (LDS list) and evaluated densely, EIGHT lanes per voxel (one per cube corner,
//      mc_group), through the staged cells: a lookup is the reference's float position -> voxel conversion, a subtraction
//      and two LDS reads — no hash, no global gather
//   5. block-wide exclusive scan of the per-voxel triangle counts in voxel order: counts[e] (COUNT) / the exact offset
//      of every voxel's triangles (EMIT) -> canonical (block, voxel, triangle) order, no atomics.
// Blocks too far from the origin for voxel -> block to be the arithmetic shift skip 2-3 and evaluate every voxel through
// the neighbour table / the hash (the literal path).
// does the 3^3 cell neighbourhood of fine voxel v (local index) touch a coarse block?  cmask: bit i = neighbour block i is
// coarse.  Per axis the cells lie in the own block, plus the previous / next one at local coordinate 0 / 7.
__device__ __forceinline__ bool voxel_touches_coarse(const int v, const u32 cmask) {
  if (cmask == 0u) return false;
  const int x = v & 7, y = (v >> 3) & 7, z = v >> 6;
  // per axis: which of the three neighbour columns (previous, own, next) the 3 cells fall into; bit i of the product set =
  // neighbour (bz, by, bx) with i = bz * 9 + by * 3 + bx, the numbering of s_nb / cmask
  const u32 tx = 2u | (x == 0 ? 1u : 0u) | (x == 7 ? 4u : 0u);
  const u32 ty = 2u | (y == 0 ? 1u : 0u) | (y == 7 ? 4u : 0u);
  const u32 tz = 2u | (z == 0 ? 1u : 0u) | (z == 7 ? 4u : 0u);
  const u32 plane = ((ty & 1u) ? tx : 0u) | (tx << 3) | ((ty & 4u) ? tx << 6 : 0u);
  const u32 cube = ((tz & 1u) ? plane : 0u) | (plane << 9) | ((tz & 4u) ? plane << 18 : 0u);
  return (cube & cmask) != 0u;
}

// Class planes of the prescreen: one byte per staged cell, rows of 16 bytes (cell lx of a row at byte lx + kHaloRim; bytes 14,
// 15 and whatever lies outside the staged rim are masked after the load), so that a row is ONE 16-byte LDS read and the OR over
// a window along x is three (seven) byte shifts of the 128-bit row — the first version read every cell of every voxel's window
// as a byte (27 reads per voxel on the narrow window, 21 per output cell on the separable wide one: a fifth of the count pass).
constexpr int kClsRow = 16;
constexpr int kClsBytes = kHaloSide * kHaloSide * kClsRow;  // 3136
struct Row128 { u64 lo, hi; };
__device__ __forceinline__ Row128 row_load(const uint8_t* base, const int row) {
  const uint4 r = *(const uint4*) (base + (size_t) row * kClsRow);
  Row128 o;
  o.lo = (u64) r.x | ((u64) r.y << 32);
  o.hi = (u64) r.z | ((u64) r.w << 32);
  return o;
}
__device__ __forceinline__ Row128 row_or(const Row128 a, const Row128 b) { Row128 o; o.lo = a.lo | b.lo; o.hi = a.hi | b.hi; return o; }
// OR of the row with itself shifted by 1 .. W bytes both ways: byte i of the result = OR of bytes i - W .. i + W
template <int W>
__device__ __forceinline__ Row128 row_spread(const Row128 r) {
  Row128 o = r;
#pragma unroll
  for (int k = 1; k <= W; k++) {
    o.lo |= (r.lo << (8 * k)) | (r.lo >> (8 * k)) | (r.hi << (64 - 8 * k));
    o.hi |= (r.hi << (8 * k)) | (r.hi >> (8 * k)) | (r.lo >> (64 - 8 * k));
  }
  return o;
}
// bytes kHaloRim .. kHaloRim + 7 of a row: the eight voxels x = 0 .. 7
__device__ __forceinline__ u64 row_voxels(const Row128 r) { return (r.lo >> (8 * kHaloRim)) | (r.hi << (64 - 8 * kHaloRim)); }

// do the 3^3 fine cells around COARSE voxel v (index in its 4^3 block; fine coordinates 2 * index) all lie in coarse blocks?
// The cells reach the previous block on an axis where the coordinate is 0 and never the next one (coordinate + 1 <= 7).
__device__ __forceinline__ bool coarse_voxel_in_coarse_cells(const int v, const u32 cmask) {
  const int x = v & 3, y = (v >> 2) & 3, z = v >> 4;
  const u32 tx = 2u | (x == 0 ? 1u : 0u), ty = 2u | (y == 0 ? 1u : 0u), tz = 2u | (z == 0 ? 1u : 0u);
  const u32 plane = ((ty & 1u) ? tx : 0u) | (tx << 3);
  const u32 cube = ((tz & 1u) ? plane : 0u) | (plane << 9);
  return (cube & ~cmask) == 0u;
}

// Staging by ROWS (round 4): a thread takes one (y, z) row of the staged region and loads it as 16-byte words — the eight
// cells inside the block's own x range (SEG 0: two words per plane from a fine block, one from a coarse block, whose four
// voxels of the row each fill two cells), the rim towards -x (SEG 1: the word holding cells 4..7 of the neighbour's row) and
// towards +x (SEG 2: cells 0..3) — instead of one cell per thread and round: 588 address computations and ~1 600 wide loads
// per block with a three-cell rim where the cell loop had 2 744 and 5 488 single-word loads, all in flight at once.
struct RowLoad {
  float4 a_sdf, b_sdf;
  uint4 a_rgbw, b_rgbw;
  u32 nval;
};
template <int SEG>
__device__ __forceinline__ void row_issue(const Tab& t, const u32* s_nb, const bool active, const int ly, const int lz, RowLoad& r) {
  r.a_sdf = r.b_sdf = make_float4(0.f, 0.f, 0.f, 0.f);
  r.a_rgbw = r.b_rgbw = make_uint4(0u, 0u, 0u, 0u);
  r.nval = kNbAbsent;
  if (!active) return;
  const u32 nval = s_nb[((lz >> 3) + 1) * 9 + ((ly >> 3) + 1) * 3 + (SEG == 0 ? 1 : (SEG == 1 ? 0 : 2))];
  r.nval = nval;
  if (nval == kNbAbsent) return;  // a missing block reads sdf 0, weight 0 (vds.cu:163-176)
  const VoxPtr vp = vox_ptr(t, nval);
  const int fy = ly & 7, fz = lz & 7;
  const bool cz = (nval & kValCoarseBit) != 0;
  const u32 off = cz ? (u32) ((fz >> 1) * 16 + (fy >> 1) * 4) : (u32) (fz * 64 + fy * 8 + (SEG == 1 ? 4 : 0));
  r.a_sdf = *(const float4*) (vp.sdf + off);
  r.a_rgbw = *(const uint4*) (vp.rgbw + off);
  if (SEG == 0 && !cz) {
    r.b_sdf = *(const float4*) (vp.sdf + off + 4);
    r.b_rgbw = *(const uint4*) (vp.rgbw + off + 4);
  }
}
template <int SEG, bool COUNT>
__device__ __forceinline__ void row_store(const RowLoad& r, const bool active, const int ly, const int lz, const int rim, const float lo, const float hi,
                                          const uint8_t unseen, float* s_sdf, u32* s_rgbw, uint8_t* s_cls) {
  if (!active) return;
  const bool cz = r.nval != kNbAbsent && (r.nval & kValCoarseBit) != 0;
  const float fa[8] = {r.a_sdf.x, r.a_sdf.y, r.a_sdf.z, r.a_sdf.w, r.b_sdf.x, r.b_sdf.y, r.b_sdf.z, r.b_sdf.w};
  const u32 ra[8] = {r.a_rgbw.x, r.a_rgbw.y, r.a_rgbw.z, r.a_rgbw.w, r.b_rgbw.x, r.b_rgbw.y, r.b_rgbw.z, r.b_rgbw.w};
  const bool wide_rim = rim == kHaloRim;  // uniform
  const int lx0 = SEG == 0 ? 0 : (SEG == 1 ? -rim : kBlockSide);
  const int row = (lz + kHaloRim) * kHaloSide + (ly + kHaloRim);
  const int base = row * kHaloSide + (lx0 + kHaloRim), cbase = row * kClsRow + (lx0 + kHaloRim);
#pragma unroll
  for (int j = 0; j < (SEG == 0 ? kBlockSide : kHaloRim); j++) {
    if (SEG != 0 && j >= rim) break;  // uniform
    // source word of cell j.  Fine block: the loaded cells are 0..7 (SEG 0), 4..7 (SEG 1; the rim's cells are 8 - rim .. 7) or
    // 0..3 (SEG 2).  Coarse block: the row's four voxels, voxel fx >> 1 for fine cell fx.
    float sf, sc;
    u32 rf, rc;
    if (SEG == 0) { sf = fa[j]; rf = ra[j]; sc = fa[j >> 1]; rc = ra[j >> 1]; }
    else if (SEG == 1) {
      // three-cell rim: cells 5, 6, 7; one-cell rim: cell 7 (j == 0 only)
      sf = wide_rim ? fa[1 + j] : fa[3]; rf = wide_rim ? ra[1 + j] : ra[3];
      sc = wide_rim ? fa[(5 + j) >> 1] : fa[3]; rc = wide_rim ? ra[(5 + j) >> 1] : ra[3];
    } else { sf = fa[j]; rf = ra[j]; sc = fa[j >> 1]; rc = ra[j >> 1]; }
    const float sv = cz ? sc : sf;
    const u32 rw = cz ? rc : rf;
    s_sdf[base + j] = sv;
    s_rgbw[base + j] = rw;
    if (COUNT) {
      uint8_t cls = unseen;
      if ((rw >> 24) != 0) cls = (sv >= lo && sv <= hi) ? 1 : ((sv <= -lo && sv >= -hi) ? 2 : 8);
      else if ((__float_as_uint(sv) & 0x7FFFFFFFu) != 0u) cls |= 16;  // unseen, but a non-zero sdf is stored (weight starved to 0)
      s_cls[cbase + j] = cls;
    }
  }
}

// The 27-block neighbourhood of every block of the sorted list, resolved by one thread per (block, neighbour): 27 independent
// probes per block at full occupancy instead of 27 lanes of one wave walking their probe paths at the head of every k_mc
// workgroup while the other 229 threads wait; both passes read the table.  Layout: nb[e * 32 + i], i = (dz+1)*9 + (dy+1)*3 + (dx+1).
constexpr int kMcNbStride = 32;
__global__ __launch_bounds__(256) void k_mc_neighbors(const Tab t, const int4* __restrict__ sorted, const int n, u32* __restrict__ nb) {
  const size_t g = (size_t) blockIdx.x * 256 + threadIdx.x;
  const size_t e = g >> 5;
  const int i = (int) (g & 31);
  if (e >= (size_t) n || i >= 27) return;
  const int4 ent = sorted[e];
  u32 val = kNbAbsent;
  if (i == 13) {
    val = (u32) ent.w;  // the block itself: the list carries its table value
  } else {
    const i3 b = mki3(ent.x + (i % 3) - 1, ent.y + ((i / 3) % 3) - 1, ent.z + (i / 9) - 1);
    u64 key;
    if (pack_key(b, key)) {
      const int slot = hash_find(t, key);
      if (slot >= 0) val = t.vals[slot];
    }
  }
  nb[e * kMcNbStride + i] = val;
}

// The count pass evaluates every candidate's eight corner values anyway; for the voxels that produce triangles it parks them
// (72 bytes a voxel) so that the emit pass is a flat interpolation of records instead of a second staging + evaluation:
//   record r = 18 words: [0..7] corner values, [8..15] corner colours (rgbw), [16] voxel | checkVertexVoxels flags << 16,
//   [17] triangles of the block's earlier voxels (patched in after the block's scan)
// Space is reserved per block by its candidate count (one atomic per block); a block that finds no room sets the overflow word
// and the host falls back to k_mc<emit> for the whole extraction (and grows the buffer for the next one).
constexpr u32 kMcNoRecords = 0xFFFFFFFFu;
constexpr int kMcRecWords = 18;
struct McRecords {
  u32* ctr;       // [0] records reserved so far (keeps counting past the capacity: the demand), [1] overflow flag
  u32* recs;      // [cap * kMcRecWords]
  u32* base;      // [n] first record of block e, or kMcNoRecords
  u32* count;     // [n] records of block e
  u32 cap;
};
constexpr int kMcThreads = 256;
// Which block a workgroup takes: its own id (flags bit 2, the default), or — MRH_MC_SLAB_LOG2=k, an experiment kept for the record
// (mrh_capi.hip: mrh_extract_triangles has the numbers) — the list cut into runs of 2^k blocks dealt to the eight XCDs in turn
// (workgroups go to the XCDs round-robin by id), the j-th workgroup of an XCD taking the j-th block of that XCD's runs, so that a
// block's neighbours are staged through the same L2.  The grid is then a multiple of 8 * 2^k: a permutation of the ids.
// flags: bit 2 = block e = workgroup id; bits 4..8 = k.
__device__ __forceinline__ int mc_first_block(const int flags) {
  const int w = (int) blockIdx.x, k = (flags >> 4) & 31;
  if ((flags & 4) || ((int) gridDim.x & ((8 << k) - 1))) return w;
  const int x = w & 7, j = w >> 3;
  return ((((j >> k) << 3) + x) << k) + (j & ((1 << k) - 1));
}
#ifdef MRH_MC_TRACE
// tuning builds only (tools/trace_mc.sh): shader-clock cycles of thread 0 per phase, one record per block and pass (no atomics:
// 14 k workgroups adding to the same few words would time the atomics).  Record: [0] class + 1 (0: fine block, no coarse
// neighbour; 1: fine next to coarse; 2: coarse), [1..4] staging, prescreen, known-stencil evaluation, literal evaluation,
// [5] whole block, [6] known candidates, [7] literal candidates
constexpr int kMcTraceBlocks = 65536;
__device__ unsigned int d_mc_trace[2][kMcTraceBlocks][8];
#define MRH_MC_TS(var) const long long var = (long long) clock64()
#define MRH_MC_ACC(slot, a, b) do { if (tid == 0 && e < kMcTraceBlocks) d_mc_trace[EMIT ? 1 : 0][e][(slot) + 1] += (unsigned int) ((b) - (a)); } while (0)
#define MRH_MC_ADD(slot, v) do { if (tid == 0 && e < kMcTraceBlocks) d_mc_trace[EMIT ? 1 : 0][e][slot] = (unsigned int) (v); } while (0)
#else
#define MRH_MC_TS(var) do { } while (0)
#define MRH_MC_ACC(slot, a, b) do { } while (0)
#define MRH_MC_ADD(slot, v) do { } while (0)
#endif
template <bool EMIT>
// 4 waves per SIMD: left alone the allocator takes 147 VGPRs (3 waves); capped at 128 it spills 8-24 bytes and both passes
// run 10-12 % faster — the staging half of the kernel is a chain of memory round trips and wants the extra workgroup per CU
// (5 waves / 96 VGPRs: 100 bytes of scratch, no better).
__global__ __launch_bounds__(kMcThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_mc(const Map m, const Tab t, const int4* __restrict__ sorted, const int n,
                                                   const u32* __restrict__ nb_table,
                                                   u32* __restrict__ counts, const u64* __restrict__ offsets,
                                                   mrh_triangle* __restrict__ out, const u64 max_tris, uint8_t* __restrict__ per_voxel,
                                                   const float sdf_bound, const int flag_overflow, const McRecords R) {
  __shared__ u32 s_nb[27];
  __shared__ float s_sdf[kHaloCells];
  __shared__ u32 s_rgbw[kHaloCells];
  __shared__ __attribute__((aligned(16))) uint8_t s_cls[kClsBytes];  // count pass: class of every staged cell (rows of kClsRow bytes)
  __shared__ u64 s_px[kHaloSide * kHaloSide];  // wide window: OR along x for the eight voxels of every staged (y, z) row
  __shared__ u64 s_py[kBlockSide * kHaloSide]; // ... then along y, for y in 0..7 and every staged z
  __shared__ u64 s_nar[kBlockSide * kBlockSide];  // per voxel (byte v of the array): OR over the 3^3 window
  __shared__ u64 s_wid[kBlockSide * kBlockSide];  // per voxel: OR over the 7^3 window
  __shared__ unsigned short s_cand[512];
  __shared__ uint8_t s_ntri[512];
  __shared__ u32 s_off[512];
  __shared__ u32 s_wave[kMcThreads / 64];
  __shared__ u32 s_ncand[2];  // [0] candidates with a known stencil (list grows from s_cand[0] up), [1] the others (from s_cand[511] down)
  __shared__ u32 s_rec[2];    // count pass: [0] first record of this block (kMcNoRecords: none), [1] records written
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  for (int e = mc_first_block(flag_overflow); e < n; e += gridDim.x) {
    MRH_MC_TS(ts0);
    const int4 ent = sorted[e];
    const u32 val = (u32) ent.w;
    const bool coarse = (val & kValCoarseBit) != 0;
    const int nvox = coarse ? kCoarseVoxels : kBlockVoxels;
    Neigh nb = neigh_none();
    nb.base = mki3(ent.x, ent.y, ent.z);
    nb.shift_limit = m.block_shift_limit;
    nb.r_vs = rcp_refined(m.vs);
    if (tid < 27) s_nb[tid] = nb_table[(size_t) e * kMcNbStride + tid];  // the 27 surrounding blocks (k_mc_neighbors)
    if (tid < 2) s_ncand[tid] = 0;
    for (int i = tid; i < 512; i += kMcThreads) s_ntri[i] = 0;
    __syncthreads();
    nb.vals = s_nb;
    // sharded maps: halo blocks imported from other ranks are read by the lookups but emit nothing themselves
    const bool mine = owns_block(m, mki3(ent.x, ent.y, ent.z));  // uniform
    const int amax = max(max(abs(ent.x), abs(ent.y)), abs(ent.z));
    const bool staged = mine && (amax + 3) * kBlockSide < m.block_shift_limit;  // uniform: every reachable voxel converts by shift
    u32 cmask = 0;  // bit i: neighbour i is a coarse block (uniform)
    if (t.multi_res)
      for (int i = 0; i < 27; i++) cmask |= (s_nb[i] != kNbAbsent && (s_nb[i] & kValCoarseBit)) ? (1u << i) : 0u;
    // cells to stage around the block: the count pass classifies up to 3 cells out when a coarse block is near, the emit
    // pass only needs what its few candidates read (anything farther falls back to the neighbour table)
    const int rim = (EMIT ? coarse : cmask != 0u) ? kHaloRim : 1;  // uniform
#ifdef MRH_MC_TRACE
    const int tr_cls = coarse ? 2 : (cmask ? 1 : 0);
#endif
    if (staged) {
      const int side = kBlockSide + 2 * rim;
      const bool wide_rim = rim == kHaloRim;  // uniform
      const bool row_active = tid < side * side;
      const int ry = wide_rim ? tid % kHaloSide : tid % (kBlockSide + 2), rz = wide_rim ? tid / kHaloSide : tid / (kBlockSide + 2);
      const int ly = ry - rim, lz = rz - rim;  // the thread's row, relative to the block
      RowLoad r0, r1, r2;  // all loads of the row in flight before the first LDS store
      row_issue<0>(t, s_nb, row_active, ly, lz, r0);
      row_issue<1>(t, s_nb, row_active, ly, lz, r1);
      row_issue<2>(t, s_nb, row_active, ly, lz, r2);
      const float lo = 1e-3f * sdf_bound, hi = 1.001f * sdf_bound;
      // third class: a cell without an observation (weight 0, or no block).  If a voxel's whole window is unseen, every
      // corner's trilinear stencil meets a weight-0 sample (vds.cu:283-284: invalid) and the raw sample it falls back to has
      // weight 0 < min_weight_threshold (marching_cubes.cu:89-93: return): no triangle.  Needs min_weight_threshold >= 1.
      const uint8_t unseen = m.min_weight_threshold >= 1 ? 4 : 8;  // 8 = "anything else": always a candidate
      row_store<0, !EMIT>(r0, row_active, ly, lz, rim, lo, hi, unseen, s_sdf, s_rgbw, s_cls);
      row_store<1, !EMIT>(r1, row_active, ly, lz, rim, lo, hi, unseen, s_sdf, s_rgbw, s_cls);
      row_store<2, !EMIT>(r2, row_active, ly, lz, rim, lo, hi, unseen, s_sdf, s_rgbw, s_cls);
      __syncthreads();
      nb.halo_sdf = s_sdf;
      nb.halo_rgbw = s_rgbw;
      nb.halo_rim = rim;
    }
    MRH_MC_TS(ts1);
    // a fine voxel whose 3^3 cells lie in fine (or absent) blocks has a known trilinear stencil (trilinear_known)
    const bool fine_known = staged && !coarse && (amax + 2) * kBlockSide < (1 << 18);  // uniform
    // ---- candidates, in two groups so that a wave evaluates voxels of ONE kind: the known-stencil evaluation is ~10x shorter
    // than the literal one, and a wave that holds a single literal voxel pays for both
    // ... and so has a coarse voxel whose 3^3 fine cells lie in coarse blocks (trilinear_coarse_known): the front list of a coarse block
    const bool coarse_known = staged && coarse && (amax + 2) * kBlockSide < (1 << 18) && !(flag_overflow & 2);  // uniform; bit 1: MRH_MC_NO_COARSE_KNOWN (A/B, tests)
    const int known_mode = coarse ? kMcCoarseKnown : kMcFineKnown;
    auto push = [&](const int v) {
      if ((fine_known && !voxel_touches_coarse(v, cmask)) || (coarse_known && coarse_voxel_in_coarse_cells(v, cmask))) s_cand[atomicAdd(&s_ncand[0], 1u)] = (unsigned short) v;
      else s_cand[511u - atomicAdd(&s_ncand[1], 1u)] = (unsigned short) v;
    };
    if (!mine) {
      // nothing to evaluate
    } else if (EMIT) {
      for (int v = tid; v < nvox; v += kMcThreads)
        if (per_voxel[(size_t) e * 512 + v] != 0) push(v);
    } else if (staged && sdf_bound > 0.f) {
      // OR of the one-hot class bits over every voxel's window, a row of eight voxels at a time (row_load / row_spread):
      // the 3^3 window of a fine block by threads 192..255 (nine row reads each), the 7^3 window — coarse voxels and fine ones
      // next to a coarse block — separably: along x for all 14 x 14 staged rows (threads 0..195, next to the narrow rows),
      // then y, then z
      const bool any_wide = coarse || cmask != 0u;  // uniform
      if (!coarse && tid >= 192) {
        const int r = tid - 192, y = r & 7, z = r >> 3;
        Row128 acc;
        acc.lo = acc.hi = 0;
#pragma unroll
        for (int dz = -1; dz <= 1; dz++)
#pragma unroll
          for (int dy = -1; dy <= 1; dy++) acc = row_or(acc, row_load(s_cls, (z + dz + kHaloRim) * kHaloSide + (y + dy + kHaloRim)));
        // only the cells lx = -1 .. 8 (bytes 2 .. 11) are part of these windows — and, with a one-cell rim, the only ones staged
        acc.lo &= 0xFFFFFFFFFFFF0000ull;
        acc.hi &= 0x00000000FFFFFFFFull;
        s_nar[r] = row_voxels(row_spread<1>(acc));
      }
      if (any_wide && tid < kHaloSide * kHaloSide) {
        Row128 r = row_load(s_cls, tid);
        r.hi &= 0x0000FFFFFFFFFFFFull;  // bytes 14, 15 are not cells
        s_px[tid] = row_voxels(row_spread<kHaloRim>(r));
      }
      __syncthreads();
      if (any_wide) {
        if (tid < kBlockSide * kHaloSide) {
          const int y = tid & 7, zz = tid >> 3;
          u64 acc = 0;
#pragma unroll
          for (int d = -kHaloRim; d <= kHaloRim; d++) acc |= s_px[zz * kHaloSide + (y + kHaloRim + d)];
          s_py[tid] = acc;
        }
        __syncthreads();
        if (tid < kBlockSide * kBlockSide) {
          const int y = tid & 7, z = tid >> 3;
          u64 acc = 0;
#pragma unroll
          for (int d = -kHaloRim; d <= kHaloRim; d++) acc |= s_py[(z + kHaloRim + d) * kBlockSide + y];
          s_wid[tid] = acc;
        }
        __syncthreads();
      }
      const uint8_t* nar8 = (const uint8_t*) s_nar;
      const uint8_t* wid8 = (const uint8_t*) s_wid;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int v = tid + h * kMcThreads;
        if (v < nvox) {
          bool empty;
          if (coarse || voxel_touches_coarse(v, cmask)) {
            int x, y, z;
            if (!coarse) { x = v & 7; y = (v >> 3) & 7; z = v >> 6; }
            else { x = 2 * (v & 3); y = 2 * ((v >> 2) & 3); z = 2 * (v >> 4); }
            const u32 acc = wid8[(z * kBlockSide + y) * kBlockSide + x];
            // A resolution jump blends the sdf of a cell in WITHOUT looking at its weight (vds.cu:268, :296-309: 0.5 pos_sdf +
            // 0.5 np_sdf), so here an unseen cell can contribute a value — but only its stored sdf, and that is +-0 unless the
            // weight was starved away (class bit 16; a missing block reads 0, vds.cu:163-176).  Every SAMPLE of a stencil must
            // still carry weight (vds.cu:283-284), and pos_sdf is sample 0.  Without class 8 / 16 and without both signs in the
            // window: a valid stencil averages (the dual cell is centred on the corner: every weight is 1/2) eight values that
            // are each a clearly-signed sdf or the mean of one and a zero — all of one sign, magnitude >= lo / 2, against an
            // evaluation error < 2e-5 sdf_bound; an invalid one falls back to the weighted raw sample (same sign) or ends the
            // voxel.  Eight corners of one sign: cube index 0 or 255, no triangle.
            empty = (acc & (8u | 16u)) == 0u && (acc & 3u) != 3u;
          } else {
            const u32 acc = nar8[v];
            // no resolution jump can occur: every corner value is a convex combination of WEIGHTED cells (a sample without
            // weight invalidates the stencil, vds.cu:283-284) or a weighted raw sample — or the voxel returns without a
            // triangle (raw sample below min_weight_threshold).  Unseen cells therefore never contribute a value: without
            // an "anything else" cell and without BOTH signs in the window every corner that has a value has the same sign.
            empty = (acc & 8u) == 0u && (acc & 3u) != 3u;
          }
          if (!empty) push(v);
        }
      }
    } else {
      for (int v = tid; v < nvox; v += kMcThreads) push(v);
    }
    __syncthreads();
    const int ncand[2] = {(int) s_ncand[0], (int) s_ncand[1]};
    if (!EMIT && R.recs) {  // room for one record per candidate
      if (tid == 0) {
        const u32 want = (u32) (ncand[0] + ncand[1]);
        u32 first = kMcNoRecords;
        if (want) {
          first = atomicAdd(&R.ctr[0], want);
          if (first > R.cap || want > R.cap - first) { first = kMcNoRecords; atomicOr(&R.ctr[1], 1u); }
        }
        s_rec[0] = first;
        s_rec[1] = 0;
      }
      __syncthreads();
    }
    MRH_MC_TS(ts2);
    auto voxel_position = [&](const int v) {
      i3 pi;
      if (!coarse) pi = mki3(ent.x * kBlockSide + (v & 7), ent.y * kBlockSide + ((v >> 3) & 7), ent.z * kBlockSide + (v >> 6));
      else pi = mki3(ent.x * kBlockSide + 2 * (v & 3), ent.y * kBlockSide + 2 * ((v >> 2) & 3), ent.z * kBlockSide + 2 * (v >> 4));
      return voxel_to_world(m.vs, pi);
    };
    const int gb = lane & ~7, corner = lane & 7;
    unsigned short* s_recv = (unsigned short*) s_px;  // voxel of record slot i (the prescreen's planes are done with)
    if (!EMIT) {
      // ---- dense evaluation of the candidates, 8 lanes each; per-voxel counts to LDS
#pragma unroll 1
      for (int kind = 0; kind < 2; kind++) {
#pragma unroll 1
        for (int base = 0; base < ncand[kind] * 8; base += kMcThreads) {
          const int i = base + tid;
          const bool active = i < ncand[kind] * 8;
          const int ci = active ? (i >> 3) : 0;
          const int v = s_cand[kind == 0 ? ci : 511 - ci];
          McCorner cr;
          const int ntri = mc_group<false>(m, t, nb, voxel_position(v), v, kind == 0 ? known_mode : kMcLiteral, corner, gb, active, nullptr, 0, &cr);
          if (active && corner == 0) s_ntri[v] = (uint8_t) ntri;
          if (R.recs && ntri > 0 && s_rec[0] != kMcNoRecords) {  // group-uniform
            u32 slot = 0;
            if (corner == 0) slot = atomicAdd(&s_rec[1], 1u);
            slot = (u32) __shfl((int) slot, gb);
            u32* r = R.recs + (size_t) (s_rec[0] + slot) * kMcRecWords;
            r[corner] = __float_as_uint(cr.dist);
            r[8 + corner] = cr.col;
            if (corner == 0) { r[16] = (u32) v | (cr.flags << 16); s_recv[slot] = (unsigned short) v; }
          }
        }
#ifdef MRH_MC_TRACE
        if (kind == 0) { __syncthreads(); MRH_MC_TS(tsk); MRH_MC_ACC(2, ts2, tsk); MRH_MC_ACC(3, 0, -tsk); }
#endif
      }
      __syncthreads();
#ifdef MRH_MC_TRACE
      { MRH_MC_TS(tsl); MRH_MC_ACC(3, 0, tsl); }
#endif
    }
    // ---- block-wide exclusive scan of the per-voxel counts, 2 voxels per thread, in voxel order
    const int v0 = 2 * tid;
    u32 c0, c1;
    if (EMIT) { c0 = per_voxel[(size_t) e * 512 + v0]; c1 = per_voxel[(size_t) e * 512 + v0 + 1]; }
    else {
      c0 = s_ntri[v0]; c1 = s_ntri[v0 + 1];
      *(unsigned short*) (per_voxel + (size_t) e * 512 + v0) = (unsigned short) (c0 | (c1 << 8));
    }
    u32 incl = c0 + c1;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    u32 wave_off = 0, total = 0;
    for (int i = 0; i < kMcThreads / 64; i++) { if (i < (tid >> 6)) wave_off += s_wave[i]; total += s_wave[i]; }
    if (!EMIT) {
      if (tid == 0) counts[e] = total;
      if (R.recs) {
        const u32 ex0 = wave_off + incl - (c0 + c1);
        s_off[v0] = ex0;
        s_off[v0 + 1] = ex0 + c0;
        __syncthreads();
        const u32 first = s_rec[0], nrec = s_rec[1];
        if (first != kMcNoRecords)
          for (u32 i = tid; i < nrec; i += kMcThreads) R.recs[(size_t) (first + i) * kMcRecWords + 17] = s_off[s_recv[i]];
        if (tid == 0) { R.base[e] = first; R.count[e] = first != kMcNoRecords ? nrec : 0u; }
      }
    } else {
      // exclusive offsets of this thread's two voxels, parked in LDS for the groups that evaluate them
      const u32 ex0 = wave_off + incl - (c0 + c1);
      s_off[v0] = ex0;
      s_off[v0 + 1] = ex0 + c0;
      __syncthreads();
#pragma unroll 1
      for (int kind = 0; kind < 2; kind++) {
#pragma unroll 1
        for (int base = 0; base < ncand[kind] * 8; base += kMcThreads) {
          const int i = base + tid;
          const bool active = i < ncand[kind] * 8;
          const int ci = active ? (i >> 3) : 0;
          const int v = s_cand[kind == 0 ? ci : 511 - ci];
          const u64 first = offsets[e] + s_off[v];
          const int room = first >= max_tris ? 0 : (int) (max_tris - first < 5 ? max_tris - first : 5);
          const int ntri = mc_group<true>(m, t, nb, voxel_position(v), v, kind == 0 ? known_mode : kMcLiteral, corner, gb, active, out + first, room);  // straight to the exact offset
          if ((flag_overflow & 1) && active && corner == 0 && ntri > room) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TRI);
        }
#ifdef MRH_MC_TRACE
        if (kind == 0) { __syncthreads(); MRH_MC_TS(tsk); MRH_MC_ACC(2, ts2, tsk); MRH_MC_ACC(3, 0, -tsk); }
#endif
      }
#ifdef MRH_MC_TRACE
      { __syncthreads(); MRH_MC_TS(tsl); MRH_MC_ACC(3, 0, tsl); }
#endif
    }
    __syncthreads();
#ifdef MRH_MC_TRACE
    {
      MRH_MC_TS(ts5);
      MRH_MC_ACC(0, ts0, ts1); MRH_MC_ACC(1, ts1, ts2); MRH_MC_ADD(5, ts5 - ts0);
      MRH_MC_ADD(0, tr_cls + 1); MRH_MC_ADD(6, ncand[0]); MRH_MC_ADD(7, ncand[1]);
    }
#endif
  }
}

// The emit pass over the records of the count pass: eight lanes per record, no staging, no evaluation.  Block e's records are
// R.base[e] .. + R.count[e]; a record's triangles start at offsets[e] + word 17 — the canonical (block, voxel, triangle) order of
// k_mc<emit>, whose vertex arithmetic (mc_emit_vertices on the recorded corner values) this shares.
__global__ __launch_bounds__(kMcThreads) void k_mc_emit_records(const Map m, const Tab t, const int4* __restrict__ sorted, const int n, const McRecords R,
                                                               const u64* __restrict__ offsets, mrh_triangle* __restrict__ out, const u64 max_tris,
                                                               const int flag_overflow) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int gb = lane & ~7, corner = lane & 7;
  for (int e = mc_first_block(flag_overflow); e < n; e += gridDim.x) {
    const u32 first_rec = R.base[e], nrec = R.count[e];
    if (first_rec == kMcNoRecords || nrec == 0) continue;  // uniform
    const int4 ent = sorted[e];
    const bool coarse = ((u32) ent.w & kValCoarseBit) != 0;
    const u64 block_first = offsets[e];
    for (u32 base = 0; base < nrec * 8u; base += kMcThreads) {
      const u32 i = base + tid;
      const bool active = i < nrec * 8u;
      const u32* r = R.recs + (size_t) (first_rec + (active ? (i >> 3) : 0u)) * kMcRecWords;
      const float dist = __uint_as_float(r[corner]);
      const u32 col = r[8 + corner];
      const u32 w16 = r[16], voff = r[17];
      const int v = (int) (w16 & 0xFFFFu);
      const u32 flags = w16 >> 16;
      i3 pi;
      if (!coarse) pi = mki3(ent.x * kBlockSide + (v & 7), ent.y * kBlockSide + ((v >> 3) & 7), ent.z * kBlockSide + (v >> 6));
      else pi = mki3(ent.x * kBlockSide + 2 * (v & 3), ent.y * kBlockSide + 2 * ((v >> 2) & 3), ent.z * kBlockSide + 2 * (v >> 4));
      const f3 pf = voxel_to_world(m.vs, pi);
      const float vvs = m.vs * (float) (1 << ((flags & 64u) ? 1 : 0));
      const float P = vvs * 0.5f;
      const float M = -P;
      f3 sP = mk3(P * 1.f, P * 1.f, P * 1.f);
      f3 sM = mk3(M * 1.f, M * 1.f, M * 1.f);
      if (flags & 1u) sP.x *= 0.499f;
      if (flags & 2u) sM.x *= 0.499f;
      if (flags & 4u) sP.y *= 0.499f;
      if (flags & 8u) sM.y *= 0.499f;
      if (flags & 16u) sP.z *= 0.499f;
      if (flags & 32u) sM.z *= 0.499f;
      const u32 cube = (u32) (__ballot(dist < 0.f) >> gb) & 0xFFu;
      const uint8_t* row = d_mc_tri[cube];
      const int ntri = active ? (int) row[0] : 0;
      const u64 first = block_first + voff;
      const int room = first >= max_tris ? 0 : (int) (max_tris - first < 5 ? max_tris - first : 5);
      mc_emit_vertices(pf, sP, sM, dist, col, row, ntri, corner, gb, out + first, room);
      if ((flag_overflow & 1) && active && corner == 0 && ntri > room) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TRI);
    }
  }
}

}  // namespace mrh
