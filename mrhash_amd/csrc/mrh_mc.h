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
};
__device__ __forceinline__ Neigh neigh_none() {
  Neigh nb;
  nb.vals = nullptr; nb.base = mki3(0, 0, 0); nb.shift_limit = 0; nb.halo_sdf = nullptr; nb.halo_rgbw = nullptr; nb.halo_rim = 0;
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
__device__ __forceinline__ VoxSample get_voxel_f(const Map& m, const Tab& t, const Neigh& nb, f3 pos) { return get_voxel_i(m, t, nb, world_to_voxel(m.vs, pos)); }

// vds.cu:236-240 getVoxelSize(float3).  With a single resolution every block (and every miss) answers vs.
__device__ __forceinline__ float get_voxel_size_f(const Map& m, const Tab& t, const Neigh& nb, f3 pos) {
  if (!t.multi_res) return m.vs * (float) (1 << 0);
  const u32 val = block_val(t, nb, world_to_block(m.vs, pos));
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

// packed keys of a block list: key order == (x, y, z) lexicographic order, the canonical order of the extraction
__global__ __launch_bounds__(256) void k_list_keys(const int4* __restrict__ list, const int n, u64* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 key = ~0ull;  // a listed block always packs (it came out of the table)
  pack_key(mki3(list[i].x, list[i].y, list[i].z), key);
  keys[i] = key;
}

// One workgroup (256 threads) per block of the sorted list.
//   1. the 27 surrounding blocks are resolved once (table value or absent)                              -> s_nb
//   2. every voxel marching cubes can sample for this block is staged in LDS ({sdf, rgbw} per fine cell) -> halo
//   3. COUNT pass: sign class of every staged cell (1: weighted and clearly positive, 2: weighted and clearly negative,
//      0: anything else) and a separable AND over the (2w + 1)^3 window of each voxel (w = 1 on a single-resolution
//      neighbourhood, 3 otherwise): everything marching cubes evaluates for a voxel — the eight trilinear corner values,
//      the coarser re-samples they blend in on a resolution jump, or the raw sample a corner falls back to — is a convex
//      combination of, or a sample from, cells of that window (fp32 evaluation error < 2e-5 x the largest magnitude), so
//      if all of them share one class every corner has that sign, the cube index is 0 or 255 and the voxel has no
//      triangle: it is not evaluated at all.  "Clearly" = 1e-3 x sdf_bound <= |sdf| <= 1.001 x sdf_bound, sdf_bound =
//      the largest truncation a sample can carry; anything outside (or NaN) is class 0.
//      EMIT pass: the candidates are the voxels the count pass found non-empty (per_voxel).
//   4. the candidates are COMPACTED (LDS list) and evaluated densely, one lane each, through the staged cells: a lookup
//      is the reference's float position -> voxel conversion, a subtraction and two LDS reads — no hash, no global gather
//   5. block-wide exclusive scan of the per-voxel triangle counts in voxel order: counts[e] (COUNT) / the exact offset
//      of every voxel's triangles (EMIT) -> canonical (block, voxel, triangle) order, no atomics.
// Blocks too far from the origin for voxel -> block to be the arithmetic shift skip 2-3 and evaluate every voxel through
// the neighbour table / the hash (the literal path).
constexpr int kMcThreads = 256;
template <bool EMIT>
__global__ __launch_bounds__(kMcThreads) void k_mc(const Map m, const Tab t, const int4* __restrict__ sorted, const int n,
                                                   u32* __restrict__ counts, const u64* __restrict__ offsets,
                                                   mrh_triangle* __restrict__ out, const u64 max_tris, uint8_t* __restrict__ per_voxel,
                                                   const float sdf_bound) {
  __shared__ u32 s_nb[27];
  __shared__ float s_sdf[kHaloCells];
  __shared__ u32 s_rgbw[kHaloCells];
  __shared__ uint8_t s_cls[2][kHaloCells];
  __shared__ unsigned short s_cand[512];
  __shared__ uint8_t s_ntri[512];
  __shared__ u32 s_wave[kMcThreads / 64];
  __shared__ u32 s_ncand;
  const int tid = threadIdx.x;
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    const int4 ent = sorted[e];
    const u32 val = (u32) ent.w;
    const bool coarse = (val & kValCoarseBit) != 0;
    const int nvox = coarse ? kCoarseVoxels : kBlockVoxels;
    Neigh nb = neigh_none();
    nb.base = mki3(ent.x, ent.y, ent.z);
    nb.shift_limit = m.block_shift_limit;
    if (tid < 27) {  // resolve the 27 surrounding blocks once
      const i3 b = mki3(ent.x + (tid % 3) - 1, ent.y + ((tid / 3) % 3) - 1, ent.z + (tid / 9) - 1);
      u64 key;
      int slot = -1;
      if (pack_key(b, key)) slot = hash_find(t, key);
      s_nb[tid] = slot >= 0 ? t.vals[slot] : kNbAbsent;
    }
    if (tid == 0) s_ncand = 0;
    for (int i = tid; i < 512; i += kMcThreads) s_ntri[i] = 0;
    __syncthreads();
    nb.vals = s_nb;
    // sharded maps: halo blocks imported from other ranks are read by the lookups but emit nothing themselves
    const bool mine = owns_block(m, mki3(ent.x, ent.y, ent.z));  // uniform
    const int amax = max(max(abs(ent.x), abs(ent.y)), abs(ent.z));
    const bool staged = mine && (amax + 3) * kBlockSide < m.block_shift_limit;  // uniform: every reachable voxel converts by shift
    u32 any_coarse = coarse ? kValCoarseBit : 0u;
    if (t.multi_res)
      for (int i = 0; i < 27; i++) any_coarse |= (s_nb[i] != kNbAbsent) ? (s_nb[i] & kValCoarseBit) : 0u;
    const int rim = any_coarse ? kHaloRim : 1;  // uniform
    if (staged) {
      const int side = kBlockSide + 2 * rim;
      for (int c = tid; c < side * side * side; c += kMcThreads) {
        const int lx = c % side - rim, ly = (c / side) % side - rim, lz = c / (side * side) - rim;  // fine cell relative to the block
        const u32 nval = s_nb[((lz >> 3) + 1) * 9 + ((ly >> 3) + 1) * 3 + ((lx >> 3) + 1)];
        float sv = 0.f;
        u32 rw = 0;
        if (nval != kNbAbsent) {
          const VoxPtr vp = vox_ptr(t, nval);
          const int fx = lx & 7, fy = ly & 7, fz = lz & 7;
          const u32 li = (nval & kValCoarseBit) ? (u32) ((fz >> 1) * 16 + (fy >> 1) * 4 + (fx >> 1)) : (u32) (fz * 64 + fy * 8 + fx);
          sv = vp.sdf[li];
          rw = vp.rgbw[li];
        }
        const int idx = ((lz + kHaloRim) * kHaloSide + (ly + kHaloRim)) * kHaloSide + (lx + kHaloRim);
        s_sdf[idx] = sv;
        s_rgbw[idx] = rw;
        if (!EMIT) {
          const float lo = 1e-3f * sdf_bound, hi = 1.001f * sdf_bound;
          uint8_t cls = 0;
          if ((rw >> 24) != 0) cls = (sv >= lo && sv <= hi) ? 1 : ((sv <= -lo && sv >= -hi) ? 2 : 0);
          s_cls[0][idx] = cls;
        }
      }
      __syncthreads();
      nb.halo_sdf = s_sdf;
      nb.halo_rgbw = s_rgbw;
      nb.halo_rim = rim;
    }
    // ---- candidates
    if (!mine) {
      // nothing to evaluate
    } else if (EMIT) {
      for (int v = tid; v < nvox; v += kMcThreads)
        if (per_voxel[(size_t) e * 512 + v] != 0) s_cand[atomicAdd(&s_ncand, 1u)] = (unsigned short) v;
    } else if (staged && sdf_bound > 0.f) {
      // separable AND of the sign classes over the window [-w, w]^3 (w = rim): x, then y, then z
      const int w = rim;
      for (int c = tid; c < kBlockSide * kHaloSide * kHaloSide; c += kMcThreads) {  // x in 0..7, all staged y, z
        const int x = c & 7, yz = c >> 3;
        const int base = yz * kHaloSide + (x + kHaloRim);
        u32 acc = 3u;
        for (int d = -w; d <= w; d++) acc &= s_cls[0][base + d];
        s_cls[1][base] = (uint8_t) acc;
      }
      __syncthreads();
      for (int c = tid; c < kBlockSide * kBlockSide * kHaloSide; c += kMcThreads) {  // x, y in 0..7, all staged z
        const int x = c & 7, y = (c >> 3) & 7, z = c >> 6;
        const int base = (z * kHaloSide + (y + kHaloRim)) * kHaloSide + (x + kHaloRim);
        u32 acc = 3u;
        for (int d = -w; d <= w; d++) acc &= s_cls[1][base + d * kHaloSide];
        s_cls[0][base] = (uint8_t) acc;
      }
      __syncthreads();
      for (int v = tid; v < nvox; v += kMcThreads) {
        int x, y, z;
        if (!coarse) { x = v & 7; y = (v >> 3) & 7; z = v >> 6; }
        else { x = 2 * (v & 3); y = 2 * ((v >> 2) & 3); z = 2 * (v >> 4); }
        const int base = ((z + kHaloRim) * kHaloSide + (y + kHaloRim)) * kHaloSide + (x + kHaloRim);
        u32 acc = 3u;
        for (int d = -w; d <= w; d++) acc &= s_cls[0][base + d * kHaloSide * kHaloSide];
        if (acc == 0u) s_cand[atomicAdd(&s_ncand, 1u)] = (unsigned short) v;  // not all positive and not all negative
      }
    } else {
      for (int v = tid; v < nvox; v += kMcThreads) s_cand[atomicAdd(&s_ncand, 1u)] = (unsigned short) v;
    }
    __syncthreads();
    const int ncand = (int) s_ncand;
    if (!EMIT) {
      // ---- dense evaluation of the candidates; per-voxel counts to LDS
      for (int i = tid; i < ncand; i += kMcThreads) {
        const int v = s_cand[i];
        i3 pi;
        if (!coarse) pi = mki3(ent.x * kBlockSide + (v & 7), ent.y * kBlockSide + ((v >> 3) & 7), ent.z * kBlockSide + (v >> 6));
        else pi = mki3(ent.x * kBlockSide + 2 * (v & 3), ent.y * kBlockSide + 2 * ((v >> 2) & 3), ent.z * kBlockSide + 2 * (v >> 4));
        s_ntri[v] = (uint8_t) mc_voxel<false>(m, t, nb, voxel_to_world(m.vs, pi), nullptr);
      }
      __syncthreads();
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
      if ((int) lane_id() >= off) incl += o;
    }
    if (lane_id() == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    u32 wave_off = 0, total = 0;
    for (int i = 0; i < kMcThreads / 64; i++) { if (i < (tid >> 6)) wave_off += s_wave[i]; total += s_wave[i]; }
    if (!EMIT) {
      if (tid == 0) counts[e] = total;
    } else {
      // exclusive offsets of this thread's two voxels, parked in LDS for the lanes that evaluate them
      u32* s_off = (u32*) s_cls;  // 512 x u32 = 2 KiB of the class arrays (unused by the emit pass)
      const u32 ex0 = wave_off + incl - (c0 + c1);
      s_off[v0] = ex0;
      s_off[v0 + 1] = ex0 + c0;
      __syncthreads();
      for (int i = tid; i < ncand; i += kMcThreads) {
        const int v = s_cand[i];
        i3 pi;
        if (!coarse) pi = mki3(ent.x * kBlockSide + (v & 7), ent.y * kBlockSide + ((v >> 3) & 7), ent.z * kBlockSide + (v >> 6));
        else pi = mki3(ent.x * kBlockSide + 2 * (v & 3), ent.y * kBlockSide + 2 * ((v >> 2) & 3), ent.z * kBlockSide + 2 * (v >> 4));
        const u64 base = offsets[e] + s_off[v];
        const int room = base >= max_tris ? 0 : (int) (max_tris - base < 5 ? max_tris - base : 5);
        const int ntri = mc_voxel<true>(m, t, nb, voxel_to_world(m.vs, pi), out + base, room);  // straight to the exact offset
        if (ntri > room) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TRI);
      }
    }
    __syncthreads();
  }
}

}  // namespace mrh
