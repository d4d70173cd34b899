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
struct Neigh {
  const u32* vals;  // LDS [27], index (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1); nullptr: no table (lookups outside k_mc)
  i3 base;          // block position of the workgroup's block
  int shift_limit;  // Map::block_shift_limit
};

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

// marching_cubes.cu:72-261 for one voxel.  Returns the triangle count; with EMIT writes them to out[0..n).
template <bool EMIT>
__device__ __forceinline__ int mc_voxel(const Map& m, const Tab& t, const Neigh& nb, f3 pf, mrh_triangle* out) {
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
    for (int j = 0; j < ntri; j++)
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

// Workgroup per block of the sorted list, lane = voxel.  EMIT = false: counts[e] = triangles of block e and
// per_voxel[e * 512 + v] = triangles of voxel v.  EMIT = true: triangles written at offsets[e] + (exclusive prefix over
// voxel index); voxels the count pass found empty (the vast majority) are not evaluated a second time.
template <bool EMIT>
__global__ __launch_bounds__(512) void k_mc(const Map m, const Tab t, const int4* __restrict__ sorted, const int n,
                                            u32* __restrict__ counts, const u64* __restrict__ offsets,
                                            mrh_triangle* __restrict__ out, const u64 max_tris, uint8_t* __restrict__ per_voxel,
                                            const float sdf_bound) {
  __shared__ u32 s_wave[8];
  __shared__ u32 s_nb[27];
  // Count pass, fine blocks with no coarse neighbour: sign class of the 10^3 cells around the block (1: weighted and clearly positive,
  // 2: weighted and clearly negative, 0: anything else).  Everything marching cubes evaluates for a voxel — the eight
  // trilinear corner values, or the raw sample a corner falls back to — is built from the 3^3 cells around it, the
  // trilinear value is a convex combination of them (fp32 evaluation error < 2e-5 x the largest magnitude), so if all 27
  // are of one class every corner has that sign, the cube index is 0 or 255 and the voxel has no triangle: it is not
  // evaluated at all.  "Clearly" = 1e-3 x sdf_bound <= |sdf| <= 1.001 x sdf_bound, sdf_bound = the largest truncation a
  // sample can carry; anything outside (or NaN) is class 0 and takes the full path.
  __shared__ uint8_t s_cls[1000];
  const int v = threadIdx.x;
  const int wave = v >> 6;
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    const int4 ent = sorted[e];
    const u32 val = (u32) ent.w;
    const bool coarse = (val & kValCoarseBit) != 0;
    Neigh nb;
    nb.vals = nullptr;
    nb.base = mki3(ent.x, ent.y, ent.z);
    nb.shift_limit = m.block_shift_limit;
    {  // resolve the 27 surrounding blocks once
      if (v < 27) {
        const i3 b = mki3(ent.x + (v % 3) - 1, ent.y + ((v / 3) % 3) - 1, ent.z + (v / 9) - 1);
        u64 key;
        int slot = -1;
        if (pack_key(b, key)) slot = hash_find(t, key);
        s_nb[v] = slot >= 0 ? t.vals[slot] : kNbAbsent;
      }
      __syncthreads();
      nb.vals = s_nb;
    }
    const int amax = max(max(abs(ent.x), abs(ent.y)), abs(ent.z));
    bool prescreen = !EMIT && !coarse && sdf_bound > 0.f && (amax + 2) * kBlockSide < m.block_shift_limit;  // uniform
    if (prescreen && t.multi_res) {  // a fine block whose whole neighbourhood is fine (or absent) is evaluated exactly as on a single-resolution map
      u32 any_coarse = 0;
      for (int i = 0; i < 27; i++) any_coarse |= (s_nb[i] != kNbAbsent) ? (s_nb[i] & kValCoarseBit) : 0u;
      prescreen = any_coarse == 0u;
    }
    if (prescreen) {
      const float lo = 1e-3f * sdf_bound, hi = 1.001f * sdf_bound;
      for (int cidx = v; cidx < 1000; cidx += 512) {
        const int lx = cidx % 10 - 1, ly = (cidx / 10) % 10 - 1, lz = cidx / 100 - 1;  // voxel coordinates relative to the block, -1 .. 8
        const int bx = lx < 0 ? 0 : (lx > 7 ? 2 : 1), by = ly < 0 ? 0 : (ly > 7 ? 2 : 1), bz = lz < 0 ? 0 : (lz > 7 ? 2 : 1);
        const u32 nval = s_nb[bz * 9 + by * 3 + bx];
        uint8_t cls = 0;
        if (nval != kNbAbsent) {
          const VoxPtr vp = vox_ptr(t, nval);
          const u32 li = (u32) ((lz & 7) * 64 + (ly & 7) * 8 + (lx & 7));
          const float sv = vp.sdf[li];
          if ((vp.rgbw[li] >> 24) != 0) cls = (sv >= lo && sv <= hi) ? 1 : ((sv <= -lo && sv >= -hi) ? 2 : 0);
        }
        s_cls[cidx] = cls;
      }
      __syncthreads();
    }
    int ntri = 0;
    mrh_triangle tris[5];
    const bool skip = EMIT && per_voxel[(size_t) e * 512 + v] == 0;
    // sharded maps: halo blocks imported from other ranks are read by the lookups but emit nothing themselves
    if (!skip && (!coarse || v < kCoarseVoxels) && owns_block(m, mki3(ent.x, ent.y, ent.z))) {
      i3 pi;
      if (!coarse) pi = mki3(ent.x * kBlockSide + (v & 7), ent.y * kBlockSide + ((v >> 3) & 7), ent.z * kBlockSide + (v >> 6));
      else pi = mki3(ent.x * kBlockSide + 2 * (v & 3), ent.y * kBlockSide + 2 * ((v >> 2) & 3), ent.z * kBlockSide + 2 * (v >> 4));
      bool empty = false;
      if (prescreen) {
        u32 acc = 3u;
        const int x = v & 7, y = (v >> 3) & 7, z = v >> 6;
#pragma unroll
        for (int dz = 0; dz < 3; dz++)
#pragma unroll
          for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) acc &= s_cls[(z + dz) * 100 + (y + dy) * 10 + (x + dx)];
        empty = acc != 0u;  // all 27 cells positive, or all 27 negative
      }
      if (!empty) ntri = mc_voxel<EMIT>(m, t, nb, voxel_to_world(m.vs, pi), tris);
    }
    if (!EMIT) per_voxel[(size_t) e * 512 + v] = (uint8_t) ntri;
    // block-wide exclusive scan of ntri in voxel-index order: wave scan + 8 wave totals through LDS
    u32 incl = (u32) ntri;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 o = __shfl_up(incl, off);
      if ((int) lane_id() >= off) incl += o;
    }
    if (lane_id() == 63) s_wave[wave] = incl;
    __syncthreads();
    u32 wave_off = 0, total = 0;
    for (int i = 0; i < 8; i++) { if (i < wave) wave_off += s_wave[i]; total += s_wave[i]; }
    if (!EMIT) {
      if (v == 0) counts[e] = total;
    } else {
      const u64 base = offsets[e] + wave_off + (incl - (u32) ntri);
      for (int j = 0; j < ntri; j++)
        if (base + j < max_tris) out[base + j] = tris[j];
        else atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TRI);
    }
    __syncthreads();
  }
}

}  // namespace mrh
