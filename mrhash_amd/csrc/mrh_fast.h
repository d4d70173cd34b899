// mrh_fast.h — the single-resolution fast path: 4 kernels per frame.
//
//   k_alloc2      rays -> LDS-deduplicated block keys -> lock-free global insert; also writes the cleaned depth
//                 ("cloud z", camera.cu:13-18) and the colour image repacked to one u32 per pixel
//   k_compact2    sweep of the block descriptors: approx-frustum test (the reference's compaction predicate) +
//                 a conservative exact-image cull that splits the compact list into VISIBLE (front of the array)
//                 and CULLED (back): a culled block provably has no voxel that projects into the image, so the
//                 integrate pass never touches its 6 KiB
//   k_fused       one wave per visible block, 8 voxels per lane as 2 x float4 per plane (1 KiB per wave
//                 instruction): depth->TSDF integration AND the garbage-collection reduction (min |sdf| over
//                 weighted voxels, max weight) in the same pass -> per-block summary
//   k_free2       GC decision from the summaries (visible: fresh; culled: unchanged since their last visit),
//                 wave-aggregated free-list pushes, cooperative zeroing
//
// The reference runs integrate (vds.cu:1095-1181) and garbageCollectIdentify (vds.cu:1674-1713) as two full
// sweeps over every in-frustum block; results here are identical (tests/test_parity_gpu.py).
#pragma once

#include "mrh_kernels.h"

namespace mrh {

constexpr int CTR_CULLED = 12;  // number of culled compact entries (stored from the back of Tab::compact)
constexpr int CTR_FREED_EARLY = 13;  // culled blocks freed inside k_compact2 this frame (counted into M)
constexpr float kFltMax = 3.402823466e+38f;
constexpr int kTileMaxPx = 576;  // LDS depth+colour tile per wave: 576 px x 8 B = 4.5 KiB (24 x 24 px: blocks beyond ~1.2 m)

struct Fast {
  float* depth_clean;  // [rows*cols] depth if in (min_depth, max_depth] else 0
  u32* rgbx;           // [rows*cols] r | g << 8 | b << 16
  uint2* summary;      // [cap_blocks] {bits of min |sdf| over weighted voxels (FLT_MAX if none), max weight}
  u32 compact_cap;     // entries in Tab::compact
  int4* bbox;          // [compact_cap] per VISIBLE compact entry: pixel footprint {col0, row0, w, h}; w == 0: none
  uint2* summary_c;    // [8 * cap_blocks] the same summary per COARSE unit (multi-resolution maps only)
#ifdef MRH_TRACE
  u64* trace;          // [compact_cap * 8] per-block phase timestamps of the last k_back launch (tuning builds only)
#endif
};

#ifdef MRH_TRACE
__device__ __forceinline__ u64 trace_now() {
  u64 t;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#define MRH_TS(slot) do { if (lane == 0) f.trace[(size_t) e * 8 + (slot)] = trace_now(); } while (0)
// k_front: one record per workgroup, stored after the k_back records (offset kTraceFront blocks)
constexpr size_t kTraceFront = 32768;
#define MRH_TSF(slot) do { if (threadIdx.x == 0) f.trace[(kTraceFront + blockIdx.x) * 8 + (slot)] = trace_now(); } while (0)
#else
#define MRH_TS(slot) do { } while (0)
#define MRH_TSF(slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------
// K1'  allocation
// ---------------------------------------------------------------------------------------------------------
constexpr int kListCap = kSetCap;

__device__ __forceinline__ void alloc_commit2(const Tab& t, const Fast& f, bool won, int slot, i3 b) {
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
    atomicAdd(&t.ctr[CTR_HEAP_FINE], 1);
    atomicExch(&t.keys[slot], kKeyTomb);
    atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_POOL);
    return;
  }
  const u32 H = t.heap_fine[idx];
  t.vals[slot] = H;
  t.desc_fine[H] = make_int4(b.x, b.y, b.z, 1);
  f.summary[H] = make_uint2(0x7F7FFFFFu, 0u);
  // the high-water mark only moves while the pool is being touched for the first time: skip the same-address
  // atomic (one per inserted block otherwise) whenever a plain read already shows a large enough value
  if ((int) H >= __hip_atomic_load(&t.ctr[CTR_HWM_FINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&t.ctr[CTR_HWM_FINE], (int) H + 1);
}

template <bool PROFILE, int TILE>
__global__ __launch_bounds__(TILE * TILE) void k_alloc2(const Cam c, const Map m, const Tab t, const Fast f,
                                                const float* __restrict__ depth, const uint8_t* __restrict__ rgb) {
  constexpr int NT = TILE * TILE;            // threads = pixels per tile
  constexpr int CAP = NT * 4;                // LDS key-set / list capacity
  __shared__ u64 set[CAP];
  __shared__ u64 list[CAP];
  __shared__ u32 s_count, s_inserted;
  const int tid = threadIdx.y * TILE + threadIdx.x;
  for (int i = tid; i < CAP; i += NT) set[i] = kKeyEmpty;
  if (tid == 0) {
    s_count = 0;
    s_inserted = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0) { t.ctr[CTR_COMPACT] = 0; t.ctr[CTR_CULLED] = 0; t.ctr[CTR_FREED_EARLY] = 0; }
  }
  __syncthreads();

  const int row = blockIdx.y * TILE + threadIdx.y;
  const int col = blockIdx.x * TILE + threadIdx.x;
  u32 my_inserted = 0;
  const bool in_img = row < c.rows && col < c.cols;
  float d = 0.f;
  if (in_img) {
    const size_t pix = (size_t) row * c.cols + col;
    d = depth[pix];
    if (d <= c.min_depth || d > c.max_depth) d = 0.f;  // camera.cu:13-18
    f.depth_clean[pix] = d;
    const uint8_t* px = rgb + pix * 3;
    f.rgbx[pix] = (u32) px[0] | ((u32) px[1] << 8) | ((u32) px[2] << 16);
  }
  const float tr = get_truncation(d, m.trunc, m.trunc_scale);
  const float dmin = fminf(c.max_int_dist, d - tr);
  const float dmax = fminf(c.max_int_dist, d + tr);
  const bool walk = in_img && !(d == 0.f) && !(dmin >= dmax);
  if (walk) {
    const f3 pw_min = se3_apply(c.R, c.t, inverse_projection(c, (u32) row, (u32) col, dmin));
    const f3 pw_max = se3_apply(c.R, c.t, inverse_projection(c, (u32) row, (u32) col, dmax));
    const f3 dd = mk3(pw_max.x - pw_min.x, pw_max.y - pw_min.y, pw_max.z - pw_min.z);
    const float inv_len = 1.0f / sqrtf(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z);
    const f3 dir = mk3(dd.x * inv_len, dd.y * inv_len, dd.z * inv_len);
    const GridRcp grid = make_grid_rcp(m.vs);
    i3 cur = world_to_block_r(grid, pw_min);
    const i3 end = world_to_block_r(grid, pw_max);
    const f3 step = mk3((float) signi(dir.x), (float) signi(dir.y), (float) signi(dir.z));
    const i3 nb = mki3(cur.x + f2i(clampf(step.x, 0.0f, 1.f)), cur.y + f2i(clampf(step.y, 0.0f, 1.f)), cur.z + f2i(clampf(step.z, 0.0f, 1.f)));
    const f3 bw = voxel_to_world(m.vs, mki3(nb.x * kBlockSide, nb.y * kBlockSide, nb.z * kBlockSide));
    const f3 boundary = mk3(bw.x - 0.5f * m.vs, bw.y - 0.5f * m.vs, bw.z - 0.5f * m.vs);
    // t_max and t_delta of an axis divide by the same direction component: one refined reciprocal per axis.  (For
    // |dir| < 1e-6 an IEEE divide would give inf / nan where this gives nan; both are overwritten just below.)
    const f3 rd = mk3(rcp_refined(dir.x), rcp_refined(dir.y), rcp_refined(dir.z));
    f3 t_max = mk3(div_rr(boundary.x - pw_min.x, dir.x, rd.x), div_rr(boundary.y - pw_min.y, dir.y, rd.y), div_rr(boundary.z - pw_min.z, dir.z, rd.z));
    f3 t_delta = mk3(div_rr(step.x * (float) kBlockSide * m.vs, dir.x, rd.x), div_rr(step.y * (float) kBlockSide * m.vs, dir.y, rd.y),
                     div_rr(step.z * (float) kBlockSide * m.vs, dir.z, rd.z));
    const i3 bound = mki3(f2i((float) end.x + step.x), f2i((float) end.y + step.y), f2i((float) end.z + step.z));
    if (fabsf(dir.x) < kFloatEps) { t_max.x = kFltMax; t_delta.x = kFltMax; }
    if (fabsf(boundary.x - dir.x) < kFloatEps) { t_max.x = kFltMax; t_delta.x = kFltMax; }
    if (fabsf(dir.y) < kFloatEps) { t_max.y = kFltMax; t_delta.y = kFltMax; }
    if (fabsf(boundary.y - dir.y) < kFloatEps) { t_max.y = kFltMax; t_delta.y = kFltMax; }
    if (fabsf(dir.z) < kFloatEps) { t_max.z = kFltMax; t_delta.z = kFltMax; }
    if (fabsf(boundary.z - dir.z) < kFloatEps) { t_max.z = kFltMax; t_delta.z = kFltMax; }
#pragma unroll 1
    for (u32 iter = 0; iter < kMaxDdaIter; iter++) {
      u64 key;
      if (!pack_key(cur, key)) {
        atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_RANGE);
      } else if (owns_block(m, cur)) {
        // tile-local set: a cheap 24-bit multiply-add mix is enough (the 64-bit murmur mix costs ~10 quarter-rate ops)
        u32 s = (u32) __mul24(cur.z, 5851) + (u32) __mul24(cur.y, 73) + (u32) cur.x;
        s = (s ^ (s >> 7)) & (CAP - 1);
        bool placed = false;
#pragma unroll 1
        for (int p = 0; p < kSetProbe; p++) {
          const u64 old = atomicCAS(&set[s], kKeyEmpty, key);
          if (old == kKeyEmpty) {  // first thread of the tile to see this block: queue it once
            list[atomicAdd(&s_count, 1u)] = key;
            placed = true;
            break;
          }
          if (old == key) { placed = true; break; }
          s = (s + 1) & (CAP - 1);
        }
        if (!placed && block_in_frustum_approx(c, m.vs, cur)) {
          // LDS set saturated (far, sparse rays): insert directly, un-aggregated
          const int slot = hash_insert(t, key);
          if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
          if (slot >= 0) {
            const int idx = atomicSub(&t.ctr[CTR_HEAP_FINE], 1);
            if (idx < 0) {
              atomicAdd(&t.ctr[CTR_HEAP_FINE], 1);
              atomicExch(&t.keys[slot], kKeyTomb);
              atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_POOL);
            } else {
              const u32 H = t.heap_fine[idx];
              t.vals[slot] = H;
              t.desc_fine[H] = make_int4(cur.x, cur.y, cur.z, 1);
              f.summary[H] = make_uint2(0x7F7FFFFFu, 0u);
              if ((int) H >= __hip_atomic_load(&t.ctr[CTR_HWM_FINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&t.ctr[CTR_HWM_FINE], (int) H + 1);
              if (PROFILE) my_inserted++;
            }
          }
        }
      }
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
  // the tile's distinct blocks, densely packed: frustum test + global insert, usually a single round
  const int n = (int) s_count;
#pragma unroll 1
  for (int base = 0; base < n; base += NT) {
    const int i = base + tid;
    const bool active = i < n;
    const u64 key = active ? list[i] : kKeyEmpty;
    const i3 b = active ? unpack_key(key) : mki3(0, 0, 0);
    bool won = false;
    int slot = -1;
    if (active && block_in_frustum_approx(c, m.vs, b)) {
      slot = hash_insert(t, key);
      if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
      won = slot >= 0;
    }
    alloc_commit2(t, f, won, slot, b);
    if (PROFILE && won) my_inserted++;
  }
  if (PROFILE) {
    if (my_inserted) atomicAdd(&s_inserted, my_inserted);
    __syncthreads();
    if (tid == 0 && s_inserted) atomicAdd(&t.prof[PROF_INSERTED], (u64) s_inserted);
  }
}

// frees one fine block from inside a wave: tombstones its key, returns its slot to the free list, clears its
// descriptor and zeroes its 6 KiB with the whole wave (garbageCollectFree + deleteHashEntryElement,
// vds.cu:1727-1844).  Called with wave-uniform arguments.
__device__ __forceinline__ void wave_free_block(const Tab& t, const int4 ent, const int lane) {
  const u32 H = (u32) ent.w;
  if (lane == 0) {
    u64 key;
    pack_key(mki3(ent.x, ent.y, ent.z), key);
    hash_erase(t, key);
    const int idx = atomicAdd(&t.ctr[CTR_HEAP_FINE], 1);
    t.heap_fine[idx + 1] = H;  // vds.cu:53-57
    t.desc_fine[H].w = 0;
  }
  uint4* p = (uint4*) (t.pool + (size_t) H * kFineBytes);
  const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < kFineBytes / 16 / kWave; k++) p[k * kWave + lane] = z;
}

// ---------------------------------------------------------------------------------------------------------
// K2'  compaction with exact-image cull
// ---------------------------------------------------------------------------------------------------------

// 0: outside the approx frustum (not compacted, vds.cu:66-77); 1: compacted, may hold voxels that project into
// the image; 2: compacted, but provably no voxel centre passes Camera::projectPoint (camera.cuh:131-147).
//
// Cull argument: voxel centres of a block are the integer lattice points of the box spanned by its 8 corner
// voxels; world->camera is affine, so every voxel's camera point is a convex combination of the 8 corner
// camera points.  z is therefore bounded by the corners' z range, and when every corner has z > 0 the ratios
// x/z, y/z are bounded by the corners' ratios.  Margins (1e-3 m in z, >= 1.5 px laterally, lateral test only
// when all corners have z >= 5 cm) cover the difference between exact and fp32 evaluation many times over.
//
// Eight lanes per block, one corner each (lane & 7 = corner, params.h:41-49 order), combined with three
// xor-shuffles: 8x the parallelism and 1/8 of the dependent chain of a thread-per-block sweep.
// FREE_CULLED: a culled block is not touched by this frame's integrate pass, so its GC decision (from the stored
// summary) can be taken — and the block freed — right here; such a block is not even entered into the culled list.
template <bool RESET_SUMMARY, bool FREE_CULLED>
__global__ __launch_bounds__(512) void k_compact2(const Cam c, const Map m, const Tab t, const Fast f, const float trunc_threshold) {
  __shared__ int s_cls[64];
  __shared__ int4 s_desc[64];
  __shared__ int4 s_bbox[64];
  const int hwm = t.ctr[CTR_HWM_FINE];
  const int corner = threadIdx.x & 7;
  const int slot = threadIdx.x >> 3;  // 0..63: block within this workgroup's batch
  for (int base = blockIdx.x * 64; base < hwm; base += gridDim.x * 64) {
    const int i = base + slot;
    int4 d = make_int4(0, 0, 0, 0);
    if (i < hwm) d = t.desc_fine[i];
    const bool live = i < hwm && (d.w & 1);
    const i3 v = mki3(d.x * kBlockSide + ((corner & 4) ? 7 : 0), d.y * kBlockSide + ((corner & 2) ? 7 : 0), d.z * kBlockSide + ((corner & 1) ? 7 : 0));
    const f3 pc = se3_apply(c.Ri, c.ti, voxel_to_world(m.vs, v));
    int r, cc;
    int any_approx = project_point<true>(c, pc, r, cc) ? 1 : 0;
    float zmin = pc.z, zmax = pc.z;
    const bool lateral = pc.z >= 0.05f;
    const float u = c.fx * pc.x / pc.z + c.cx;
    const float w = c.fy * pc.y / pc.z + c.cy;
    float umin = lateral ? u : kFltMax, umax = lateral ? u : -kFltMax;
    float vmin = lateral ? w : kFltMax, vmax = lateral ? w : -kFltMax;
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      any_approx |= __shfl_xor(any_approx, off);
      zmin = fminf(zmin, __shfl_xor(zmin, off)); zmax = fmaxf(zmax, __shfl_xor(zmax, off));
      umin = fminf(umin, __shfl_xor(umin, off)); umax = fmaxf(umax, __shfl_xor(umax, off));
      vmin = fminf(vmin, __shfl_xor(vmin, off)); vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    }
    if (corner == 0) {
      int cls = 0;
      if (live && any_approx) {
        bool cull = (zmax <= c.min_depth - 1e-3f) || (zmin > c.max_depth + 1e-3f);
        if (!cull && zmin >= 0.05f)
          cull = umax < -3.f || umin > (float) c.cols + 1.f || vmax < -3.f || vmin > (float) c.rows + 1.f;
        cls = cull ? 2 : 1;
      }
      // pixel footprint of the block for the LDS tile of k_fused: every voxel's (row, col) = int(v + .5), int(u + .5)
      // lies between the corners' extremes (same convexity argument), +-1 px for fp32 slack.  It is only a cache
      // hint: k_fused falls back to a direct gather for any pixel outside it.
      int4 bb = make_int4(0, 0, 0, 0);
      if (cls == 1 && zmin >= 0.05f) {
        int c0 = f2i_hw(floorf(umin + 0.5f)) - 1, c1 = f2i_hw(floorf(umax + 0.5f)) + 1;
        int r0 = f2i_hw(floorf(vmin + 0.5f)) - 1, r1 = f2i_hw(floorf(vmax + 0.5f)) + 1;
        c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0;
        c1 = c1 > c.cols - 1 ? c.cols - 1 : c1; r1 = r1 > c.rows - 1 ? c.rows - 1 : r1;
        const int bw = c1 - c0 + 1, bh = r1 - r0 + 1;
        if (bw > 0 && bh > 0 && bw * bh <= kTileMaxPx) bb = make_int4(c0, r0, bw, bh);
      }
      s_cls[slot] = cls;
      s_desc[slot] = make_int4(d.x, d.y, d.z, i);
      s_bbox[slot] = bb;
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // wave 0: one ballot + at most two atomics for the whole 64-block batch
      int cls = s_cls[threadIdx.x];
      const int4 e = s_desc[threadIdx.x];
      if (FREE_CULLED) {
        bool fr = false;
        if (cls == 2) {
          const uint2 sm = f.summary[e.w];
          fr = (__uint_as_float(sm.x) >= trunc_threshold) || (sm.y == 0u);
        }
        u64 todo = __ballot(fr);
        if (todo) {
          if (threadIdx.x == 0) {
            atomicAdd(&t.ctr[CTR_FREED_EARLY], __popcll(todo));  // still part of M (stats)
            if (t.prof) atomicAdd(&t.prof[PROF_FREED], (u64) __popcll(todo));
          }
          while (todo) {
            const int src = __ffsll((long long) todo) - 1;
            todo &= todo - 1;
            const int4 fe = make_int4(__shfl(e.x, src), __shfl(e.y, src), __shfl(e.z, src), __shfl(e.w, src));
            wave_free_block(t, fe, (int) threadIdx.x);
          }
          if (fr) cls = 0;
        }
      }
      const u64 bv = __ballot(cls == 1), bc = __ballot(cls == 2);
      int wbv = 0, wbc = 0;
      if (threadIdx.x == 0) {
        if (bv) wbv = atomicAdd(&t.ctr[CTR_COMPACT], __popcll(bv));
        if (bc) wbc = atomicAdd(&t.ctr[CTR_CULLED], __popcll(bc));
      }
      wbv = __shfl(wbv, 0);
      wbc = __shfl(wbc, 0);
      if (cls == 1) {
        const int idx = wbv + __popcll(bv & lanemask_lt());
        t.compact[idx] = e;
        f.bbox[idx] = s_bbox[threadIdx.x];
        if (RESET_SUMMARY) f.summary[e.w] = make_uint2(0x7F7FFFFFu, 0u);  // re-accumulated by k_fused<.., 1>
      } else if (cls == 2) {
        t.compact[(int) f.compact_cap - 1 - (wbc + __popcll(bc & lanemask_lt()))] = e;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// K3'  fused integrate + GC summary, one wave per visible block
// ---------------------------------------------------------------------------------------------------------

// ---- per-voxel work, split so that every memory round trip of a wave is issued before anything waits -----------
//
// A float4-group q of a block plane holds the 4 x-adjacent voxels q*4 .. q*4+3 (x4 = q & 1, y = (q >> 1) & 7,
// z = q >> 4).  Arithmetic is integrate_voxel's (vds.cu:1095-1181): the products R[.]*py, R[.]*pz are shared
// between the four voxels (equal inputs, equal results), every sum keeps the reference's association, both
// pixel coordinates share one refined reciprocal of pc.z (div_rr, bit-identical to IEEE division).

struct Proj4 {
  float pcz[4];
  int row[4], col[4];
  u32 mask;  // bit k: voxel k projects into the image (camera.cuh:131-147)
};

__device__ __forceinline__ Proj4 project4(const Cam& c, const Map& m, const int4 ent, const int q) {
  Proj4 P;
  const int x0 = ent.x * kBlockSide + (q & 1) * 4;
  const int y = ent.y * kBlockSide + ((q >> 1) & 7);
  const int z = ent.z * kBlockSide + (q >> 4);
  const float py = y * m.vs, pz = z * m.vs;
  const float a1 = c.Ri[1] * py, a2 = c.Ri[2] * pz;
  const float b1 = c.Ri[4] * py, b2 = c.Ri[5] * pz;
  const float c1 = c.Ri[7] * py, c2 = c.Ri[8] * pz;
  P.mask = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float px = (x0 + k) * m.vs;
    const float X = ((c.Ri[0] * px + a1) + a2) + c.ti[0];
    const float Y = ((c.Ri[3] * px + b1) + b2) + c.ti[1];
    const float Z = ((c.Ri[6] * px + c1) + c2) + c.ti[2];
    P.pcz[k] = Z;
    const bool depth_ok = !(Z <= c.min_depth || Z > c.max_depth);
    const float rz = rcp_refined(Z);
    const int row = f2i_hw((div_rr(c.fy * Y, Z, rz) + c.cy) + 0.5f);
    const int col = f2i_hw((div_rr(c.fx * X, Z, rz) + c.cx) + 0.5f);
    const bool ok = depth_ok && row >= 0 && col >= 0 && row < c.rows && col < c.cols;
    P.row[k] = row;
    P.col[k] = col;
    P.mask |= ok ? (1u << k) : 0u;
  }
  return P;
}

// which of the 4 voxels get written: depth valid and sdf > -truncation (vds.cu:1134-1145)
template <typename PT>
__device__ __forceinline__ u32 update_mask4(const Cam& c, const Map& m, const PT& P, const float (&d)[4]) {
  u32 mask = P.mask;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const bool depth_bad = (d[k] == 0.f) || (d[k] > c.max_int_dist);
    const float sdf = d[k] - P.pcz[k];
    const float trn = get_truncation(d[k], m.trunc, m.trunc_scale);
    if (depth_bad || sdf <= -trn) mask &= ~(1u << k);
  }
  return mask;
}

// running weighted mean, colour blend, weight clamp, variance term (vds.cu:1147-1180, vhu.cuh:167-181), two voxels
// per instruction where the arithmetic is plain fp32 (mrh_device.h, v2f).  The clamp `sd >= 0 ? min(t, sd) :
// max(-t, sd)` is one v_med3_f32 (t >= 0 is checked at mrh_create; a NaN sd yields -t on both sides).
template <typename PT>
__device__ __forceinline__ void blend4(const Map& m, const PT& P, const u32 mask, const float (&d)[4], const u32 (&cpx)[4],
                                       const float r_half_vs, float (&s)[4], u32 (&w)[4], float (&ss)[4]) {
  const u32 w1 = (u32) (m.weight_sample & 0xFF);
  const u32 wmax = (u32) (m.weight_max & 0xFF);
  const v2f half_vs = splat2(m.vs / 2), rh = splat2(r_half_vs), w1f = splat2((float) w1);
#pragma unroll
  for (int k = 0; k < 4; k += 2) {
    const v2f dd = mk2(d[k], d[k + 1]);
    const v2f trn = splat2(m.trunc) + splat2(m.trunc_scale) * dd;  // get_truncation
    v2f sd = dd - mk2(P.pcz[k], P.pcz[k + 1]);
    sd = mk2(__builtin_amdgcn_fmed3f(sd.x, -trn.x, trn.x), __builtin_amdgcn_fmed3f(sd.y, -trn.y, trn.y));
    const v2f s0 = mk2(s[k], s[k + 1]);
    const u32 old0 = w[k], old1 = w[k + 1];
    const u32 w00 = old0 >> 24, w01 = old1 >> 24;
    const v2f curr_mean = mk2(w00 > 0 ? s0.x : sd.x, w01 > 0 ? s0.y : sd.y);
    const v2f delta = div_rr2(sd - curr_mean, half_vs, rh);
    const v2f wsum = mk2((float) (int) (w00 + w1), (float) (int) (w01 + w1));
    const v2f sn = div_rr2(s0 * mk2((float) w00, (float) w01) + sd * w1f, wsum, rcp_refined2(wsum));
    const v2f delta2 = div_rr2(sd - sn, half_vs, rh);
    const v2f sq = splat2(0.f) + delta * delta2;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      // combineVoxel's colour (vhu.cuh:170-176): u8(0.5*c0 + 0.5*c1 + 0.5) per channel, with c0 := c1 for a fresh voxel.
      // 0.5*c0 + 0.5*c1 + 0.5 is exact in fp32 for 8-bit inputs and truncates to (c0 + c1 + 1) >> 1, i.e. the
      // rounded-up byte average; computed for the three channels at once: (a | b) - (((a ^ b) >> 1) & 0x7f7f7f).
      const u32 old = j ? old1 : old0, w0 = j ? w01 : w00;
      const u32 c1x = cpx[k + j] & 0x00FFFFFFu;
      const u32 c0x = (w0 == 0) ? c1x : (old & 0x00FFFFFFu);
      const u32 rgbn = (c0x | c1x) - (((c0x ^ c1x) >> 1) & 0x007F7F7Fu);
      const u32 wn = (w0 + w1) < wmax ? (w0 + w1) : wmax;
      if ((mask >> (k + j)) & 1u) {
        s[k + j] = j ? sn.y : sn.x;
        w[k + j] = rgbn | (wn << 24);
        ss[k + j] = j ? sq.y : sq.x;
      }
    }
  }
}

// ---- LDS pixel tile ----------------------------------------------------------------------------------------
// dense, row-contiguous fill of the block's pixel footprint {depth bits, colour}
__device__ __forceinline__ float pixel_reach(const Cam& c, const Map& m, const float d) {
  // d == 0 (invalid pixel, camera.cu:13-18) or d > max_int_dist: rejected for every voxel (vds.cu:1134-1137) -> 0;
  // d > 0: d + truncation(d); anything else (negative, NaN): not provably skippable -> FLT_MAX
  const bool rejected = (d == 0.f) || (d > c.max_int_dist);
  return rejected ? 0.f : (d > 0.f ? d + get_truncation(d, m.trunc, m.trunc_scale) : kFltMax);
}
// Returns this lane's largest d + truncation(d) over the valid pixels it staged (0 if none): k_back's early-out.
// Depths are > 0, so the float maximum is also the maximum of the raw bits.
__device__ __forceinline__ float tile_fill(const Cam& c, const Map& m, const Fast& f, const int4 bb, const int lane, uint2* tile) {
  const int npx = bb.z * bb.w;
  float reach = 0.f;
  if (npx <= 0) return reach;
  const float inv_w = 1.0f / (float) bb.z;
#pragma unroll 1
  for (int p0 = lane; p0 < npx; p0 += 256) {  // 4 pixels per lane and round: all 8 gathers in flight before the first wait
    float dv[4];
    u32 cv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = min(p0 + 64 * j, npx - 1);  // lanes past the end re-read the last pixel (same cache line, no branch)
      const int r = (int) (((float) p + 0.5f) * inv_w);  // p / w for p < 2^10 (exact: slack 0.5 / w >> fp32 error)
      const int cc = p - r * bb.z;
      const u32 g = (u32) (__mul24(bb.y + r, c.cols) + bb.x + cc);
      dv[j] = f.depth_clean[g];
      cv[j] = f.rgbx[g];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = p0 + 64 * j;
      const float d = dv[j];
      if (p < npx) tile[p] = make_uint2(__float_as_uint(d), cv[j]);
      reach = __uint_as_float(umax_(__float_as_uint(reach), __float_as_uint(pixel_reach(c, m, d))));
    }
  }
  return reach;
}

// The same fill split in two so that a wave can do memory-independent work (the projections) between issuing the
// gathers of the first 256 footprint pixels and parking them in LDS; footprints > 256 px finish through a second round.
struct TileRegs {
  float dv[4];
  u32 cv[4];
};
__device__ __forceinline__ void tile_issue(const Cam& c, const Fast& f, const int4 bb, const int lane, TileRegs& tr) {
  const int npx = bb.z * bb.w;
  if (npx <= 0) return;
  const float inv_w = 1.0f / (float) bb.z;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int p = min(lane + 64 * j, npx - 1);
    const int r = (int) (((float) p + 0.5f) * inv_w);
    const int cc = p - r * bb.z;
    const u32 g = (u32) (__mul24(bb.y + r, c.cols) + bb.x + cc);
    tr.dv[j] = f.depth_clean[g];
    tr.cv[j] = f.rgbx[g];
  }
}
__device__ __forceinline__ float tile_commit(const Cam& c, const Map& m, const Fast& f, const int4 bb, const int lane, uint2* tile,
                                             const TileRegs& tr) {
  const int npx = bb.z * bb.w;
  float reach = 0.f;
  if (npx <= 0) return reach;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int p = lane + 64 * j;
    if (p < npx) tile[p] = make_uint2(__float_as_uint(tr.dv[j]), tr.cv[j]);
    reach = __uint_as_float(umax_(__float_as_uint(reach), __float_as_uint(pixel_reach(c, m, tr.dv[j]))));
  }
  if (npx > 256) {
    const float inv_w = 1.0f / (float) bb.z;
#pragma unroll 1
    for (int p0 = 256 + lane; p0 < npx; p0 += 128) {
      float dv[2];
      u32 cv[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int p = min(p0 + 64 * j, npx - 1);
        const int r = (int) (((float) p + 0.5f) * inv_w);
        const int cc = p - r * bb.z;
        const u32 g = (u32) (__mul24(bb.y + r, c.cols) + bb.x + cc);
        dv[j] = f.depth_clean[g];
        cv[j] = f.rgbx[g];
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int p = p0 + 64 * j;
        if (p < npx) tile[p] = make_uint2(__float_as_uint(dv[j]), cv[j]);
        reach = __uint_as_float(umax_(__float_as_uint(reach), __float_as_uint(pixel_reach(c, m, dv[j]))));
      }
    }
  }
  return reach;
}

// branch-free lookups: all NB x 4 ds_read_b64 are issued back to back; pixels outside the footprint (or blocks
// without a tile) take ONE wave-uniform fallback branch with direct gathers.
template <int NB>
__device__ __forceinline__ void tile_lookup(const Fast& f, const int cols, const int4 bb, const uint2* tile, const Proj4 (&P)[NB],
                                            float (&d)[NB][4], u32 (&cpx)[NB][4]) {
  u32 miss = 0;
  u32 li[NB][4];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 lr = (u32) (P[b].row[k] - bb.y), lc = (u32) (P[b].col[k] - bb.x);
      const bool in = lr < (u32) bb.w && lc < (u32) bb.z;
      li[b][k] = in ? lr * (u32) bb.z + lc : 0u;
      if (!in && ((P[b].mask >> k) & 1u)) miss |= 1u << (b * 4 + k);
    }
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint2 px = tile[li[b][k]];
      d[b][k] = __uint_as_float(px.x);
      cpx[b][k] = px.y;
    }
  if (__ballot(miss != 0)) {
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int k = 0; k < 4; k++)
        if ((miss >> (b * 4 + k)) & 1u) {
          const u32 pix = (u32) (__mul24(P[b].row[k], cols) + P[b].col[k]);  // miss implies the voxel is in the image
          d[b][k] = f.depth_clean[pix];
          cpx[b][k] = f.rgbx[pix];
        }
  }
}

// One wave owns NB x 256 voxels: NB = 2 -> a whole block per wave (summary written with a plain store),
// NB = 1 -> half a block per wave (twice the parallelism; the two halves meet in the summary through
// atomicMin / atomicMax on the raw bits — |sdf| >= 0, so float order == unsigned order; k_compact2 resets it).
//
// Dependency chain of a wave: block entry -> { voxel planes (HBM)  ||  projections -> depth+colour gathers (L2) }
// -> blend -> stores.  All loads are in flight before the first wait; the colour gather is issued together with
// the depth gather (same pixel index) instead of after the depth test.
// FREE (NB == 2 only): the wave that just computed a block's summary also takes the garbage-collection decision
// (vds.cu:1708-1711) and frees the block on the spot, so the fast path needs no separate free kernel.
template <bool INTEGRATE, int NB, bool FREE>
__global__ __launch_bounds__(256) void k_fused(const Cam c, const Map m, const Tab t, const Fast f, const float trunc_threshold) {
  extern __shared__ __attribute__((aligned(16))) uint2 s_tile[];  // (blockDim.x / 64) x kTileMaxPx
  const int nvis = t.ctr[CTR_COMPACT];
  const int nitems = nvis * (2 / NB);
  const int lane = threadIdx.x & 63;
  const int wpw = blockDim.x >> 6;
  const int gw = blockIdx.x * wpw + (threadIdx.x >> 6);
  const int nw = gridDim.x * wpw;
  const float r_half_vs = rcp_refined(m.vs / 2);
  for (int e = gw; e < nitems; e += nw) {
    const int4 ent = t.compact[NB == 2 ? e : (e >> 1)];
    const int half = NB == 2 ? 0 : (e & 1);
    const u32 H = (u32) ent.w;
    float4* ps = (float4*) (t.pool + (size_t) H * kFineBytes);
    float4* pq = ps + 128;
    uint4* pw = (uint4*) (ps + 256);
    float4 S[NB];
    uint4 W[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const int q = lane + 64 * (b + half);
      S[b] = ps[q];
      W[b] = pw[q];
    }
    Proj4 P[NB];
    float d[NB][4];
    u32 cpx[NB][4];
    if (INTEGRATE) {
      // stage the block's pixel footprint {depth, colour} in this wave's LDS tile with dense, row-contiguous
      // loads; the per-voxel lookups then hit LDS instead of issuing 64-address global gathers (one lane per
      // cycle in the texture addresser — measured as half of this kernel's time)
      const int4 bb = f.bbox[NB == 2 ? e : (e >> 1)];
      uint2* tile = &s_tile[(threadIdx.x >> 6) * kTileMaxPx];
      tile_fill(c, m, f, bb, lane, tile);
#pragma unroll
      for (int b = 0; b < NB; b++) P[b] = project4(c, m, ent, lane + 64 * (b + half));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      tile_lookup<NB>(f, c.cols, bb, tile, P, d, cpx);
      __builtin_amdgcn_wave_barrier();  // the tile is rewritten by this wave's next item
    }
    float mn = kFltMax;
    u32 mx = 0;
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const int q = lane + 64 * (b + half);
      float s[4] = {S[b].x, S[b].y, S[b].z, S[b].w};
      u32 w[4] = {W[b].x, W[b].y, W[b].z, W[b].w};
      if (INTEGRATE) {
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
        const u32 mask = update_mask4(c, m, P[b], d[b]);
        blend4(m, P[b], mask, d[b], cpx[b], r_half_vs, s, w, ss);
        if (mask) {
          ps[q] = make_float4(s[0], s[1], s[2], s[3]);
          pw[q] = make_uint4(w[0], w[1], w[2], w[3]);
          if (mask == 0xF) {
            pq[q] = make_float4(ss[0], ss[1], ss[2], ss[3]);
          } else {
            float* pqs = (float*) (pq + q);
#pragma unroll
            for (int k = 0; k < 4; k++)
              if (mask & (1u << k)) pqs[k] = ss[k];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 wk = w[k] >> 24;
        if (wk != 0) mn = fminf(mn, fabsf(s[k]));
        mx = wk > mx ? wk : mx;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, off));
      const u32 o = __shfl_xor(mx, off);
      mx = o > mx ? o : mx;
    }
    if (lane == 0) {
      if (NB == 2) {
        f.summary[H] = make_uint2(__float_as_uint(mn), mx);
      } else {
        atomicMin(&f.summary[H].x, __float_as_uint(mn));
        atomicMax(&f.summary[H].y, mx);
      }
    }
    if (FREE && NB == 2 && (mn >= trunc_threshold || mx == 0)) {
      wave_free_block(t, ent, lane);
      if (lane == 0 && t.prof) atomicAdd(&t.prof[PROF_FREED], 1ull);
    }
  }
}

// Software-pipelined variant: a wave that owns several items requests the voxel planes of item i+1 right after
// item i's pixel tile has been consumed, so the HBM latency of the next block hides under the blend arithmetic
// and the stores of the current one (the plain variant has every wave of the chip load, compute and store in
// lock-step, which leaves the memory system idle while the VALUs work and vice versa).
template <int NB>
__global__ __launch_bounds__(256) void k_fused_pipe(const Cam c, const Map m, const Tab t, const Fast f) {
  __shared__ uint2 s_tile[4 * kTileMaxPx];
  const int nvis = t.ctr[CTR_COMPACT];
  const int nitems = nvis * (2 / NB);
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  const float r_half_vs = rcp_refined(m.vs / 2);
  uint2* tile = &s_tile[(threadIdx.x >> 6) * kTileMaxPx];
  int e = gw;
  if (e >= nitems) return;
  int4 ent = t.compact[NB == 2 ? e : (e >> 1)];
  int4 bb = f.bbox[NB == 2 ? e : (e >> 1)];
  float4 S[NB];
  uint4 W[NB];
  {
    const float4* ps = (const float4*) (t.pool + (size_t) (u32) ent.w * kFineBytes);
    const uint4* pw = (const uint4*) (ps + 256);
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const int q = lane + 64 * (b + (NB == 2 ? 0 : (e & 1)));
      S[b] = ps[q];
      W[b] = pw[q];
    }
  }
  while (true) {
    const int half = NB == 2 ? 0 : (e & 1);
    const int en = e + nw;
    const bool more = en < nitems;
    // (1) pixel tile of the current item
    tile_fill(c, m, f, bb, lane, tile);
    // (2) descriptor of the next item (tiny loads, consumed after the blend)
    int4 ent_n = ent, bb_n = bb;
    if (more) {
      ent_n = t.compact[NB == 2 ? en : (en >> 1)];
      bb_n = f.bbox[NB == 2 ? en : (en >> 1)];
    }
    // (3) projections, then tile lookups
    Proj4 P[NB];
    float d[NB][4];
    u32 cpx[NB][4];
#pragma unroll
    for (int b = 0; b < NB; b++) P[b] = project4(c, m, ent, lane + 64 * (b + half));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    tile_lookup<NB>(f, c.cols, bb, tile, P, d, cpx);
    __builtin_amdgcn_wave_barrier();
    // (4) voxel planes of the next item: in flight during the blend + stores below
    float4 Sn[NB];
    uint4 Wn[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) { Sn[b] = S[b]; Wn[b] = W[b]; }
    if (more) {
      const float4* psn = (const float4*) (t.pool + (size_t) (u32) ent_n.w * kFineBytes);
      const uint4* pwn = (const uint4*) (psn + 256);
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const int q = lane + 64 * (b + (NB == 2 ? 0 : (en & 1)));
        Sn[b] = psn[q];
        Wn[b] = pwn[q];
      }
    }
    // (5) blend + stores + summary of the current item
    const u32 H = (u32) ent.w;
    float4* ps = (float4*) (t.pool + (size_t) H * kFineBytes);
    float4* pq = ps + 128;
    uint4* pw = (uint4*) (ps + 256);
    float mn = kFltMax;
    u32 mx = 0;
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const int q = lane + 64 * (b + half);
      float s[4] = {S[b].x, S[b].y, S[b].z, S[b].w};
      u32 w[4] = {W[b].x, W[b].y, W[b].z, W[b].w};
      float ss[4] = {0.f, 0.f, 0.f, 0.f};
      const u32 mask = update_mask4(c, m, P[b], d[b]);
      blend4(m, P[b], mask, d[b], cpx[b], r_half_vs, s, w, ss);
      if (mask) {
        ps[q] = make_float4(s[0], s[1], s[2], s[3]);
        pw[q] = make_uint4(w[0], w[1], w[2], w[3]);
        if (mask == 0xF) {
          pq[q] = make_float4(ss[0], ss[1], ss[2], ss[3]);
        } else {
          float* pqs = (float*) (pq + q);
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (mask & (1u << k)) pqs[k] = ss[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 wk = w[k] >> 24;
        if (wk != 0) mn = fminf(mn, fabsf(s[k]));
        mx = wk > mx ? wk : mx;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, off));
      const u32 o = __shfl_xor(mx, off);
      mx = o > mx ? o : mx;
    }
    if (lane == 0) {
      if (NB == 2) {
        f.summary[H] = make_uint2(__float_as_uint(mn), mx);
      } else {
        atomicMin(&f.summary[H].x, __float_as_uint(mn));
        atomicMax(&f.summary[H].y, mx);
      }
    }
    if (!more) break;
    e = en; ent = ent_n; bb = bb_n;
#pragma unroll
    for (int b = 0; b < NB; b++) { S[b] = Sn[b]; W[b] = Wn[b]; }
  }
}

// profile mode only: U = voxels the next k_fused launch will write (the predicate depends on pose, depth image and
// block list only, not on voxel contents), M = compact blocks.  Runs outside the timed bracket.
__global__ __launch_bounds__(256) void k_count_updates(const Cam c, const Map m, const Tab t, const Fast f, u64* __restrict__ partials,
                                                       const int merged_set) {
  // merged_set >= 0: list counters of the two-launch path live at ctr[merged_set .. merged_set + 2]
  const int nvis = merged_set >= 0 ? t.ctr[merged_set] : t.ctr[CTR_COMPACT];
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  u32 cnt = 0;
  for (int e = gw; e < nvis; e += nw) {
    const int4 ent = t.compact[e];
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const Proj4 P = project4(c, m, ent, lane + 64 * b);
      float d[4];
#pragma unroll
      for (int k = 0; k < 4; k++) d[k] = f.depth_clean[((P.mask >> k) & 1u) ? (u32) (__mul24(P.row[k], c.cols) + P.col[k]) : 0u];
      cnt += __popc(update_mask4(c, m, P, d));
    }
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if (lane == 0) {
    partials[gw] += (u64) cnt;
    if (gw == 0)
      t.prof[PROF_COMPACT] += merged_set >= 0 ? (u64) (nvis + t.ctr[merged_set + 1] + t.ctr[merged_set + 2])
                                              : (u64) (nvis + t.ctr[CTR_CULLED] + t.ctr[CTR_FREED_EARLY]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// K4'  free by summary  (garbageCollectIdentify's decision, vds.cu:1708-1711, + garbageCollectFree :1827-1844)
// ---------------------------------------------------------------------------------------------------------
template <bool PROFILE>
__global__ __launch_bounds__(256) void k_free2(const Tab t, const Fast f, const float trunc_threshold) {
  const int nvis = t.ctr[CTR_COMPACT];
  const int total = nvis + t.ctr[CTR_CULLED];
  for (int base = blockIdx.x * 256; base < total; base += gridDim.x * 256) {
    const int e = base + threadIdx.x;
    bool fr = false;
    int4 ent = make_int4(0, 0, 0, 0);
    if (e < total) {
      ent = t.compact[e < nvis ? e : (int) f.compact_cap - 1 - (e - nvis)];
      const uint2 sm = f.summary[ent.w];
      fr = (__uint_as_float(sm.x) >= trunc_threshold) || (sm.y == 0u);
    }
    const u32 H = (u32) ent.w;
    if (fr) {
      u64 key;
      pack_key(mki3(ent.x, ent.y, ent.z), key);
      fr = hash_erase(t, key);
    }
    const u64 ballot = __ballot(fr);
    if (!ballot) continue;
    const int leader = __ffsll((long long) ballot) - 1;
    int b0 = 0;
    if ((int) lane_id() == leader) b0 = atomicAdd(&t.ctr[CTR_HEAP_FINE], __popcll(ballot));
    b0 = __shfl(b0, leader);
    if (fr) {
      t.heap_fine[b0 + 1 + __popcll(ballot & lanemask_lt())] = H;  // vds.cu:53-57
      t.desc_fine[H].w = 0;
    }
    if (PROFILE && lane_id() == 0) atomicAdd(&t.prof[PROF_FREED], (u64) __popcll(ballot));
    u64 todo = ballot;
    while (todo) {
      const int src = __ffsll((long long) todo) - 1;
      todo &= todo - 1;
      const u32 bH = __shfl(H, src);
      uint4* p = (uint4*) (t.pool + (size_t) bH * kFineBytes);
      const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < kFineBytes / 16 / kWave; k++) p[k * kWave + lane_id()] = z;
    }
  }
}

}  // namespace mrh
