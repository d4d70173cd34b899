// mrh_fast2.h — the fast path: TWO launches per frame (single-resolution and, with MULTI, multi-resolution maps).
//
//   k_front   workgroups [0, S):            sweep of the block descriptors that existed before this frame (they come first
//                                           in the grid so that they start first): approx-frustum predicate + exact-image
//                                           cull -> VISIBLE list (+ pixel footprint, nearest corner z) or, for culled
//                                           blocks whose stored summary says "collect", the CULLED-FREE list
//             workgroups [S, S + tiles):    block allocation for one 16x16 pixel tile (rays -> LDS key set -> one probe
//                                           per distinct key -> lock-free insert); a new block is listed by its inserter
//             The two roles never touch the same block (descriptors carry the frame stamp of their insertion), the
//             allocation role only POPS the free list and nothing is freed in this launch, so they run concurrently:
//             the latency-bound sweep hides under the ray marching.
//   k_back    one wave per visible block:   depth->TSDF integration + GC summary + GC decision + free;
//             then the same waves free the CULLED-FREE list.  This launch only PUSHES the fine free list.
//
// List counters rotate over six sets: k_front(f) appends to set f % 6, k_back(f) reads it and zeroes set (f - 1) % 6, so no
// reset launch or memset is needed (six, not two: a pipelining context keeps up to four frames in flight).
// Reference mapping: allocBlocksKernel vds.cu:758-857, flatAndReduceHashTableKernel :406-434, integrateDepthMapKernel
// :1095-1181, garbageCollectIdentify/Free :1674-1713 / :1827-1844 — four+ launches with host round trips there.
#pragma once

#include "mrh_pipe.h"

namespace mrh {

// counter slots (ints) inside Tab::ctr for the SIX list sets: frame g appends to set g % 6; the launch that integrates frame g
// reads it and zeroes set (g - 1) % 6 — the set of the frame before it, whose integration is complete (stream order) and whose
// next user, frame g + 5, is not enqueued before this launch is known to have started (integrate_lazy's throttle)
constexpr int CTR_SET0 = 32;       // set p lives at CTR_SET0 + 4 * p: {n_visible, n_culled_kept, n_culled_free, unused}
constexpr int kListSets = 6;       // = kPipeRing: frames a pipelining context may have between its newest front half and the oldest unfinished integration, + 2
constexpr u32 kZombieBit = 0x80000000u;  // in Fast::summary[H].y: the block was emptied by a pipelined frame's garbage collection
                                         // and still sits in the table (pipelined frames)

struct Lists {
  int4* vis;        // = Tab::compact (front)
  int4* bbox;       // pixel footprint per visible entry
  int4* cfree;      // culled blocks to free this frame {x, y, z, H}
  float* zmin;      // per visible entry: smallest camera-frame z of the block's corners (= of all its voxels)
  u32 cap;
};

__device__ __forceinline__ int desc_w(u32 stamp) { return (int) (1u | (stamp << 1)); }

// LDS of the allocation / sweep roles
struct FrontShared {
  u64 set[kRayCap];
  u64 list[kRayCap];
  u32 count, inserted;
  int nvis, nfree, nkeep, bv, bf;
};

// stamped pop + descriptor of a block this lane just won in the table; returns the block index or -1 (pool exhausted)
__device__ __forceinline__ int commit_block(const Tab& t, const Fast& f, const int slot, const int heap_idx, const i3 b, const u32 stamp,
                                            const int hwm_at_start) {
  if (heap_idx < 0) {
    publish_without_storage(t, slot, CTR_HEAP_FINE);
    return -1;
  }
  const u32 H = t.heap_fine[heap_idx];
  t.vals[slot] = H;
  t.desc_fine[H] = make_int4(b.x, b.y, b.z, desc_w(stamp));
  f.summary[H] = make_uint2(0x7F7FFFFFu, 0u);
  // the high-water mark only grows: below its value at launch nothing is to do, above it the max is fire-and-forget
  if ((int) H >= hwm_at_start) atomicMax(&t.ctr[CTR_HWM_FINE], (int) H + 1);
  return (int) H;
}

// Rays of one 16x16 pixel tile (allocBlocksKernel vds.cu:758-857 up to the insert): distinct keys of the traversed blocks in
// sh.list[0, sh.count).  Also writes the cleaned depth / packed colour of its pixels (calculateCloudKernel's validity rule,
// camera.cu:13-18).  Returns true for a lane whose pixel needs the literal walk (LDS set saturated by far, sparse rays; keys
// out of range): the caller re-walks it with walk_ray.  Ends with the workgroup barrier that completes the list.
// SPH: the spherical camera model (camera.cuh:91-99): the "depth" every later step reads is getDepth(cloud) — the norm of the
// back-projected point (k_cloud_depth in the general path) —, rays leave through inverse_projection_m (mrh_softmath.h, D8).
template <bool SPH = false>
__device__ __forceinline__ bool tile_rays(const Cam& c, const Map& m, uint2* __restrict__ dcx, const float* __restrict__ depth,
                                          const uint8_t* __restrict__ rgb, const int tiles_x, const int tile_id, FrontShared& sh, float& d_out) {
  constexpr int NT = 256;
  const int tid = threadIdx.x;
  for (int i = tid; i < kRayCap; i += NT) sh.set[i] = kKeyEmpty;
  if (tid == 0) { sh.count = 0; sh.inserted = 0; }
  __syncthreads();
  const int ty = tile_id / tiles_x, tx = tile_id - ty * tiles_x;
  const int row = ty * kRayTile + (tid >> 4), col = tx * kRayTile + (tid & 15);
  bool slow = false;
  d_out = 0.f;
  if (row < c.rows && col < c.cols) {
    const size_t pix = (size_t) row * c.cols + col;
    float d = depth[pix];
    if (d <= c.min_depth || d > c.max_depth) d = 0.f;  // camera.cu:13-18
    if (SPH && d != 0.f) d = get_depth(c, inverse_projection_m(c, (u32) row, (u32) col, d));  // camera.cuh:120-129
    d_out = d;
    const uint8_t* px = rgb + pix * 3;
    dcx[pix] = make_uint2(__float_as_uint(d), (u32) px[0] | ((u32) px[1] << 8) | ((u32) px[2] << 16));
    // hot path: keys into the LDS set; anything rare is left to the literal walk, outside the loop every lane runs
    const RayState ray = ray_setup<SPH>(c, m, row, col, d);
    if (ray.valid) {
      if (!ray_keys_in_range(ray)) {
        slow = true;
      } else {
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
    }
  }
  __syncthreads();
  return slow;
}

// the literal walk of one pixel with a direct insert of every block of its ray (the rare path of a tile)
template <bool PROFILE, bool MARK, bool SPH = false>
__device__ __forceinline__ void tile_slow_pixel(const Cam& c, const Map& m, const Tab& t, const Fast& f, const Lists& L, const int row, const int col,
                                                const float d, const u32 stamp, const int cs, const int hwm0) {
  walk_ray<SPH>(c, m, t, row, col, d, [&](const i3 cur, const u64 key) {
    if (MARK) {  // a block that is in the table is wanted by this frame (pipelined frames)
      const int s0 = hash_find(t, key);
      if (s0 >= 0) { f.want[s0] = stamp; return; }
    }
    if (!(SPH ? block_in_frustum_approx_m(c, m.vs, cur) : block_in_frustum_approx(c, m.vs, cur))) return;
    const int slot = hash_insert(t, key);
    if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
    if (slot < 0) return;
    const int H = commit_block(t, f, slot, atomicSub(&t.ctr[CTR_HEAP_FINE], 1), cur, stamp, hwm0);
    if (H < 0) return;
    // listed without a pixel footprint (k_back derives it)
    const int li = atomicAdd(&t.ctr[cs + 0], 1);
    L.vis[li] = make_int4(cur.x, cur.y, cur.z, H);
    L.bbox[li] = make_int4(0, 0, 0, 0);
    if (PROFILE) atomicAdd(&t.prof[PROF_INSERTED], 1ull);
  });
}

// Probe + insert of up to 64 distinct keys of a tile by one wave (`key` = kKeyEmpty on idle lanes): one probe per key
// (hash_find_claim also remembers where an insert would land) -> frustum test -> one CAS on the remembered slot -> free-list
// pop wave-aggregated, list append issued together with the read of the popped entries.  Returns the blocks this lane inserted.
template <bool MARK, bool SPH = false>
__device__ __forceinline__ u32 wave_insert_keys(const Cam& c, const Map& m, const Tab& t, const Fast& f, const Lists& L, const u64 key,
                                                const u32 stamp, const int cs, const int hwm0) {
  const bool active = key != kKeyEmpty;
  const i3 b = active ? unpack_key(key) : mki3(0, 0, 0);
  bool won = false;
  int slot = -1;
  // probe first: ~95 % of a tile's blocks already exist, and for those neither the 8-corner frustum test nor
  // the insert protocol is needed
  int claim;
  u64 claim_val;
  if (active) {
    const int found = hash_find_claim(t, key, claim, claim_val);
    if (found >= 0) {
      // pipelined frames: an earlier frame's garbage collection runs next to this probe and may be emptying this very block; it then
      // stays in the table, and whether it lives on is decided by this mark: "frame `stamp` has a ray through it"
      if (MARK) f.want[found] = stamp;
    } else if (SPH ? block_in_frustum_approx_m(c, m.vs, b) : block_in_frustum_approx(c, m.vs, b)) {
      slot = hash_insert_at(t, key, claim, claim_val);
      if (slot == -2) atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_TABLE);
      won = slot >= 0;
    }
  }
  // wave-aggregated pop of the fine free list and append to the list
  const u64 ballot = __ballot(won);
  u32 inserted = 0;
  if (ballot) {
    const int leader = __ffsll((long long) ballot) - 1;
    int hb = 0;
    if ((int) lane_id() == leader) hb = atomicSub(&t.ctr[CTR_HEAP_FINE], __popcll(ballot));
    hb = __shfl(hb, leader);
    // the pop decides who got a block (index >= 0); the list append is then issued together with the read of the
    // popped stack entries, so the two round trips overlap
    const int hidx = hb - __popcll(ballot & lanemask_lt());
    const u64 ok = __ballot(won && hidx >= 0);
    int lb = 0;
    if (ok && (int) lane_id() == __ffsll((long long) ok) - 1) lb = atomicAdd(&t.ctr[cs + 0], __popcll(ok));
    int H = -1;
    if (won) H = commit_block(t, f, slot, hidx, b, stamp, hwm0);
    if (ok) {
      const int lead2 = __ffsll((long long) ok) - 1;
      lb = __shfl(lb, lead2);
      if (H >= 0) {
        const int idx = lb + __popcll(ok & lanemask_lt());
        L.vis[idx] = make_int4(b.x, b.y, b.z, H);
        L.bbox[idx] = make_int4(0, 0, 0, 0);  // k_back derives footprint and zmin itself
        inserted = 1;
      }
    }
  }
  return inserted;
}

// Allocation for one 16x16 pixel tile in ONE launch (k_front): rays, then probe / insert of the tile's keys.
template <bool PROFILE, bool MARK, bool SPH = false>
__device__ __forceinline__ void front_tile(const Cam& c, const Map& m, const Tab& t, const Fast& f, const Lists& L,
                                           const float* __restrict__ depth, const uint8_t* __restrict__ rgb, const int tiles_x,
                                           const int tile_id, const u32 stamp, const int cs, FrontShared& sh) {
  constexpr int NT = 256;
  const int tid = threadIdx.x;
  const int hwm0 = t.ctr[CTR_HWM_FINE];  // scalar load, before any store of this launch
  MRH_TSF(1);
  float d;
  const bool slow = tile_rays<SPH>(c, m, f.dcx, depth, rgb, tiles_x, tile_id, sh, d);
  MRH_TSF(3);
  if (slow) {
    const int ty = tile_id / tiles_x, tx = tile_id - ty * tiles_x;
    tile_slow_pixel<PROFILE, MARK, SPH>(c, m, t, f, L, ty * kRayTile + (tid >> 4), tx * kRayTile + (tid & 15), d, stamp, cs, hwm0);
  }
  const int n = (int) sh.count;
  u32 my_inserted = 0;
#pragma unroll 1
  for (int base = 0; base < n; base += NT) {
    const int i = base + tid;
    my_inserted += wave_insert_keys<MARK, SPH>(c, m, t, f, L, i < n ? sh.list[i] : kKeyEmpty, stamp, cs, hwm0);
  }
  MRH_TSF(4);
#ifdef MRH_TRACE
  if (tid == 0) {
    u32 hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    u32 xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    f.trace[(kTraceFront + blockIdx.x) * 8 + 7] = (u64) n;
    f.trace[(kTraceFront + blockIdx.x) * 8 + 5] = (u64) hw | ((u64) xcc << 32);
  }
#endif
  if (PROFILE) {
    if (my_inserted) atomicAdd(&sh.inserted, my_inserted);
    __syncthreads();
    if (tid == 0 && sh.inserted) atomicAdd(&t.prof[PROF_INSERTED], (u64) sh.inserted);
  }
}

// Sweep of the block descriptors that existed before this frame, chunk `sw` of `n_sweep`
// (flatAndReduceHashTableKernel vds.cu:406-434 + the exact-image cull).
// Same-address atomics retire at roughly one per 7-10 ns on this chip, so a sweep that appends per 32-block batch
// (~3 atomics x 1250 batches per frame) is bound by exactly that.  Each sweep workgroup therefore owns one
// contiguous chunk of descriptors, stages its results in LDS (aliasing the key set / list of the allocation role)
// and publishes them with three atomics per workgroup.
// DEFER (pipelined frames): the previous frame's integration runs next to this sweep, so the stored summaries are not final: every culled
// block goes on the culled list as a CANDIDATE and the launch that integrates this frame decides from the then-final summary.
// SPH: the approx-frustum predicate of the spherical model (camera.cuh:184-201); no exact-image cull and no pixel footprint (both
// rest on the pinhole projection being a ratio of affine functions): every block of the approx frustum is VISIBLE, its lookups gather.
template <bool MULTI, bool DEFER = false, bool SPH = false>
__device__ __forceinline__ void front_sweep(const Cam& c, const Map& m, const Tab& t, const Fast& f, const Lists& L, const u32 stamp,
                                            const int cs, const int gc_on, const float trunc_threshold, const int sw, const int n_sweep,
                                            FrontShared& sh) {
  constexpr int NT = 256;
  const int tid = threadIdx.x;
  const int hwm = t.ctr[CTR_HWM_FINE];
  int4* st_vis = (int4*) sh.list;   // 256 x {entry, bbox}
  int4* st_free = (int4*) sh.set;   // 512 entries
  constexpr int kStVis = kRayCap * 8 / 32, kStFree = kRayCap * 8 / 16;
  if (tid == 0) { sh.nvis = 0; sh.nfree = 0; sh.nkeep = 0; }
  __syncthreads();
  auto flush = [&]() {  // all threads of the workgroup
    __syncthreads();
    if (tid == 0) {
      sh.bv = sh.nvis ? atomicAdd(&t.ctr[cs + 0], sh.nvis) : 0;
      sh.bf = sh.nfree ? atomicAdd(&t.ctr[cs + 2], sh.nfree) : 0;
    }
    __syncthreads();
    for (int i = tid; i < sh.nvis; i += NT) {
      const int4 pk = st_vis[2 * i + 1];  // {col0, row0, w | h << 16, bits(zmin)}
      L.vis[sh.bv + i] = st_vis[2 * i];
      L.bbox[sh.bv + i] = make_int4(pk.x, pk.y, pk.z & 0xFFFF, (int) ((u32) pk.z >> 16));
      L.zmin[sh.bv + i] = __int_as_float(pk.w);
    }
    for (int i = tid; i < sh.nfree; i += NT) L.cfree[sh.bf + i] = st_free[i];
    __syncthreads();
    if (tid == 0) { sh.nvis = 0; sh.nfree = 0; }
    __syncthreads();
  };
  const int grp = tid >> 2;      // block within the 64-block batch
  const int sub = tid & 3;       // this lane evaluates corners sub and sub + 4
  // one pass over descs[0, total): fine blocks (valbit 0, index = block index) or coarse units (MULTI)
  auto sweep_range = [&](const int4* __restrict__ descs, const uint2* __restrict__ sums, const int total, const u32 valbit) {
  const int chunk = (((total + n_sweep - 1) / n_sweep) + 63) & ~63;
  const int lo = sw * chunk, hi = min(total, lo + chunk);
  int4 d_next = make_int4(0, 0, 0, 0);
  uint2 sm_next = make_uint2(0, 0);
  if (lo + grp < hi) { d_next = descs[lo + grp]; sm_next = sums[lo + grp]; }
  for (int base = lo; base < hi; base += 64) {
    const int i = base + grp;
    const int4 d = d_next;
    const uint2 sm = sm_next;
    if (i + 64 < hi) { d_next = descs[i + 64]; sm_next = sums[i + 64]; }  // next batch in flight during this one
    // live and not inserted by this very launch (those are handled by their inserter)
    const bool live = i < hi && (d.w & 1) && ((u32) d.w >> 1) != stamp;
    if (__ballot(live) != 0) {  // wave-uniform: nothing live among this wave's 16 descriptors (common in the sparse coarse range)
      int any_approx = 0;
      float zmin = kFltMax, zmax = -kFltMax, umin = kFltMax, umax = -kFltMax, vmin = kFltMax, vmax = -kFltMax;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int corner = sub + 4 * h;
        const i3 v = mki3(d.x * kBlockSide + ((corner & 4) ? 7 : 0), d.y * kBlockSide + ((corner & 2) ? 7 : 0), d.z * kBlockSide + ((corner & 1) ? 7 : 0));
        const f3 pc = se3_apply(c.Ri, c.ti, voxel_to_world(m.vs, v));
        int r, cc;
        any_approx |= (SPH ? project_point_m<true>(c, pc, r, cc) : project_point<true>(c, pc, r, cc)) ? 1 : 0;
        zmin = fminf(zmin, pc.z); zmax = fmaxf(zmax, pc.z);
        if (!SPH && pc.z >= 0.05f) {
          const float u = c.fx * pc.x / pc.z + c.cx;
          const float w = c.fy * pc.y / pc.z + c.cy;
          umin = fminf(umin, u); umax = fmaxf(umax, u);
          vmin = fminf(vmin, w); vmax = fmaxf(vmax, w);
        }
      }
#pragma unroll
      for (int off = 1; off < 4; off <<= 1) {
        any_approx |= __shfl_xor(any_approx, off);
        zmin = fminf(zmin, __shfl_xor(zmin, off)); zmax = fmaxf(zmax, __shfl_xor(zmax, off));
        umin = fminf(umin, __shfl_xor(umin, off)); umax = fmaxf(umax, __shfl_xor(umax, off));
        vmin = fminf(vmin, __shfl_xor(vmin, off)); vmax = fmaxf(vmax, __shfl_xor(vmax, off));
      }
      if (sub == 0 && live && any_approx) {
        bool cull = !SPH && ((zmax <= c.min_depth - 1e-3f) || (zmin > c.max_depth + 1e-3f));
        if (!SPH && !cull && zmin >= 0.05f) cull = umax < -3.f || umin > (float) c.cols + 1.f || vmax < -3.f || vmin > (float) c.rows + 1.f;
        const int4 e = make_int4(d.x, d.y, d.z, (int) ((u32) i | valbit));
        if (!cull) {
          int4 bb = make_int4(0, 0, 0, 0);
          if (!SPH && zmin >= 0.05f) {
            int c0 = f2i_hw(floorf(umin + 0.5f)) - 1, c1 = f2i_hw(floorf(umax + 0.5f)) + 1;
            int r0 = f2i_hw(floorf(vmin + 0.5f)) - 1, r1 = f2i_hw(floorf(vmax + 0.5f)) + 1;
            c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0;
            c1 = c1 > c.cols - 1 ? c.cols - 1 : c1; r1 = r1 > c.rows - 1 ? c.rows - 1 : r1;
            const int bw = c1 - c0 + 1, bh = r1 - r0 + 1;
            if (bw > 0 && bh > 0 && bw * bh <= kTileMaxPx) bb = make_int4(c0, r0, bw | (bh << 16), __float_as_int(zmin));
          }
          const int k2 = atomicAdd(&sh.nvis, 1);
          st_vis[2 * k2] = e;
          st_vis[2 * k2 + 1] = bb;
        } else {
          bool collect = false;
          if (gc_on) {  // culled: untouched by this frame, so the stored summary already decides (vds.cu:1708-1711)
            collect = DEFER || (__uint_as_float(sm.x) >= trunc_threshold) || (sm.y == 0u);
          }
          if (collect) st_free[atomicAdd(&sh.nfree, 1)] = e;
          else atomicAdd(&sh.nkeep, 1);
        }
      }
    }
    // the staging areas hold at least 4 / 8 more batches than one iteration can add: flush only when nearly full
    __syncthreads();
    if (sh.nvis > kStVis - 64 || sh.nfree > kStFree - 64) flush();
  }
  };
  sweep_range(t.desc_fine, f.summary, hwm, 0u);
  if (MULTI) sweep_range(t.desc_coarse, f.summary_c, 8 * hwm, kValCoarseBit);
  flush();
  if (tid == 0 && sh.nkeep) atomicAdd(&t.ctr[cs + 1], sh.nkeep);
  MRH_TSF(4);
#ifdef MRH_TRACE
  if (tid == 0) f.trace[(kTraceFront + blockIdx.x) * 8 + 7] = (u64) (hwm / n_sweep) | (1ull << 63);
#endif
}

// allocateMemoryLow (vds.cu:860-871, host test :885-891) as extra workgroups of k_front: when the coarse free list has
// run low (flag, decided by the previous launch in the stream), each thread turns one fine slot into eight coarse units.
// Safe beside the allocation workgroups: both only POP the fine list (which slot a block gets is not observable), and
// nothing in k_front pops the coarse list.
__device__ __forceinline__ void front_refill(const Tab& t, const int low_blocks_to_allocate, const int* __restrict__ flag, const int wg) {
  if (*flag == 0) return;
  const int i = wg * 256 + (int) threadIdx.x;
  if (i >= low_blocks_to_allocate) return;
  const int addr_high = atomicSub(&t.ctr[CTR_HEAP_FINE], 1);
  if (addr_high < 0) { atomicAdd(&t.ctr[CTR_HEAP_FINE], 1); atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_POOL); return; }
  const u32 H = t.heap_fine[addr_high];
  atomicMax(&t.ctr[CTR_HWM_FINE], (int) H + 1);
  const int addr_low = atomicAdd(&t.ctr[CTR_HEAP_COARSE], 8);
  for (int idx = 1; idx <= 8; idx++) t.heap_coarse[addr_low + idx] = H * 8 + 8 - idx;
}

// LAZY (pipelined frames, see k_back<..., LZ = 2>): the launch may run next to the integration of earlier frames on another
// stream: its probes stamp Fast::want for every key they find, its sweep leaves the culled decisions to the integration.
template <bool PROFILE, bool MULTI, bool LAZY = false, bool SPH = false>
__global__ __launch_bounds__(256) void k_front(const Cam c, const Map m, const Tab t, const Fast f, const Lists L,
                                               const float* __restrict__ depth, const uint8_t* __restrict__ rgb, const int tiles_x,
                                               const int n_tiles, const u32 stamp, const int set, const int gc_on,
                                               const float trunc_threshold, const int n_refill, const int low_blocks_to_allocate,
                                               const int* __restrict__ refill_flag) {
  __shared__ FrontShared sh;
  const int cs = CTR_SET0 + 4 * set;
  // grid = [sweep | tiles | refill]: the sweep workgroups come FIRST so that they start first
  const int n_sweep = (int) gridDim.x - n_tiles - n_refill;
  MRH_TSF(0);
  if (MULTI && (int) blockIdx.x >= n_sweep + n_tiles) front_refill(t, low_blocks_to_allocate, refill_flag, (int) blockIdx.x - n_sweep - n_tiles);
  else if ((int) blockIdx.x >= n_sweep) front_tile<PROFILE, LAZY, SPH>(c, m, t, f, L, depth, rgb, tiles_x, (int) blockIdx.x - n_sweep, stamp, cs, sh);
  else front_sweep<MULTI, LAZY, SPH>(c, m, t, f, L, stamp, cs, gc_on, trunc_threshold, (int) blockIdx.x, n_sweep, sh);
}

// pixel footprint of a block computed by the wave that is about to integrate it (lanes 0..7 take one corner each):
// used for blocks inserted this frame, which k_front lists without a footprint.  Returns {0,0,0,0} when a corner
// is closer than 5 cm or the footprint exceeds the tile (the lookups then fall back to direct gathers).
__device__ __forceinline__ int4 wave_bbox(const Cam& c, const float vs, const int4 ent, const int lane, float& zmin_out) {
  const int corner = lane & 7;
  const i3 v = mki3(ent.x * kBlockSide + ((corner & 4) ? 7 : 0), ent.y * kBlockSide + ((corner & 2) ? 7 : 0), ent.z * kBlockSide + ((corner & 1) ? 7 : 0));
  const f3 pc = se3_apply(c.Ri, c.ti, voxel_to_world(vs, v));
  float zmin = pc.z;
  float umin = c.fx * pc.x / pc.z + c.cx, umax = umin;
  float vmin = c.fy * pc.y / pc.z + c.cy, vmax = vmin;
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    zmin = fminf(zmin, __shfl_xor(zmin, off));
    umin = fminf(umin, __shfl_xor(umin, off)); umax = fmaxf(umax, __shfl_xor(umax, off));
    vmin = fminf(vmin, __shfl_xor(vmin, off)); vmax = fmaxf(vmax, __shfl_xor(vmax, off));
  }
  int4 bb = make_int4(0, 0, 0, 0);
  if (zmin >= 0.05f) {
    int c0 = f2i_hw(floorf(umin + 0.5f)) - 1, c1 = f2i_hw(floorf(umax + 0.5f)) + 1;
    int r0 = f2i_hw(floorf(vmin + 0.5f)) - 1, r1 = f2i_hw(floorf(vmax + 0.5f)) + 1;
    c0 = c0 < 0 ? 0 : c0; r0 = r0 < 0 ? 0 : r0;
    c1 = c1 > c.cols - 1 ? c.cols - 1 : c1; r1 = r1 > c.rows - 1 ? c.rows - 1 : r1;
    const int bw = c1 - c0 + 1, bh = r1 - r0 + 1;
    if (bw > 0 && bh > 0 && bw * bh <= kTileMaxPx) bb = make_int4(c0, r0, bw, bh);
  }
  zmin_out = __shfl(zmin, 0);
  return make_int4(__shfl(bb.x, 0), __shfl(bb.y, 0), __shfl(bb.z, 0), __shfl(bb.w, 0));
}

// A visible-list record through the scalar unit: the entry index is wave-uniform, the lists were written by an earlier
// launch, and a scalar-cache hit returns in a fraction of a vector load's round trip — this load heads every wave's
// dependency chain (entry -> voxel-plane addresses -> HBM).
typedef int v4i_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_entry_scalar(const Lists& L, const int e, int4& ent, int4& bb, float& zmin) {
  v4i_ a, b;
  int z;
  const u32 off16 = (u32) e * 16u, off4 = (u32) e * 4u;
  asm volatile("s_load_dwordx4 %0, %3, %6\n\ts_load_dwordx4 %1, %4, %6\n\ts_load_dword %2, %5, %7\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(a), "=&s"(b), "=&s"(z)
               : "s"(L.vis), "s"(L.bbox), "s"(L.zmin), "s"(off16), "s"(off4)
               : "memory");
  ent = make_int4(a.x, a.y, a.z, a.w);
  bb = make_int4(b.x, b.y, b.z, b.w);
  zmin = __int_as_float(z);
}

// ---- multi-resolution maps (sdf_var_threshold > 0): coarse units on the fused path ---------------------------
// A coarse unit u = 8 H + k lives in fine slot H at byte k * 768 as f32[64] | f32[64] | u32[64] (mrh_device.h).
// Freeing one pushes the coarse free list, which the SAME launch pops when it coarsens fine blocks, and a stack
// cannot take pushes and pops concurrently: the unit goes on the deferred list, k_mr_tail pushes it after the launch.
__device__ __forceinline__ void wave_free_coarse(const Tab& t, const int4 ent, const int lane, u32* __restrict__ deferred) {
  const u32 u = (u32) ent.w & ~kValCoarseBit;
  if (lane == 0) {
    u64 key;
    pack_key(mki3(ent.x, ent.y, ent.z), key);
    hash_erase(t, key);
    t.desc_coarse[u].w = 0;
    deferred[atomicAdd(&t.ctr[CTR_NREINT], 1)] = u;
  }
  u32* p = (u32*) (t.pool + (size_t) (u >> 3) * kFineBytes + (size_t) (u & 7) * kCoarseBytes);
#pragma unroll
  for (int k = 0; k < kCoarseBytes / 4 / kWave; k++) p[k * kWave + lane] = 0u;
}
__device__ __forceinline__ void wave_free_any(const Tab& t, const int4 ent, const int lane, u32* __restrict__ deferred) {
  if ((u32) ent.w & kValCoarseBit) wave_free_coarse(t, ent, lane, deferred);
  else wave_free_block(t, ent, lane);
}

// integrateDepthMapKernel on a coarse unit (64 voxels at twice the spacing, vds.cu:1114-1118), lane = voxel, then the
// GC summary of the unit.  first_n < 64 restates reintegrateDepthMapKernel's launch shape (voxels 0..31 only, no
// variance term: D2 in the oracle header).  Returns the GC decision (wave-uniform).
template <bool VARIANCE>
__device__ __forceinline__ bool coarse_block(const Cam& c, const Map& m, const Tab& t, const Fast& f, const float* __restrict__ depth,
                                             const uint8_t* __restrict__ rgb, const int4 ent, const int lane, const int first_n,
                                             const float trunc_threshold) {
  const u32 val = (u32) ent.w;
  const u32 u = val & ~kValCoarseBit;
  const VoxPtr vp = vox_ptr(t, val);
  const int v = lane;
  float sd = vp.sdf[v], sq = vp.sumsq[v];
  u32 rw = vp.rgbw[v];
  if (v < first_n) {
    const i3 pi = mki3(ent.x * kBlockSide + 2 * (v & 3), ent.y * kBlockSide + 2 * ((v >> 2) & 3), ent.z * kBlockSide + 2 * (v >> 4));
    if (integrate_voxel<VARIANCE>(c, m, depth, rgb, pi, &sd, &sq, &rw)) { vp.sdf[v] = sd; vp.sumsq[v] = sq; vp.rgbw[v] = rw; }
  }
  const u32 wk = rw >> 24;
  const float mn = __uint_as_float(wave_min_u32(umin_(0x7F7FFFFFu, wk != 0 ? (__float_as_uint(sd) & 0x7FFFFFFFu) : 0xFFFFFFFFu)));
  const u32 mx = wave_max_u32(wk);
  if (lane == 0) f.summary_c[u] = make_uint2(__float_as_uint(mn), mx);
  return mn >= trunc_threshold || mx == 0;
}

// checkVarSDFKernel's decision for one fine block (vds.cu:1857-1939), from the block's (sum_squared, weight) pairs
// parked in LDS: thread `lane` sums its 2x2x2 sub-cube in the kernel's dz, dy, dx order, then the kernel's 64-thread
// shfl_down tree — the same association order, so the same float, so the same decision.
__device__ __forceinline__ bool variance_says_coarsen(const Map& m, const uint2* __restrict__ sv, const int lane) {
  float local_sum_sq = 0.f, local_weight = 0.f;
  const int gx = (lane % 4) * 2, gy = ((lane / 4) % 4) * 2, gz = (lane / 16) * 2;
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const uint2 e = sv[(gz + dz) * 64 + (gy + dy) * 8 + (gx + dx)];
        const u32 w = e.y >> 24;
        if (w > 0) { local_sum_sq += __uint_as_float(e.x); local_weight += (float) w; }
      }
  for (int stride = 32; stride > 0; stride >>= 1) {
    const float os = __shfl_down(local_sum_sq, stride);
    const float ow = __shfl_down(local_weight, stride);
    if (lane < stride) { local_sum_sq += os; local_weight += ow; }
  }
  int coarsen = 0;
  if (lane == 0 && !(local_weight < 2)) {
    const double avg_var = (double) (local_sum_sq / (local_weight - 1));
    if ((local_weight - 1) > 1e-6f && avg_var > 0.f && avg_var < (double) m.var_threshold) coarsen = 1;
  }
  return __shfl(coarsen, 0) != 0;
}

// Integration of a list of blocks, one wave per block (integrateDepthMapKernel vds.cu:1095-1181 + GC summary + GC
// decision, vds.cu:1674-1713); wave `gw` of `nw` takes entries gw, gw + nw, ...
//   FREE  false: no GC here (starve frames decide after the weights changed); true: free on the spot (tombstone the
//         key, push the free list, zero the 6 KiB)
//   MULTI multi-resolution map: entries may be coarse units; a fine block is variance-checked right after its update
//         and, if the reference would coarsen it, converted on the spot (checkVarSDF -> reallocBlocks ->
//         reintegrateDepthMap, vds.cu:1857-2107): same table slot, new coarse unit, fine slot zeroed and released
// ---- lazy garbage collection of pipelined frames ---------------------------------------------------------------------
// A pipelining context (mrh_capi.hip: integrate_lazy) launches the FRONT half of frame g + 1 (k_front<..., LAZY>: rays, probes,
// inserts, sweep) on a second stream, where it runs next to the integration of frame g (k_back<..., LZ = 2>) — and possibly of
// frame g - 1 — instead of behind it.  A block that a frame's garbage collection empties can therefore not leave the table in
// that launch: erasing a key would break the invariant the lock-free insert rests on (an occupied slot stays occupied while
// inserts run), pushing the free list would race with the pops, and the probes of the later frame could not tell "still
// there" from "gone".  Instead such a block becomes a ZOMBIE: payload zeroed (as a freed block's is), summary {FLT_MAX,
// kZombieBit}, key, descriptor and pool slot kept.  In the reference the block is gone after frame g and comes back — empty —
// iff a ray of a later frame crosses it inside that frame's approx frustum (allocBlocksKernel); here the rays of every frame
// stamp Fast::want[slot] (one array per frame in flight) of every key they FIND, and whoever meets the zombie on a frame's
// lists checks that frame's stamp: wanted -> it IS that fresh block (zero payload, nothing to insert), not wanted -> it is not
// a block of that frame and is skipped.  Which pool slot a block occupies is not observable, so "freed and re-inserted" and
// "emptied and kept" are the same map.  Zombies that nobody wants are taken out of the table by k_reclaim, which runs alone
// (every few frames, and before anything that is not a pipelined frame looks at the map).  The two halves exchange nothing
// inside a launch: every hand-over (lists, stamps, summaries) crosses a kernel boundary and a stream dependency.
__device__ __forceinline__ void wave_zombify(const Tab& t, const Fast& f, const int4 ent, const int lane) {
  const u32 H = (u32) ent.w;
  if (lane == 0) {
    f.summary[H] = make_uint2(0x7F7FFFFFu, kZombieBit);
    // the list holds one entry per pool block; a block can be listed again every frame (emptied, wanted, emptied again), so on a
    // small pool with heavy churn the count may pass the capacity before the next reclaim: the entry is then left out and
    // k_reclaim, seeing a count above the capacity, finds the zombies by their flag instead of by the list
    const int zi = atomicAdd(&t.ctr[CTR_ZOMBIES], 1);
    if ((u32) zi < f.zlist_cap) f.zlist[zi] = ent;
  }
  uint4* p = (uint4*) (t.pool + (size_t) H * kFineBytes);
  const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < kFineBytes / 16 / kWave; k++) p[k * kWave + lane] = z;
}
// is the zombie behind list entry `ent` wanted by the frame with stamp `want_stamp`?  (wave-uniform; lane 0 walks the table)
__device__ __forceinline__ bool zombie_wanted(const Tab& t, const Fast& f, const int4 ent, const u32 want_stamp, const int lane) {
  int wanted = 0;
  if (lane == 0) {
    u64 key;
    pack_key(mki3(ent.x, ent.y, ent.z), key);
    const int slot = hash_find(t, key);
    wanted = (slot >= 0 && f.want[slot] == want_stamp) ? 1 : 0;
  }
  return __builtin_amdgcn_readfirstlane(wanted) != 0;
}

// LZ: 0 = no zombies can exist (strict launches of contexts that never pipeline); 1 = zombie-aware, collected blocks are freed
// on the spot (a launch that runs alone); 2 = zombie-aware, collected blocks become zombies (pipelined frames)
// SPH: spherical camera model — voxels project through atan2 / asin (project4_sph), "depth" of a voxel is its range; entries carry
// no footprint, so every pixel lookup is a gather and the behind-the-surface early-out does not apply
template <bool FREE, bool PROFILE, bool MULTI, bool SAFEDIV, int LZ = 0, bool SPH = false>
__device__ __forceinline__ void back_range(const Cam& c, const Map& m, const Tab& t, const Fast& f, const Lists& L,
                                           const float trunc_threshold, const int n, const int gw, const int nw, const int lane,
                                           uint2* tile, const float* __restrict__ depth_raw, const uint8_t* __restrict__ rgb_raw,
                                           u32* __restrict__ deferred, const u32 want_stamp = 0) {
  static_assert(!(MULTI && LZ), "multi-resolution maps are not pipelined");
  static_assert(!(MULTI && SPH), "multi-resolution maps under the spherical model take the general kernels");
  for (int e = __builtin_amdgcn_readfirstlane(gw); e < n; e += nw) {
    MRH_TS(0);
    int4 ent, bb;
    float zmin;
    load_entry_scalar(L, e, ent, bb, zmin);
    if (MULTI && ((u32) ent.w & kValCoarseBit)) {
      const bool collect = coarse_block<true>(c, m, t, f, depth_raw, rgb_raw, ent, lane, kCoarseVoxels, trunc_threshold);
      if (FREE && collect) {
        wave_free_coarse(t, ent, lane, deferred);
        if (PROFILE && lane == 0) atomicAdd(&t.prof[PROF_FREED], 1ull);
      }
      continue;
    }
    const u32 H = (u32) ent.w;
    float4* ps = (float4*) (t.pool + (size_t) H * kFineBytes);
    float4* pq = ps + 128;
    uint4* pw = (uint4*) (ps + 256);
    uint2 sm0 = make_uint2(0u, 0u);
    if (LZ) sm0 = f.summary[H];  // requested first: it is needed first (zombie?), and loads return in order
    float4 S[2];
    uint4 W[2];
#pragma unroll
    for (int b = 0; b < 2; b++) { S[b] = ps[lane + 64 * b]; W[b] = pw[lane + 64 * b]; }
    float4 Q[2];  // MULTI: the variance check needs sum_squared of the voxels this frame does not update, too
    if (MULTI) {
#pragma unroll
      for (int b = 0; b < 2; b++) Q[b] = pq[lane + 64 * b];
    }
    if (!SPH && bb.z == 0) bb = wave_bbox(c, m.vs, ent, lane, zmin);  // listed without a footprint (inserted this frame)
    Proj4 P[2];
    float d[2][4];
    u32 cpx[2][4];
    MRH_TS(1);
    TileRegs tr;
    tile_issue(c, f, bb, lane, tr);  // footprint gathers in flight behind the voxel planes ...
#pragma unroll
    for (int b = 0; b < 2; b++) P[b] = SPH ? project4_sph(c, m, ent, lane + 64 * b) : project4(c, m, ent, lane + 64 * b);  // ... while the projections (entry-only) run
    if (LZ && (sm0.y & kZombieBit)) {  // wave-uniform; rare
      if (!zombie_wanted(t, f, ent, want_stamp, lane)) {  // emptied by the previous frame's GC and no ray of this frame crosses it:
        if (lane == 0) atomicAdd(&t.ctr[CTR_ZSKIP], 1);   // not a block of this frame (the sweep listed it by its descriptor)
        continue;
      }
      // wanted: this IS the block allocBlocksKernel would have inserted for this frame — empty payload, fresh summary
      sm0 = make_uint2(0x7F7FFFFFu, 0u);
      if (lane == 0) f.summary[H] = sm0;
      if (PROFILE && lane == 0) atomicAdd(&t.prof[PROF_INSERTED], 1ull);
    }
    const float reach = tile_commit(c, m, f, bb, lane, tile, tr);
    // Early out (exact): camera-frame z is monotone in each voxel coordinate under fp32 rounding, so every voxel of
    // the block has pc.z >= zmin (the smallest corner value); every in-image voxel projects into the footprint
    // (convexity, +-1 px slack).  If d + trunc(d) + 1e-4 <= zmin for every valid footprint pixel, then
    // fl(d - pc.z) <= -trunc(d) for every voxel (1e-4 >> the rounding of the two sums), i.e. integrateDepthMapKernel
    // (vds.cu:1134-1145) updates nothing: the block, and therefore its stored summary, stay as they are.
    const bool skip = bb.z != 0 && __uint_as_float(wave_max_u32(__float_as_uint(reach))) + 1e-4f <= zmin;  // wave-uniform
    float mn;
    u32 mx;
#ifdef MRH_TRACE
    u32 trace_upd = 0;
#endif
    if (skip) {
      const uint2 sm = LZ ? sm0 : f.summary[H];
      mn = __uint_as_float(sm.x);
      mx = sm.y;
    } else {
      MRH_TS(2);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      MRH_TS(3);
      tile_lookup<2>(f, c.cols, bb, tile, P, d, cpx);
      __builtin_amdgcn_wave_barrier();
      MRH_TS(4);
      u32 mnb = 0x7F7FFFFFu;  // bits of min |sdf| over weighted voxels: |x| >= 0, so unsigned order == float order, and
      mx = 0;                 // NaN / inf patterns sort above FLT_MAX exactly as fminf ignores them
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int q = lane + 64 * b;
        float s[4] = {S[b].x, S[b].y, S[b].z, S[b].w};
        u32 w[4] = {W[b].x, W[b].y, W[b].z, W[b].w};
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
        if (MULTI) { ss[0] = Q[b].x; ss[1] = Q[b].y; ss[2] = Q[b].z; ss[3] = Q[b].w; }
        const u32 mask = update_mask4(c, m, P[b], d[b]);
#ifdef MRH_TRACE
        trace_upd += __popc(mask);
#endif
        blend4<SAFEDIV>(m, P[b], mask, d[b], cpx[b], s, w, ss);
        if (MULTI) {  // (sum_squared, rgbw) of this lane's voxels q * 4 + k for the variance check below
          W[b] = make_uint4(w[0], w[1], w[2], w[3]);
          Q[b] = make_float4(ss[0], ss[1], ss[2], ss[3]);
        }
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
          const u32 ab = __float_as_uint(s[k]) & 0x7FFFFFFFu;
          mnb = umin_(mnb, wk != 0 ? ab : 0xFFFFFFFFu);
          mx = umax_(mx, wk);
        }
      }
      mn = __uint_as_float(wave_min_u32(mnb));
      mx = wave_max_u32(mx);
      MRH_TS(5);
      if (MULTI) {
        __builtin_amdgcn_wave_barrier();  // every lane is done with the pixel tile: its LDS now holds the (sum_sq, rgbw) pairs
        uint2* sv = tile;
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const int q = lane + 64 * b;
          sv[q * 4 + 0] = make_uint2(__float_as_uint(Q[b].x), W[b].x);
          sv[q * 4 + 1] = make_uint2(__float_as_uint(Q[b].y), W[b].y);
          sv[q * 4 + 2] = make_uint2(__float_as_uint(Q[b].z), W[b].z);
          sv[q * 4 + 3] = make_uint2(__float_as_uint(Q[b].w), W[b].w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool coarsen = variance_says_coarsen(m, sv, lane);
        __builtin_amdgcn_wave_barrier();
        if (coarsen) {
          // the block keeps its table slot; its payload moves to a coarse unit, the fine slot is zeroed and released
          int uval = -1;
          if (lane == 0) {
            u64 key;
            pack_key(mki3(ent.x, ent.y, ent.z), key);
            const int slot = hash_find(t, key);
            const int idx = atomicSub(&t.ctr[CTR_HEAP_COARSE], 1);
            if (idx < 0 || slot < 0) {
              atomicAdd(&t.ctr[CTR_HEAP_COARSE], 1);
              if (slot >= 0) t.keys[slot] = kKeyTomb;
              atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_POOL);
            } else {
              const u32 u = t.heap_coarse[idx];
              t.vals[slot] = u | kValCoarseBit;
              t.desc_coarse[u] = make_int4(ent.x, ent.y, ent.z, 1);
              uval = (int) u;
            }
            const int fi = atomicAdd(&t.ctr[CTR_HEAP_FINE], 1);
            t.heap_fine[fi + 1] = H;  // vds.cu:53-57
            t.desc_fine[H].w = 0;
          }
          uint4* pz = (uint4*) (t.pool + (size_t) H * kFineBytes);
          const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
          for (int k = 0; k < kFineBytes / 16 / kWave; k++) pz[k * kWave + lane] = z;
          uval = __shfl(uval, 0);
          if (uval >= 0) {
            const int4 ce = make_int4(ent.x, ent.y, ent.z, (int) ((u32) uval | kValCoarseBit));
            const bool collect = coarse_block<false>(c, m, t, f, depth_raw, rgb_raw, ce, lane, 32, trunc_threshold);
            if (FREE && collect) {
              wave_free_coarse(t, ce, lane, deferred);
              if (PROFILE && lane == 0) atomicAdd(&t.prof[PROF_FREED], 1ull);
            }
          }
          continue;
        }
      }
      if (lane == 0) f.summary[H] = make_uint2(__float_as_uint(mn), mx);
    }
    if (FREE && (mn >= trunc_threshold || mx == 0)) {
      if (LZ == 2) wave_zombify(t, f, ent, lane);
      else wave_free_block(t, ent, lane);
      if (PROFILE && lane == 0) atomicAdd(&t.prof[PROF_FREED], 1ull);
    }
    MRH_TS(6);
#ifdef MRH_TRACE
    for (int off = 32; off > 0; off >>= 1) trace_upd += __shfl_xor(trace_upd, off);
    if (lane == 0) { u32 hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); f.trace[(size_t) e * 8 + 7] = hw | ((u64) trace_upd << 32) | ((u64) ((bb.z * bb.w) & 0x7FFF) << 48) | ((u64) (skip ? 1 : 0) << 63); }
#endif
  }
}

// frees entries [0, n) of the culled-free (+ deferred-free) list, one wave per block
template <bool PROFILE, bool MULTI>
__device__ __forceinline__ void free_range(const Tab& t, const Lists& L, const int n, const int gw, const int nw, const int lane,
                                           u32* __restrict__ deferred) {
  for (int e = gw; e < n; e += nw) {
    if (MULTI) wave_free_any(t, L.cfree[e], lane, deferred);
    else wave_free_block(t, L.cfree[e], lane);
    if (PROFILE && lane == 0) atomicAdd(&t.prof[PROF_FREED], 1ull);
  }
}
// The culled list of a pipelined frame holds CANDIDATES (front_sweep<DEFER>): every culled block, undecided.  One lane per
// candidate reads the now-final summary (vds.cu:1708-1711), the wave then empties the blocks to collect one after the other.
// A zombie on the list is not a block of this frame (no voxel of a culled block can be updated: wanted or not, the reference
// would have inserted it empty and collected it again).
template <bool PROFILE, int LZ>
__device__ __forceinline__ void free_candidates(const Tab& t, const Fast& f, const Lists& L, const float trunc_threshold, const int n, const int gw,
                                                const int nw, const int lane, const u32 want_stamp) {
  for (int base = gw * kWave; base < n; base += nw * kWave) {
    const int e = base + lane;
    int4 ent = make_int4(0, 0, 0, 0);
    bool collect = false;
    if (e < n) {
      ent = L.cfree[e];
      const uint2 sm = f.summary[ent.w];
      collect = !(sm.y & kZombieBit) && ((__uint_as_float(sm.x) >= trunc_threshold) || (sm.y == 0u));
      if (sm.y & kZombieBit) {  // counted as a block of this frame only if its rays want it (the frame's M, mrh_get_stats)
        u64 key;
        pack_key(mki3(ent.x, ent.y, ent.z), key);
        const int slot = hash_find(t, key);
        if (!(slot >= 0 && f.want[slot] == want_stamp)) atomicAdd(&t.ctr[CTR_ZSKIP], 1);
      }
    }
    u64 todo = __ballot(collect);
    if (PROFILE && todo && lane == 0) atomicAdd(&t.prof[PROF_FREED], (u64) __popcll(todo));
    while (todo) {
      const int src = __ffsll((long long) todo) - 1;
      todo &= todo - 1;
      const int4 b = make_int4(__shfl(ent.x, src), __shfl(ent.y, src), __shfl(ent.z, src), __shfl(ent.w, src));
      if (LZ == 2) wave_zombify(t, f, b, lane);
      else wave_free_block(t, b, lane);
    }
  }
}

// end of a frame: the counters of set `zero_set` are zeroed for a later frame's appends; stats mirror for the host
__device__ __forceinline__ void frame_epilogue(const Tab& t, const int zero_set, const int n_integrated, const int n_culled, const int lane, const int seq) {
  if (lane < 4) t.ctr[CTR_SET0 + 4 * zero_set + lane] = 0;
  if (lane == 0 && t.h_levels) {  // for the host's choice of the next launch (pinned memory): pool level and zombie count as this
    t.h_levels[0] = t.ctr[CTR_HEAP_FINE];  // launch found them, and the sequence number of the frame whose integration has now
    t.h_levels[1] = t.ctr[CTR_ZOMBIES];    // STARTED — every earlier integration is complete, its buffers are free
    t.h_levels[2] = seq;
  }
  if (lane == 0) {
    t.ctr[CTR_COMPACT] = n_integrated;  // M = visible + culled
    t.ctr[CTR_CULLED] = n_culled;
    t.ctr[CTR_FREED_EARLY] = 0;
  }
}

// ---- K2 = integrate + summary + GC of the visible list, then the culled-free list -------------
// `gw` of `nw` waves; `tile`: this wave's LDS pixel tile; `set`: the frame's list-counter set, `zero_set`: the one to clear
template <bool FREE, bool PROFILE, bool MULTI, bool SAFEDIV, int LZ = 0, bool SPH = false>
__device__ __forceinline__ void back_role(const Cam& c, const Map& m, const Tab& t, const Fast& f, const Lists& L, const int set, const int zero_set,
                                          const float trunc_threshold, const float* __restrict__ depth_raw, const uint8_t* __restrict__ rgb_raw,
                                          u32* __restrict__ deferred, const int gw, const int nw, const int lane, uint2* tile, const u32 want_stamp = 0,
                                          const int seq = 0) {
  const int cs = CTR_SET0 + 4 * set;
  // the counters are read (scalar loads) BEFORE wave 0 stores to the counter array: a load after those stores would
  // have to be a vector load with a full memory round trip at the head of every wave
  const int nvis = t.ctr[cs + 0];
  const int ncfree = FREE ? t.ctr[cs + 2] : 0;
  const int nkept = t.ctr[cs + 1];
  if (gw == 0) frame_epilogue(t, zero_set, nvis, nkept + t.ctr[cs + 2], lane, seq);
  back_range<FREE, PROFILE, MULTI, SAFEDIV, LZ, SPH>(c, m, t, f, L, trunc_threshold, nvis, gw, nw, lane, tile, depth_raw, rgb_raw, deferred, want_stamp);
  if (FREE) {
    // a zombie-aware launch takes every culled-list entry as a candidate: the list may come from a sweep that could not decide
    // (front_sweep<DEFER>), and for one that did decide the then-stable summary gives the same answer again
    if (LZ) free_candidates<PROFILE, LZ>(t, f, L, trunc_threshold, ncfree, gw, nw, lane, want_stamp);
    else free_range<PROFILE, MULTI>(t, L, ncfree, gw, nw, lane, deferred);
  }
}

// LZ = 2: the integration of a pipelined frame (zombies on its lists: `want_stamp` = the stamp of ITS frame; its culled list
// holds candidates; what it collects becomes a zombie).  LZ = 1: the same reading, collected blocks freed on the spot.
// `seq`: the frame's sequence number in a pipelining context (reported to the host through Tab::h_levels).
template <bool FREE, bool PROFILE, bool MULTI, bool SAFEDIV, int LZ = 0, bool SPH = false>
__global__ __launch_bounds__(256) void k_back(const Cam c, const Map m, const Tab t, const Fast f, const Lists L, const int set, const int zero_set,
                                              const float trunc_threshold, const float* __restrict__ depth_raw,
                                              const uint8_t* __restrict__ rgb_raw, u32* __restrict__ deferred, const u32 want_stamp, const int seq) {
  extern __shared__ __attribute__((aligned(16))) uint2 s_tile[];
  const int wpw = blockDim.x >> 6;
  back_role<FREE, PROFILE, MULTI, SAFEDIV, LZ, SPH>(c, m, t, f, L, set, zero_set, trunc_threshold, depth_raw, rgb_raw, deferred, blockIdx.x * wpw + (threadIdx.x >> 6),
                                               gridDim.x * wpw, threadIdx.x & 63, &s_tile[(threadIdx.x >> 6) * kTileMaxPx], want_stamp, seq);
}

// Takes the zombies nobody wanted out of the table (deleteHashEntryElement + the free-list push that their garbage collection
// left out).  Runs ALONE on the stream — no probe, insert or pop next to it.  A block on the list twice (emptied, wanted,
// emptied again) or one that lives again (wanted: its summary was rewritten) is recognised by the flag, cleared atomically.
__global__ __launch_bounds__(256) void k_reclaim(const Tab t, const Fast f) {
  const int n = t.ctr[CTR_ZOMBIES];
  const bool by_flag = (u32) n > f.zlist_cap;  // the list overflowed (wave_zombify): every block below the high-water mark is a candidate
  const int total = by_flag ? t.ctr[CTR_HWM_FINE] : n;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    int4 ent;
    if (by_flag) {
      ent = t.desc_fine[e];
      if (!(ent.w & 1)) continue;
      ent.w = e;
    } else {
      ent = f.zlist[e];
    }
    const u32 H = (u32) ent.w;
    const u32 old = atomicAnd(&f.summary[H].y, ~kZombieBit);
    if (!(old & kZombieBit)) continue;
    u64 key;
    pack_key(mki3(ent.x, ent.y, ent.z), key);
    hash_erase(t, key);
    const int idx = atomicAdd(&t.ctr[CTR_HEAP_FINE], 1);
    t.heap_fine[idx + 1] = H;  // vds.cu:53-57
    t.desc_fine[H].w = 0;
  }
}
__global__ void k_reclaim_done(const Tab t) { t.ctr[CTR_ZOMBIES] = 0; }

// after a multi-resolution k_back: the coarse units freed during the launch go onto the coarse free list, and the
// refill test of the NEXT frame (vds.cu:885-891: coarse list below low_blocks_to_allocate?) is taken here, on the final
// level — the next k_front carries the refill itself, so a steady-state frame is three launches
__global__ __launch_bounds__(256) void k_mr_tail(const Tab t, const u32* __restrict__ deferred, const int low_blocks_to_allocate,
                                                 int* __restrict__ refill_flag) {
  const int n = t.ctr[CTR_NREINT];
  __shared__ int s_base;
  if (threadIdx.x == 0) s_base = n ? atomicAdd(&t.ctr[CTR_HEAP_COARSE], n) : t.ctr[CTR_HEAP_COARSE];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) t.heap_coarse[s_base + 1 + i] = deferred[i];
  if (threadIdx.x == 0) {
    t.ctr[CTR_NREINT] = 0;
    *refill_flag = (s_base + n + 1 < low_blocks_to_allocate) ? 1 : 0;
  }
}

// GC summaries of the blocks on this frame's visible list (single-resolution maps): after the starve step, which only
// decrements weights of voxels of exactly those blocks (k_starve walks compact[0 .. CTR_COMPACT)) — every other block's
// summary still describes its payload
__global__ __launch_bounds__(256) void k_summarize_visible(const Tab t, const Fast f) {
  const int n = t.ctr[CTR_COMPACT];
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  for (int e = gw; e < n; e += nw) {
    const u32 H = (u32) t.compact[e].w;
    const VoxPtr vp = vox_ptr(t, H);
    u32 mnb = 0x7F7FFFFFu, mx = 0;
    for (int v = lane; v < 512; v += 64) {
      const u32 wk = vp.rgbw[v] >> 24;
      mnb = umin_(mnb, wk != 0 ? (__float_as_uint(vp.sdf[v]) & 0x7FFFFFFFu) : 0xFFFFFFFFu);
      mx = umax_(mx, wk);
    }
    mnb = wave_min_u32(mnb);
    mx = wave_max_u32(mx);
    if (lane == 0) f.summary[H] = make_uint2(mnb, mx);
  }
}

// GC summaries of every live block and coarse unit from their payload (one wave each): run once when the fused
// multi-resolution path takes over from frames that went through the general kernels (mrh_kernels.h)
__global__ __launch_bounds__(256) void k_summarize_all(const Tab t, const Fast f) {
  const int hwm = t.ctr[CTR_HWM_FINE];
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  for (int i = gw; i < (t.multi_res ? 9 : 1) * hwm; i += nw) {
    const bool coarse = i >= hwm;
    const u32 idx = coarse ? (u32) (i - hwm) : (u32) i;
    const int4 d = coarse ? t.desc_coarse[idx] : t.desc_fine[idx];
    if (!(d.w & 1)) continue;
    const VoxPtr vp = vox_ptr(t, coarse ? (idx | kValCoarseBit) : idx);
    u32 mnb = 0x7F7FFFFFu, mx = 0;
    for (int v = lane; v < (coarse ? kCoarseVoxels : 512); v += 64) {
      const u32 wk = vp.rgbw[v] >> 24;
      mnb = umin_(mnb, wk != 0 ? (__float_as_uint(vp.sdf[v]) & 0x7FFFFFFFu) : 0xFFFFFFFFu);
      mx = umax_(mx, wk);
    }
    mnb = wave_min_u32(mnb);
    mx = wave_max_u32(mx);
    if (lane == 0) (coarse ? f.summary_c : f.summary)[idx] = make_uint2(mnb, mx);
  }
}

// ---- starve frames on the two-launch path (single-resolution maps) -------------------------------------------------------------
// voxel_data_structures.cpp:139 — every n_frames_invalidate_voxels-th frame the front-most voxel of every pixel loses one unit of
// weight (starveVoxelsKernel vds.cu:1597-1649) BEFORE garbage collection looks at the frame's blocks.  k_starve (mrh_kernels.h)
// states the z-buffer with its canonical 72-bit key in three passes; a frame used to be: flush of the pipeline + reclaim, serial
// k_front, k_back<no free>, zbuf fill, three passes, k_summarize_visible, k_free_lists — eight launches, 0.2-0.4 ms, and a host
// synchronisation in front of the next pipelined frame.  Here the frame stays a frame of the pipeline:
//   k_back<FREE = false>      integrates, collects nothing (the front half of the next frame runs next to all of this as usual)
//   k_starve_z<0>, <1>        the two min-passes over the frame's own visible list (ring slot), one wave per block, skipping the entries the integration
//                             skipped: a zombie nobody wanted still carries its flag, a wanted one had its summary rewritten
//   k_starve_tail<LZ>         pass 2 (the winner's weight), the block's summary from its planes as they now are, the GC decision
//                             (visible list by that summary, culled list as free_candidates / free_range do) — zombies under
//                             LZ = 2, frees on a serial frame —, and the OTHER pair of z-buffers back to "empty" for the next
//                             starve frame (the pair in use is still being read by other workgroups)
// Same keys, same winner, same decrement, same summaries, same decisions as the eight launches.
// One wave per block, eight voxels per lane, through project4 / project4_sph — integrate_voxel's arithmetic, which IS the starve
// kernel's: the same se3_apply association, the same IEEE quotients (div_rr), `depth_ok && in image` == `!(dep < min_depth) &&
// projectPoint` (camera.cuh:131-164) — at a third of the instructions of one thread per voxel with compiler divisions.
// Voxel q * 4 + k of float4-group q: its linear index in the block IS 4 q + k (x + 8 y + 64 z), the `v` of the 72-bit key.
struct StarveKeys {
  u32 pix[4];  // pixel index (valid voxels only)
  u32 dep[4];  // depth bits
  u32 mask;
};
template <bool SPH>
__device__ __forceinline__ StarveKeys starve_keys4(const Cam& c, const Map& m, const int4 ent, const int q) {
  const Proj4 P = SPH ? project4_sph(c, m, ent, q) : project4(c, m, ent, q);
  StarveKeys k;
  k.mask = P.mask;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    k.pix[i] = ((P.mask >> i) & 1u) ? (u32) (__mul24(P.row[i], c.cols) + P.col[i]) : 0u;
    k.dep[i] = __float_as_uint(P.pcz[i]);
  }
  return k;
}
// (Pass 0 is ~1 M successful min-updates of 0.3 M words — 80 of the step's 125 us.  One copy of the buffer per XCD, indexed by
// HW_REG_XCC_ID and updated with L2-local atomics, then folded by a merge launch, was measured in round 6: each atomic is three
// times cheaper, but a pixel sees a new minimum 1.7 times per XCD instead of 3.6 times in all, i.e. there are four times as many
// of them: 92 + 9 us.  Pass 0 riding on the frame's integration (the voxels are projected there anyway: k_back<FREE = false> with the
// look + atomic per voxel) was measured, too: 109 us for the one launch against 31 + 79 for the two — the atomics are the time,
// and nothing of them hides under the integration's arithmetic.)
template <int PASS, bool SPH>
__global__ __launch_bounds__(256) void k_starve_z(const Cam c, const Map m, const Tab t, const Fast f, const int4* __restrict__ vis, const int set,
                                                  u64* __restrict__ zbuf0, u64* __restrict__ zbuf1) {
  const int count = t.ctr[CTR_SET0 + 4 * set];
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  for (int e = __builtin_amdgcn_readfirstlane(gw); e < count; e += nw) {
    const int4 ent = vis[e];
    if (f.summary[ent.w].y & kZombieBit) continue;  // not a block of this frame (lazy garbage collection); wave-uniform
    u64 key;
    pack_key(mki3(ent.x, ent.y, ent.z), key);
    const u32 key_hi = (u32) (key >> 31);              // with the depth bits: the top 64 of the 72-bit (depth, key63 << 9 | v)
    const u64 key_lo = (key & 0x7FFFFFFFull) << 9;     // the low 40 bits, before `| v`
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const int q = lane + 64 * b;
      const StarveKeys k = starve_keys4<SPH>(c, m, ent, q);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (!((k.mask >> i) & 1u)) continue;
        const u64 hi = ((u64) k.dep[i] << 32) | key_hi;
        // the values only fall: a plain look first saves the atomic for every voxel that is not in front of what the pixel
        // already holds (most of them; a stale look costs an atomic that changes nothing)
        if (PASS == 0) {
          if (zbuf0[k.pix[i]] > hi) atomicMin(&zbuf0[k.pix[i]], hi);
        } else if (zbuf0[k.pix[i]] == hi) {
          const u64 lo = key_lo | (u64) (q * 4 + i);
          if (zbuf1[k.pix[i]] > lo) atomicMin(&zbuf1[k.pix[i]], lo);
        }
      }
    }
  }
}
template <int LZ, bool SPH>
__global__ __launch_bounds__(256) void k_starve_tail(const Cam c, const Map m, const Tab t, const Fast f, const Lists L, const int set,
                                                     const float trunc_threshold, const u32 want_stamp, const u64* __restrict__ zbuf0,
                                                     const u64* __restrict__ zbuf1, u64* __restrict__ clear, const size_t n_clear) {
  const int cs = CTR_SET0 + 4 * set;
  const int nvis = t.ctr[cs + 0], ncfree = t.ctr[cs + 2];
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  for (int e = __builtin_amdgcn_readfirstlane(gw); e < nvis; e += nw) {
    const int4 ent = L.vis[e];
    const u32 H = (u32) ent.w;
    if (f.summary[H].y & kZombieBit) continue;  // wave-uniform
    float4* ps = (float4*) (t.pool + (size_t) H * kFineBytes);
    uint4* pw = (uint4*) (ps + 256);
    u64 key;
    pack_key(mki3(ent.x, ent.y, ent.z), key);
    const u32 key_hi = (u32) (key >> 31);
    const u64 key_lo = (key & 0x7FFFFFFFull) << 9;
    u32 mnb = 0x7F7FFFFFu, mx = 0;
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const int q = lane + 64 * b;
      const float4 S = ps[q];
      uint4 W = pw[q];
      const StarveKeys k = starve_keys4<SPH>(c, m, ent, q);
      u32 w[4] = {W.x, W.y, W.z, W.w};
      u32 won = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {  // pass 2: the unique winner of its pixel loses one unit of weight (vds.cu:1640-1647)
        if (!((k.mask >> i) & 1u)) continue;
        const u64 hi = ((u64) k.dep[i] << 32) | key_hi;
        if (zbuf0[k.pix[i]] == hi && zbuf1[k.pix[i]] == (key_lo | (u64) (q * 4 + i))) {
          const u32 wk = w[i] >> 24;
          w[i] = (w[i] & 0x00FFFFFFu) | ((wk > 0 ? wk - 1 : 0) << 24);
          won |= 1u << i;
        }
      }
      if (won) pw[q] = make_uint4(w[0], w[1], w[2], w[3]);
      const float s[4] = {S.x, S.y, S.z, S.w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const u32 wk = w[i] >> 24;
        mnb = umin_(mnb, wk != 0 ? (__float_as_uint(s[i]) & 0x7FFFFFFFu) : 0xFFFFFFFFu);
        mx = umax_(mx, wk);
      }
    }
    const u32 mn = wave_min_u32(mnb);
    mx = wave_max_u32(mx);
    if (lane == 0) f.summary[H] = make_uint2(mn, mx);
    if (__uint_as_float(mn) >= trunc_threshold || mx == 0) {  // vds.cu:1708-1711
      if (LZ == 2) wave_zombify(t, f, ent, lane);
      else wave_free_block(t, ent, lane);
      if (lane == 0) atomicAdd(&t.prof[PROF_FREED], 1ull);
    }
  }
  // the culled list: candidates under lazy garbage collection, decided entries on a serial frame
  if (LZ) free_candidates<true, LZ>(t, f, L, trunc_threshold, ncfree, gw, nw, lane, want_stamp);
  else free_range<true, false>(t, L, ncfree, gw, nw, lane, nullptr);
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n_clear; i += (size_t) gridDim.x * 256) clear[i] = 0x7FFFFFFFFFFFFFFFull;
}

// starve frames: GC after the weights changed — visible list by refreshed summary, plus the culled-free list
__global__ __launch_bounds__(256) void k_free_lists(const Tab t, const Fast f, const Lists L, const int set, const float trunc_threshold) {
  const int cs = CTR_SET0 + 4 * set;
  const int nvis = t.ctr[cs + 0], ncfree = t.ctr[cs + 2];
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  for (int e = gw; e < nvis + ncfree; e += nw) {
    const int4 ent = e < nvis ? L.vis[e] : L.cfree[e - nvis];
    bool fr = e >= nvis;
    if (!fr) {
      const uint2 sm = f.summary[ent.w];
      fr = (__uint_as_float(sm.x) >= trunc_threshold) || (sm.y == 0u);
    }
    if (fr) {
      wave_free_block(t, ent, lane);
      if (lane == 0) atomicAdd(&t.prof[PROF_FREED], 1ull);
    }
  }
}

}  // namespace mrh
