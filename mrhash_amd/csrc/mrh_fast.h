// mrh_fast.h — building blocks of the fast path (mrh_fast2.h holds the two launches): per-block summaries, the
// exact-image cull argument, four-voxel projection / update / blend, the per-wave LDS pixel tile.
//
// The reference runs integrate (vds.cu:1095-1181) and garbageCollectIdentify (vds.cu:1674-1713) as two full sweeps
// over every in-frustum block; here one pass per visible block does both (results identical: tests/test_parity_gpu.py).
// Earlier kernel structures of this round (five, four and three launches per frame) are kept as measurements only:
// profiles/r01/history_*.
#pragma once

#include "mrh_kernels.h"

namespace mrh {

constexpr int CTR_CULLED = 12;  // number of culled compact entries (stored from the back of Tab::compact)
constexpr int CTR_FREED_EARLY = 13;  // (kept zero: culled blocks are freed by k_back from the CULLED-FREE list)
constexpr float kFltMax = 3.402823466e+38f;
constexpr int kTileMaxPx = 576;  // LDS depth+colour tile per wave: 576 px x 8 B = 4.5 KiB (24 x 24 px: blocks beyond ~1.2 m)

struct Fast {
  uint2* dcx;          // [rows*cols] {bits of the cleaned depth (depth if in (min_depth, max_depth] else 0), r | g << 8 | b << 16}:
                       // one 8-byte access per pixel, and exactly the element of k_back's LDS footprint tile
  uint2* summary;      // [cap_blocks] {bits of min |sdf| over weighted voxels (FLT_MAX if none), max weight}
  u32 zlist_cap;       // entries `zlist` holds (= pool blocks; MRH_ZLIST_CAP shrinks it for the overflow test)
  int4* bbox;          // [pool blocks] per VISIBLE compact entry: pixel footprint {col0, row0, w, h}; w == 0: none
  uint2* summary_c;    // [8 * cap_blocks] the same summary per COARSE unit (multi-resolution maps only)
  u32* want;           // [hash slots] stamp of the last frame whose rays found this slot's key in the table (pipelined frames,
                       // mrh_fast2.h: decides whether a block that the previous frame's garbage collection emptied lives on)
  int4* zlist;         // [cap_blocks] blocks emptied by a pipelined frame's garbage collection and not yet taken out of the table
#ifdef MRH_TRACE
  u64* trace;          // [pool blocks * 8] per-block phase timestamps of the last k_back launch (tuning builds only)
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

// ---- exact-image cull (used by the descriptor sweep, mrh_fast2.h: front_sweep) ------------------------------
// A block in the reference's approx frustum (vds.cu:66-77) is VISIBLE if it may hold voxels that project into the
// image, CULLED if provably no voxel centre passes Camera::projectPoint (camera.cuh:131-147).
// Cull argument: voxel centres of a block are the integer lattice points of the box spanned by its 8 corner
// voxels; world->camera is affine, so every voxel's camera point is a convex combination of the 8 corner
// camera points.  z is therefore bounded by the corners' z range, and when every corner has z > 0 the ratios
// x/z, y/z are bounded by the corners' ratios.  Margins (1e-3 m in z, >= 1.5 px laterally, lateral test only
// when all corners have z >= 5 cm) cover the difference between exact and fp32 evaluation many times over.
// A culled block is not touched by the frame's integrate pass, so its GC decision follows from the stored summary.


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

// The same under the spherical camera model (Camera::projectPoint, camera.cuh:147-164, as project_point_m<false> states it): range,
// azimuth = atan2(y, x), elevation = asin(z / range) through mrh_softmath.h (D8); `pcz` carries getDepth(pc) = the range
// (camera.cuh:120-129), which is what integrateDepthMapKernel subtracts from the pixel's depth (vds.cu:1138).
__device__ __forceinline__ Proj4 project4_sph(const Cam& c, const Map& m, const int4 ent, const int q) {
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
    const float range = sqrtf(X * X + Y * Y + Z * Z);
    P.pcz[k] = range;
    const bool depth_ok = !(range < c.min_depth || range > c.max_depth);
    const float az = mrh_atan2f(Y, X);
    const float el = mrh_asinf(Z / range);
    const int row = f2i_hw((c.fy * el + c.cy) + 0.5f);
    const int col = f2i_hw((c.fx * az + c.cx) + 0.5f);
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
template <bool SAFEDIV, typename PT>
__device__ __forceinline__ void blend4(const Map& m, const PT& P, const u32 mask, const float (&d)[4], const u32 (&cpx)[4],
                                       float (&s)[4], u32 (&w)[4], float (&ss)[4]) {
  const u32 w1 = (u32) (m.weight_sample & 0xFF);
  const u32 wmax = (u32) (m.weight_max & 0xFF);
  const v2f half_vs = splat2(m.vs / 2), rh = splat2(m.r_half_vs), w1f = splat2((float) w1);
  // SAFEDIV: a context whose voxel size (or device) failed the checks of mrh_create keeps the second residual step
  constexpr bool two = SAFEDIV, two_w = SAFEDIV;
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
    const v2f delta = div_cr2(sd - curr_mean, half_vs, rh, two);
    const v2f wsum = mk2((float) (int) (w00 + w1), (float) (int) (w01 + w1));
    // weight sums are integers <= 510: for those, v_rcp_f32 + one Newton step already IS the correctly rounded reciprocal
    const v2f sn = div_cr2(s0 * mk2((float) w00, (float) w01) + sd * w1f, wsum, rcp_refined2(wsum), two_w);
    const v2f delta2 = div_cr2(sd - sn, half_vs, rh, two);
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
// Dense, row-contiguous fill of the block's pixel footprint {depth bits, colour}, split in two so that a wave can do
// memory-independent work (the projections) between issuing the gathers of the first 256 footprint pixels and parking
// them in LDS; footprints > 256 px finish through a second round.  tile_commit returns this lane's largest
// d + truncation(d) over the pixels it staged (k_back's early-out).
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
    const uint2 v = f.dcx[g];
    tr.dv[j] = __uint_as_float(v.x);
    tr.cv[j] = v.y;
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
        const uint2 v = f.dcx[g];
        dv[j] = __uint_as_float(v.x);
        cv[j] = v.y;
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
          const uint2 v = f.dcx[pix];
          d[b][k] = __uint_as_float(v.x);
          cpx[b][k] = v.y;
        }
  }
}

// profile mode only: U = voxels the next k_back launch will write (the predicate depends on pose, depth image and
// block list only, not on voxel contents), M = compact blocks.  Runs outside the timed bracket.
__global__ __launch_bounds__(256) void k_count_updates(const Cam c, const Map m, const Tab t, const Fast f, u64* __restrict__ partials,
                                                       const int merged_set, const int4* __restrict__ vis, const int4* __restrict__ cfree, const u32 want_stamp, const int zombies) {
  // merged_set >= 0: list counters of the two-launch path live at ctr[merged_set .. merged_set + 2]
  // zombies: entries of blocks the previous (pipelined) frame's GC emptied may be on the list (mrh_fast2.h); the ones this
  // frame's rays do not want are not blocks of this frame — neither updated nor counted
  const int nvis = merged_set >= 0 ? t.ctr[merged_set] : t.ctr[CTR_COMPACT];
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  u32 cnt = 0, zskip = 0, zcul = 0;
  for (int e = gw; e < nvis; e += nw) {
    const int4 ent = vis[e];
    if (zombies && (f.summary[ent.w].y & 0x80000000u)) {
      int wanted = 0;
      if (lane == 0) {
        u64 key;
        pack_key(mki3(ent.x, ent.y, ent.z), key);
        const int slot = hash_find(t, key);
        wanted = (slot >= 0 && f.want[slot] == want_stamp) ? 1 : 0;
      }
      if (!__builtin_amdgcn_readfirstlane(wanted)) { zskip++; continue; }
    }
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const Proj4 P = c.model ? project4_sph(c, m, ent, lane + 64 * b) : project4(c, m, ent, lane + 64 * b);
      float d[4];
#pragma unroll
      for (int k = 0; k < 4; k++) d[k] = __uint_as_float(f.dcx[((P.mask >> k) & 1u) ? (u32) (__mul24(P.row[k], c.cols) + P.col[k]) : 0u].x);
      cnt += __popc(update_mask4(c, m, P, d));
    }
  }
  if (zombies && merged_set >= 0) {  // the same for the culled list (all candidates: one lane each)
    const int ncf = t.ctr[merged_set + 2];
    for (int e = gw * 64 + lane; e < ncf; e += nw * 64) {
      const int4 ent = cfree[e];
      if (f.summary[ent.w].y & 0x80000000u) {
        u64 key;
        pack_key(mki3(ent.x, ent.y, ent.z), key);
        const int slot = hash_find(t, key);
        if (!(slot >= 0 && f.want[slot] == want_stamp)) zcul++;
      }
    }
    for (int off = 32; off > 0; off >>= 1) zcul += __shfl_xor(zcul, off);
    zskip += zcul;  // zskip (visible list) is wave-uniform, zcul was per lane
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
  if (lane == 0) {
    partials[gw] += (u64) cnt;
    if (zskip) atomicAdd((unsigned long long*) &t.prof[PROF_COMPACT], (unsigned long long) (0ull - (u64) zskip));  // modular: taken off the frame's M
    if (gw == 0)
      atomicAdd((unsigned long long*) &t.prof[PROF_COMPACT], merged_set >= 0 ? (unsigned long long) (nvis + t.ctr[merged_set + 1] + t.ctr[merged_set + 2])
                                                                             : (unsigned long long) (nvis + t.ctr[CTR_CULLED] + t.ctr[CTR_FREED_EARLY]));
  }
}

}  // namespace mrh
