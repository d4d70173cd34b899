// mrh_device.h — POD parameter blocks and device-side arithmetic for the gfx950 fusion kernels.
//
// Arithmetic spec (shared, by construction, with the CPU oracle used in tests): IEEE-754 binary32,
// no FMA contraction (-ffp-contract=off), correctly rounded / and sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt), rsqrtf restated as 1.0f / sqrtf, float->int conversions
// saturating with NaN -> 0.  Each helper cites the reference lines whose semantics it restates
// (paths relative to mrhash/src/sdf/ of rvp-group/mrhash).
//
// Kernels receive these structs BY VALUE (kernarg segment -> SGPRs); the reference instead
// dereferences device-resident copies of host C++ objects (voxel_data_structures.cuh:63,116-118).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrh_softmath.h"

namespace mrh {

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- capacities / constants (params.h:4-38) ----------------------------------------------------
constexpr int kBlockSide = 8;            // sdf_block_size
constexpr int kBlockVoxels = 512;        // total_sdf_block_size
constexpr int kCoarseVoxels = 64;
constexpr int kFineBytes = 6144;         // 512 * 12
constexpr int kCoarseBytes = 768;        // 64 * 12
constexpr u32 kMaxDdaIter = 1024;        // max_dda_iteration_count
constexpr float kFloatEps = 1e-6f;       // FLOAT_EPSILON

constexpr u64 kKeyEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr u64 kKeyTomb = 0xFFFFFFFFFFFFFFFEull;
constexpr int kKeyBias = 1 << 20;        // block coordinates in [-2^20, 2^20)
constexpr u32 kValCoarseBit = 0x80000000u;
constexpr u32 kValNone = 0xFFFFFFFFu;    // table value of a key that holds no storage (pool exhausted at insert time): lookups
                                         // treat it as absent, the next table rebuild (k_rehash_*) drops the key

// device counters (one int array, indices below)
enum Ctr : int {
  CTR_HEAP_FINE = 0,     // stack top of the fine free list  (= d_heapCounterHigh_, vds.cu:33-40)
  CTR_HEAP_COARSE = 1,   // stack top of the coarse free list (= d_heapCounterLow_)
  CTR_COMPACT = 2,       // number of compact (in-frustum) entries of this frame
  CTR_HWM_FINE = 3,      // 1 + highest fine block index ever handed out
  CTR_ERROR = 4,         // sticky error flags
  CTR_NREALLOC = 5,
  CTR_NREINT = 6,
  CTR_LIVE_FINE = 7,
  CTR_LIVE_COARSE = 8,
  CTR_TOMBS = 9,         // tombstones found by the last table census (k_table_census)
  CTR_NTRI = 10,
  CTR_DUMP = 11,
  // 12, 13: CTR_CULLED, CTR_FREED_EARLY (mrh_fast.h)
  CTR_REHASH = 14,       // decision of the last census: 1 = rebuild the table from the dense descriptors
  CTR_ORPHANS = 15,      // keys published without storage since the last rebuild (kValNone)
  CTR_NREHASH = 24,      // table rebuilds since create / reset
  CTR_HALO = 25,         // halo blocks imported from other shards (mrh_halo_import), dropped by mrh_halo_drop
  CTR_MAXPROBE = 26,     // longest probe path of a live key (filled by k_count_live for mrh_get_stats)
  CTR_TOMBS_NOW = 27,    // census accumulator
  CTR_PACK = 28,         // blocks selected by k_halo_select / k_owner_select
  CTR_ZOMBIES = 29,      // entries on the zombie list (mrh_fast2.h: lazy garbage collection of pipelined frames)
  CTR_ZSKIP = 30,        // list entries of the frame being integrated that turned out to be unwanted zombies (not blocks of that frame)
  // 32..55: the six list-counter sets of the fast path (mrh_fast2.h)
  CTR_COUNT = 56
};
// 64-bit profile counters
enum Prof : int { PROF_UPDATED = 0, PROF_INSERTED = 1, PROF_FREED = 2, PROF_COMPACT = 3, PROF_COUNT = 4 };

enum ErrBit : u32 { ERR_POOL = 1u, ERR_TABLE = 2u, ERR_RANGE = 4u, ERR_TRI = 8u, ERR_SCAN = 16u };

struct Cam {
  float fx, fy, cx, cy, ifx, ify;
  int rows, cols, row_thr, col_thr;
  float min_depth, max_depth, max_int_dist;
  float R[9], t[3];    // camera in world
  float Ri[9], ti[3];  // world in camera = (R^T, -(R^T t))   cuda_algebra.cuh:137-143
  int model;           // 0 pinhole, 1 spherical (camera.cuh:9); the two-launch fast path is pinhole-only
};

struct Map {
  float vs;           // virtual_voxel_size
  float trunc;        // sdf_truncation
  float trunc_scale;  // sdf_truncation_scale
  float var_threshold;
  float mc_threshold;
  int weight_sample;  // as u8 (vds.cu:1101)
  int weight_max;     // as u8 (vds.cu:1102)
  int min_weight_threshold;
  int shard_rank, shard_count, shard_chunk_log2;
  int block_shift_limit;  // for |voxel coordinate| < this, voxel_to_block(v) == v >> 3 (checked exhaustively at mrh_create)
  float r_half_vs;        // RN(1 / (vs / 2)), the correctly rounded reciprocal (fp64 on the host)
  int half_vs_two_steps;  // 0: div_cr by vs / 2 equals IEEE division for every dividend in the working range (checked exhaustively at mrh_create)
  int wsum_two_steps;     // 0: rcp_refined(w) IS the correctly rounded reciprocal for every weight sum w = 1 .. 510 (checked at mrh_create)
};

// Open-address table + pools.  Layout in HBM:
//   keys[slots]      u64  packed (x,y,z) | EMPTY | TOMB, linear probing
//   vals[slots]      u32  bit31 = coarse, bits0..30 = fine block index H or coarse unit u = 8H + k
//   desc_fine[cap]   int4 {x, y, z, live}   indexed by H
//   desc_coarse[8cap] int4 {x, y, z, live}  indexed by u
//   pool[cap * 6144] per fine block: sdf f32[512] | sum_squared f32[512] | rgbw u32[512]
//                    a coarse unit u lives at H*6144 + k*768 as sdf f32[64] | sum_sq f32[64] | rgbw u32[64]
struct Tab {
  u64* keys;
  u32* vals;
  u32 slot_mask;
  u32 max_probe;
  u32* heap_fine;
  u32* heap_coarse;
  int4* desc_fine;
  int4* desc_coarse;
  char* pool;
  int4* compact;  // {x, y, z, val}
  int* ctr;
  u64* prof;
  u32 cap_blocks;
  u32 multi_res;  // 1 when sdf_var_threshold > 0
  int* h_levels;  // pinned host memory {fine free-list level, zombies} written by the integration launch of a pipelining context (else nullptr)
};

// ---- scalar helpers -----------------------------------------------------------------------------

// float -> int: truncate toward zero, saturate, NaN -> 0 (what v_cvt_i32_f32 does; spelled out so the
// compiler cannot exploit out-of-range UB).
__device__ __forceinline__ int f2i(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int) v;
}
// float -> int with the same semantics in ONE instruction: v_cvt_i32_f32 truncates, saturates and maps NaN to 0
// on gfx950; inline asm so the compiler cannot treat out-of-range inputs as undefined.  Bit-identical to f2i.
__device__ __forceinline__ int f2i_hw(float v) {
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// Correctly rounded fp32 division for operands that need no exponent pre-scaling.  hipcc expands `a / b`
// (with -fhip-fp32-correctly-rounded-divide-sqrt) into v_div_scale x2, v_rcp, one Newton step on the
// reciprocal, q = a*r, two residual corrections (the second one as v_div_fmas) and v_div_fixup.  v_div_scale
// only rescales when an exponent is extreme / denormal and v_div_fixup only patches inf/nan/zero operands, so
// for ordinary operands the value is exactly the chain below.  rcp_refined(b) can be shared by every division
// with the same denominator (both pixel coordinates divide by pc.z; delta and delta2 divide by vs/2).
// mrh_selftest_division() checks bit equality against `a / b` on the device.
__device__ __forceinline__ float rcp_refined(float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float e = fmaf(-b, r, 1.0f);
  return fmaf(e, r, r);
}
__device__ __forceinline__ float div_rr(float a, float b, float r) {
  float q = a * r;
  float e = fmaf(-b, q, a);
  q = fmaf(e, r, q);
  e = fmaf(-b, q, a);
  return fmaf(e, r, q);
}

// ---- two-wide fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: full rate on CDNA3/4, each element rounded exactly
// like its scalar instruction, so pairing two voxels halves the instruction count without changing a bit) -------
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2f splat2(float a) { v2f r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f rcp_refined2(v2f b) {
  const v2f r = mk2(__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y));
  const v2f e = fma2(-b, r, splat2(1.0f));
  return fma2(e, r, r);
}
// Division with a CORRECTLY ROUNDED reciprocal r = RN(1 / b): q = RN(a r) is then within an ulp of a / b and ONE exact
// residual correction rounds to the IEEE quotient (Markstein).  tools/micro/div_cr_exhaustive.hip checks it against the
// compiler's correctly rounded division for all 2^32 dividends of every divisor 1 .. 510 (the weight sums) and of ten
// half-voxel sizes: 0 mismatches for 2^-100 <= |a| <= 2^100 and a == +0 (outside that range an intermediate under- or
// overflows; the 5-instruction div_rr behaves the same there).  mrh_create repeats the check for the context's voxel size.
__device__ __forceinline__ v2f div_cr2(v2f a, v2f b, v2f r, const bool two_steps) {
  v2f q = a * r;
  v2f e = fma2(-b, q, a);
  q = fma2(e, r, q);
  if (two_steps) {  // wave-uniform fallback: a divisor for which the check at mrh_create found a mismatch
    e = fma2(-b, q, a);
    q = fma2(e, r, q);
  }
  return q;
}
__device__ __forceinline__ float div_cr(float a, float b, float r) {
  const float q = a * r;
  return fmaf(fmaf(-b, q, a), r, q);
}
constexpr int kRcpWeightEntries = 512;  // weight sums w = 1 .. 510 (index w; entry 0 unused)

__device__ __forceinline__ v2f div_rr2(v2f a, v2f b, v2f r) {
  v2f q = a * r;
  v2f e = fma2(-b, q, a);
  q = fma2(e, r, q);
  e = fma2(-b, q, a);
  return fma2(e, r, q);
}

// ---- wave64 reductions on the VALU (DPP), result wave-uniform in an SGPR: no LDS traffic, no NaN canonicalisation
#define MRH_DPP_STEP(OP, x, ctrl, rmask) x = OP(x, (u32) __builtin_amdgcn_update_dpp((int) (x), (int) (x), ctrl, rmask, 0xF, false))
__device__ __forceinline__ u32 umin_(u32 a, u32 b) { return a < b ? a : b; }
__device__ __forceinline__ u32 umax_(u32 a, u32 b) { return a > b ? a : b; }
__device__ __forceinline__ u32 wave_min_u32(u32 x) {
  MRH_DPP_STEP(umin_, x, 0xB1, 0xF);   // quad_perm [1,0,3,2]
  MRH_DPP_STEP(umin_, x, 0x4E, 0xF);   // quad_perm [2,3,0,1]
  MRH_DPP_STEP(umin_, x, 0x141, 0xF);  // row_half_mirror
  MRH_DPP_STEP(umin_, x, 0x140, 0xF);  // row_mirror
  MRH_DPP_STEP(umin_, x, 0x142, 0xA);  // row_bcast:15 -> rows 1, 3
  MRH_DPP_STEP(umin_, x, 0x143, 0xC);  // row_bcast:31 -> rows 2, 3
  return (u32) __builtin_amdgcn_readlane((int) x, 63);
}
__device__ __forceinline__ u32 wave_max_u32(u32 x) {
  MRH_DPP_STEP(umax_, x, 0xB1, 0xF);
  MRH_DPP_STEP(umax_, x, 0x4E, 0xF);
  MRH_DPP_STEP(umax_, x, 0x141, 0xF);
  MRH_DPP_STEP(umax_, x, 0x140, 0xF);
  MRH_DPP_STEP(umax_, x, 0x142, 0xA);
  MRH_DPP_STEP(umax_, x, 0x143, 0xC);
  return (u32) __builtin_amdgcn_readlane((int) x, 63);
}

// cuda_math.cuh:62-64
__device__ __forceinline__ int signi(float v) { return (0.f < v) - (v < 0.f); }
// cuda_math.cuh:947-949
__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

struct f3 { float x, y, z; };
struct i3 { int x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ i3 mki3(int x, int y, int z) { i3 r; r.x = x; r.y = y; r.z = z; return r; }

// cuda_algebra.cuh:71-75, 146-148: rotation * point + translation (sum left to right, then + t)
__device__ __forceinline__ f3 se3_apply(const float* R, const float* t, f3 p) {
  f3 r;
  r.x = R[0] * p.x + R[1] * p.y + R[2] * p.z;
  r.y = R[3] * p.x + R[4] * p.y + R[5] * p.z;
  r.z = R[6] * p.x + R[7] * p.y + R[8] * p.z;
  r.x = r.x + t[0];
  r.y = r.y + t[1];
  r.z = r.z + t[2];
  return r;
}

// vhu.cuh:66-68
__device__ __forceinline__ f3 voxel_to_world(float vs, i3 v) { return mk3(v.x * vs, v.y * vs, v.z * vs); }

// vhu.cuh:143-151 worldPointToVirtualVoxelPos
__device__ __forceinline__ i3 world_to_voxel(float vs, f3 pt) {
  const f3 p = mk3(pt.x / vs, pt.y / vs, pt.z / vs);
  const float epsilon = 1e-5;
  f3 a = mk3(p.x + (float) signi(p.x) * 0.5f, p.y + (float) signi(p.y) * 0.5f, p.z + (float) signi(p.z) * 0.5f);
  a.x = (a.x >= 0) ? floorf(a.x + epsilon) : ceilf(a.x - epsilon);
  a.y = (a.y >= 0) ? floorf(a.y + epsilon) : ceilf(a.y - epsilon);
  a.z = (a.z >= 0) ? floorf(a.z + epsilon) : ceilf(a.z - epsilon);
  return mki3(f2i(a.x), f2i(a.y), f2i(a.z));
}

// vhu.cuh:75-103 virtualVoxelPosToSDFBlock (through float world coordinates, absolute 1e-5 epsilon)
__device__ __forceinline__ i3 voxel_to_block(i3 v, float vs) {
  const float epsilon = 1e-5;
  if (v.x < 0) v.x -= (kBlockSide - 1);
  if (v.y < 0) v.y -= (kBlockSide - 1);
  if (v.z < 0) v.z -= (kBlockSide - 1);
  const f3 pw = voxel_to_world(vs, v);
  const float mbs = (1.0f * (float) kBlockSide) * vs;  // voxel_extents == 1 (only coherent value)
  i3 b;
  b.x = f2i((pw.x >= 0) ? floorf((pw.x + epsilon) / mbs) : ceilf((pw.x - epsilon) / mbs));
  b.y = f2i((pw.y >= 0) ? floorf((pw.y + epsilon) / mbs) : ceilf((pw.y - epsilon) / mbs));
  b.z = f2i((pw.z >= 0) ? floorf((pw.z + epsilon) / mbs) : ceilf((pw.z - epsilon) / mbs));
  return b;
}
__device__ __forceinline__ i3 world_to_block(float vs, f3 pt) { return voxel_to_block(world_to_voxel(vs, pt), vs); }

// The same two conversions with the divisions by the (kernel-uniform) voxel size / block size done through one
// shared refined reciprocal each (div_rr is bit-identical to the IEEE divide for these operands; a -0 quotient may
// come out as +0, which both sign(.) and the >= 0 tests below treat identically).
struct GridRcp {
  float vs, r_vs, mbs, r_mbs;
};
__device__ __forceinline__ GridRcp make_grid_rcp(float vs) {
  GridRcp g;
  g.vs = vs;
  g.r_vs = rcp_refined(vs);
  g.mbs = (1.0f * (float) kBlockSide) * vs;
  g.r_mbs = rcp_refined(g.mbs);
  return g;
}
__device__ __forceinline__ i3 world_to_block_r(const GridRcp& g, f3 pt) {
  const float epsilon = 1e-5;
  const f3 p = mk3(div_rr(pt.x, g.vs, g.r_vs), div_rr(pt.y, g.vs, g.r_vs), div_rr(pt.z, g.vs, g.r_vs));
  f3 a = mk3(p.x + (float) signi(p.x) * 0.5f, p.y + (float) signi(p.y) * 0.5f, p.z + (float) signi(p.z) * 0.5f);
  a.x = (a.x >= 0) ? floorf(a.x + epsilon) : ceilf(a.x - epsilon);
  a.y = (a.y >= 0) ? floorf(a.y + epsilon) : ceilf(a.y - epsilon);
  a.z = (a.z >= 0) ? floorf(a.z + epsilon) : ceilf(a.z - epsilon);
  i3 v = mki3(f2i_hw(a.x), f2i_hw(a.y), f2i_hw(a.z));
  if (v.x < 0) v.x -= (kBlockSide - 1);
  if (v.y < 0) v.y -= (kBlockSide - 1);
  if (v.z < 0) v.z -= (kBlockSide - 1);
  const f3 pw = voxel_to_world(g.vs, v);
  i3 b;
  b.x = f2i_hw((pw.x >= 0) ? floorf(div_rr(pw.x + epsilon, g.mbs, g.r_mbs)) : ceilf(div_rr(pw.x - epsilon, g.mbs, g.r_mbs)));
  b.y = f2i_hw((pw.y >= 0) ? floorf(div_rr(pw.y + epsilon, g.mbs, g.r_mbs)) : ceilf(div_rr(pw.y - epsilon, g.mbs, g.r_mbs)));
  b.z = f2i_hw((pw.z >= 0) ? floorf(div_rr(pw.z + epsilon, g.mbs, g.r_mbs)) : ceilf(div_rr(pw.z - epsilon, g.mbs, g.r_mbs)));
  return b;
}

// world_to_block_r with the voxel -> block half as an arithmetic shift wherever that is KNOWN to equal the reference's
// float detour (vhu.cuh:75-103: voxel -> world -> +-1e-5 -> / (8 * vs) -> floor / ceil): mrh_create evaluates the
// float form for every voxel coordinate and records the first |v| at which it leaves floor(v / 8)
// (k_block_shift_limit); beyond that bound (hundreds of metres from the origin) the float form is used.
__device__ __forceinline__ i3 world_to_block_fast(const GridRcp& g, f3 pt, const int limit) {
  const float epsilon = 1e-5;
  const f3 p = mk3(div_rr(pt.x, g.vs, g.r_vs), div_rr(pt.y, g.vs, g.r_vs), div_rr(pt.z, g.vs, g.r_vs));
  f3 a = mk3(p.x + (float) signi(p.x) * 0.5f, p.y + (float) signi(p.y) * 0.5f, p.z + (float) signi(p.z) * 0.5f);
  a.x = (a.x >= 0) ? floorf(a.x + epsilon) : ceilf(a.x - epsilon);
  a.y = (a.y >= 0) ? floorf(a.y + epsilon) : ceilf(a.y - epsilon);
  a.z = (a.z >= 0) ? floorf(a.z + epsilon) : ceilf(a.z - epsilon);
  const i3 v = mki3(f2i_hw(a.x), f2i_hw(a.y), f2i_hw(a.z));
  const int ax = v.x < 0 ? -v.x : v.x, ay = v.y < 0 ? -v.y : v.y, az = v.z < 0 ? -v.z : v.z;
  if ((u32) (ax | ay | az) < (u32) limit) return mki3(v.x >> 3, v.y >> 3, v.z >> 3);  // (a | b | c) < 2^k bound, limit is a power of two
  return voxel_to_block(v, g.vs);
}

// vhu.cuh:184-187
__device__ __forceinline__ float get_truncation(float z, float trunc, float scale) { return trunc + scale * z; }

// local index of a voxel inside its block, dense in the block's own side (fine: 8, coarse: 4);
// see DESIGN.md "coarse reads" for why this is not vhu.cuh:110-128's stride-8 form.
__device__ __forceinline__ u32 voxel_local_index(i3 v, int res) {
  int lx = v.x % kBlockSide, ly = v.y % kBlockSide, lz = v.z % kBlockSide;
  if (lx < 0) lx += kBlockSide;
  if (ly < 0) ly += kBlockSide;
  if (lz < 0) lz += kBlockSide;
  const int side = kBlockSide >> res;
  lx >>= res; ly >>= res; lz >>= res;
  return (u32) (lz * side * side + ly * side + lx);
}

// ---- camera (pinhole): camera.cuh:84-203 ----------------------------------------------------------

// camera.cuh:88 (pinhole: what the two-launch fast path is specialised for)
__device__ __forceinline__ f3 inverse_projection(const Cam& c, u32 row, u32 col, float d) {
  return mk3(d * (c.ifx * ((float) col - c.cx - 0.5f)), d * (c.ify * ((float) row - c.cy - 0.5f)), d * 1.f);
}
// camera.cuh:84-103, either model (the general kernels).  Spherical: sinf / cosf through the one implementation shared
// with the oracle (mrh_softmath.h, D8)
__device__ __forceinline__ f3 inverse_projection_m(const Cam& c, u32 row, u32 col, float d) {
  if (c.model == 0) return inverse_projection(c, row, col, d);
  const float az = c.ifx * ((float) col - c.cx - 0.5f);
  const float el = c.ify * ((float) row - c.cy - 0.5f);
  float s0, c0, s1, c1;
  mrh_sincosf(az, &s0, &c0);
  mrh_sincosf(el, &s1, &c1);
  return mk3(d * (c0 * c1), d * (s0 * c1), d * s1);
}
// camera.cuh:120-129 getDepth
__device__ __forceinline__ float get_depth(const Cam& c, f3 p) {
  if (c.model == 0) return p.z;
  return sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
}

// camera.cuh:131-147 (exact image bounds) / :167-182 (bounds enlarged by half the image, APPROX)
template <bool APPROX>
__device__ __forceinline__ bool project_point(const Cam& c, f3 pc, int& row, int& col) {  // pinhole
  if (pc.z <= c.min_depth || pc.z > c.max_depth) return false;
  row = f2i((c.fy * pc.y / pc.z + c.cy) + 0.5f);
  col = f2i((c.fx * pc.x / pc.z + c.cx) + 0.5f);
  if (APPROX)
    return row >= -c.row_thr && col >= -c.col_thr && row < (c.rows + c.row_thr) && col < (c.cols + c.col_thr);
  return row >= 0 && col >= 0 && row < c.rows && col < c.cols;
}
template <bool APPROX>
__device__ __forceinline__ bool project_point_m(const Cam& c, f3 pc, int& row, int& col) {  // either model
  if (c.model == 0) {
    return project_point<APPROX>(c, pc, row, col);
  } else {  // camera.cuh:147-164 / :184-201
    const float range = sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z);
    if (range < c.min_depth || range > c.max_depth) return false;
    const float px = mrh_atan2f(pc.y, pc.x);
    const float py = mrh_asinf(pc.z / range);
    row = f2i((c.fy * py + c.cy) + 0.5f);
    col = f2i((c.fx * px + c.cx) + 0.5f);
  }
  if (APPROX)
    return row >= -c.row_thr && col >= -c.col_thr && row < (c.rows + c.row_thr) && col < (c.cols + c.col_thr);
  return row >= 0 && col >= 0 && row < c.rows && col < c.cols;
}

// vds.cu:66-77 isSDFBlockInCameraFrustumApprox: any of the 8 corner voxels (offsets 0 / 7, params.h:41-49)
__device__ __forceinline__ bool block_in_frustum_approx(const Cam& c, float vs, i3 b) {  // pinhole
#pragma unroll 1
  for (int i = 0; i < 8; i++) {
    // params.h:41-49 order: z toggles fastest, then y, then x
    const i3 v = mki3(b.x * kBlockSide + ((i & 4) ? 7 : 0), b.y * kBlockSide + ((i & 2) ? 7 : 0), b.z * kBlockSide + ((i & 1) ? 7 : 0));
    const f3 pc = se3_apply(c.Ri, c.ti, voxel_to_world(vs, v));
    int r, cc;
    if (project_point<true>(c, pc, r, cc)) return true;
  }
  return false;
}
__device__ __forceinline__ bool block_in_frustum_approx_m(const Cam& c, float vs, i3 b) {  // either model
#pragma unroll 1
  for (int i = 0; i < 8; i++) {
    const i3 v = mki3(b.x * kBlockSide + ((i & 4) ? 7 : 0), b.y * kBlockSide + ((i & 2) ? 7 : 0), b.z * kBlockSide + ((i & 1) ? 7 : 0));
    const f3 pc = se3_apply(c.Ri, c.ti, voxel_to_world(vs, v));
    int r, cc;
    if (project_point_m<true>(c, pc, r, cc)) return true;
  }
  return false;
}

// ---- packed block key ------------------------------------------------------------------------------

__device__ __forceinline__ bool pack_key(i3 b, u64& k) {
  const u32 x = (u32) (b.x + kKeyBias), y = (u32) (b.y + kKeyBias), z = (u32) (b.z + kKeyBias);
  if ((x | y | z) >> 21) return false;
  k = ((u64) x << 42) | ((u64) y << 21) | (u64) z;  // order of keys == (x, y, z) lexicographic order
  return true;
}
__device__ __forceinline__ i3 unpack_key(u64 k) {
  return mki3((int) ((k >> 42) & 0x1FFFFF) - kKeyBias, (int) ((k >> 21) & 0x1FFFFF) - kKeyBias, (int) (k & 0x1FFFFF) - kKeyBias);
}
__device__ __forceinline__ u32 hash_key(u64 k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (u32) k;
}

// multi-GPU tile ownership: cubes of 2^shard_chunk_log2 blocks, hashed (DESIGN.md "sharding");
// mirrored on the host by mrhash_amd/parallel.py:owner_of_blocks
__device__ __forceinline__ bool owns_block(const Map& m, i3 b) {
  if (m.shard_count <= 1) return true;
  const int sh = m.shard_chunk_log2;
  const u32 cx = (u32) (b.x >> sh), cy = (u32) (b.y >> sh), cz = (u32) (b.z >> sh);
  const u32 h = (cx * 73856093u) ^ (cy * 19349669u) ^ (cz * 83492791u);
  return (int) ((h ^ (h >> 15)) % (u32) m.shard_count) == m.shard_rank;
}

// ---- voxel pool addressing -----------------------------------------------------------------------

struct VoxPtr {
  float* sdf;
  float* sumsq;
  u32* rgbw;
};
__device__ __forceinline__ VoxPtr vox_ptr(const Tab& t, u32 val) {
  VoxPtr p;
  if (val & kValCoarseBit) {
    const u32 u = val & ~kValCoarseBit;
    char* base = t.pool + (size_t) (u >> 3) * kFineBytes + (size_t) (u & 7) * kCoarseBytes;
    p.sdf = (float*) base; p.sumsq = (float*) (base + 256); p.rgbw = (u32*) (base + 512);
  } else {
    char* base = t.pool + (size_t) val * kFineBytes;
    p.sdf = (float*) base; p.sumsq = (float*) (base + 2048); p.rgbw = (u32*) (base + 4096);
  }
  return p;
}

// ---- hash table --------------------------------------------------------------------------------

// lookup: slot index or -1.  vds.cu:80-127 getHashEntry (semantics: find the entry of a block position)
__device__ __forceinline__ int hash_find(const Tab& t, u64 key) {
  u32 s = hash_key(key) & t.slot_mask;
  for (u32 i = 0; i < t.max_probe; i++) {
    const u64 k = t.keys[s];
    if (k == key) return (int) s;
    if (k == kKeyEmpty) return -1;
    s = (s + 1) & t.slot_mask;
  }
  return -1;
}

// lookup that also remembers where an insert of this key would land: `claim` = first TOMB on the probe path, else the
// EMPTY slot that ended it (-1: path exhausted), `claim_val` = what that slot held.  Feeds hash_insert_at.
__device__ __forceinline__ int hash_find_claim(const Tab& t, u64 key, int& claim, u64& claim_val) {
  u32 s = hash_key(key) & t.slot_mask;
  claim = -1;
  claim_val = kKeyTomb;
  for (u32 i = 0; i < t.max_probe; i++) {
    const u64 k = t.keys[s];
    if (k == key) return (int) s;
    if (k == kKeyEmpty) {
      if (claim < 0) { claim = (int) s; claim_val = kKeyEmpty; }
      return -1;
    }
    if (k == kKeyTomb && claim < 0) claim = (int) s;
    s = (s + 1) & t.slot_mask;
  }
  claim = -1;  // no EMPTY within max_probe: absence is not established, use the general protocol
  return -1;
}

// concurrent insert.  Returns slot >= 0 if THIS thread inserted the key (it must then publish vals[slot]),
// -1 if the key is already present (possibly inserted concurrently by another thread), -2 on overflow.
// Lock-free: a slot only ever goes EMPTY/TOMB -> key during an insert kernel, every thread holding the same
// key walks the same probe sequence and CASes the first claimable slot, so exactly one of them wins
// (replaces the bucket mutex + host retry loop of vds.cu:502-624, 874-922).
__device__ __forceinline__ int hash_insert(const Tab& t, u64 key) {
  const u32 h = hash_key(key) & t.slot_mask;
  // phase A: is the key present before the first never-used slot?  (a TOMB may only be claimed once the key
  // is known to be absent further down the probe sequence; EMPTY slots have never been occupied)
  for (u32 i = 0; i < t.max_probe; i++) {
    const u64 k = __hip_atomic_load(&t.keys[(h + i) & t.slot_mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) return -1;
    if (k == kKeyEmpty) break;
  }
  // phase B: claim the first claimable slot in probe order
  for (u32 i = 0; i < t.max_probe; i++) {
    const u32 s = (h + i) & t.slot_mask;
    u64 k = __hip_atomic_load(&t.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) return -1;
    if (k == kKeyEmpty || k == kKeyTomb) {
      const u64 old = atomicCAS(&t.keys[s], k, key);
      if (old == k) return (int) s;
      if (old == key) return -1;
      // another key took it; keep walking
    }
  }
  return -2;
}

// Insert of a key that hash_find_claim just reported absent: ONE CAS on the remembered slot instead of two more walks
// of the probe path (each step of which is a dependent memory round trip on the inserting workgroup's critical path).
// Why this is hash_insert's protocol: the walk established "absent before the first EMPTY" (phase A).  Slots before
// `claim` held other keys at that time, and an occupied slot stays occupied for the rest of the launch, so neither
// this key nor a claimable slot can appear before `claim` later: `claim` is still the first candidate of phase B, for
// every thread holding this key.  If the CAS loses to another key, the general protocol continues.
__device__ __forceinline__ int hash_insert_at(const Tab& t, u64 key, int claim, u64 claim_val) {
  if (claim >= 0) {
    const u64 old = atomicCAS(&t.keys[claim], claim_val, key);
    if (old == claim_val) return claim;
    if (old == key) return -1;
  }
  return hash_insert(t, key);
}

// A key was published but the pool had no block for it (the reference prints and skips, vds.cu:566-569).  The slot may
// not go back to TOMB inside the insert launch — an occupied slot has to stay occupied until the launch ends, or a second
// inserter of ANOTHER key that already walked past it and a third one that now claims it could both win (a duplicate
// entry).  So the key stays, marked as holding no storage: every lookup treats kValNone as absent, and the next table
// rebuild (k_rehash_*, triggered by CTR_ORPHANS) drops it, after which the block can be allocated again.
__device__ __forceinline__ void publish_without_storage(const Tab& t, const int slot, const int heap_ctr) {
  atomicAdd(&t.ctr[heap_ctr], 1);  // undo the pop
  t.vals[slot] = kValNone;
  atomicAdd(&t.ctr[CTR_ORPHANS], 1);
  atomicOr((u32*) &t.ctr[CTR_ERROR], ERR_POOL);
}

}  // namespace mrh
