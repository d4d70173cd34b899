/*
 * mrh_oracle.c — CPU restatement of the rvp-group/mrhash fusion path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file.  The product library (libmrhash_hip.so) never links or loads it.
 *
 * What it is: a sequential, deterministic C restatement of the reference's CUDA kernels and
 * host control flow for   setDepthImage -> compute() -> extractMesh()   — same data structures
 * (24-byte HashEntry, bucket of 10 + overflow list of 7, two free heaps, 12-byte Voxel), same
 * float operations in the same order.  Every function cites the reference lines it follows
 * (paths relative to mrhash/src/sdf/ in the reference checkout).  It implements the C ABI of
 * include/mrhash_hip.h so tests can drive it and the HIP library through identical code.
 *
 * Arithmetic spec (shared with the HIP kernels): IEEE-754 binary32, round-to-nearest-even,
 * NO fused multiply-add contraction (-ffp-contract=off), correctly rounded / and sqrt,
 * `rsqrtf(x)` restated as `1.0f / sqrtf(x)`, float->int conversions saturating with NaN -> 0.
 *
 * Parity pinning: the reference as a whole cannot be built or run in this environment (no nvcc, no
 * CUDA device, Eigen/nanobind/OpenCV absent) and its tests hold no golden values for integration,
 * variance adaptation, GC or marching cubes.  The oracle is pinned against (a) the slice of the
 * reference that DOES build from its own sources (oracle/_ref, Makefile target `ref`:
 * voxel_hash_utils.cuh's host-visible structs, constants and index helpers by g++, params.h's
 * marching-cubes tables by hipcc; tests/test_oracle_pinning.py, tests/test_parity_gpu.py), (b) the
 * invariants the reference's own tests assert (tests/test_hash_utils.cu, test_projections.cu,
 * test_marching_cubes.cpp), (c) a second, independent restatement of every step
 * (tests/independent.py) and analytic known answers.  For integrate / variance / GC /
 * marching-cubes RESULTS the status remains "PARITY UNPINNED against reference GPU output".
 *
 * Canonicalisation of race-ordered reference behaviour (documented deviations):
 *   C1  compact-list order = block position ascending in (x,y,z)   (reference: atomicAdd winners,
 *       vds.cu:419-426).  Affects only starve tie-breaks and triangle order.
 *   D1  reads of coarse (4^3) blocks use the dense 0..63 index the writers use; the reference
 *       linearises the coarse local position with stride 8 (vhu.cuh:110-128) and so reads
 *       memory of sibling blocks — a heap-layout-dependent result with no canonical value.
 *   D2  kept literally: reintegrateDepthMap touches only voxel indices 0..31 (launch-shape
 *       quirk, vds.cu:2097) and sum_squared holds only the latest delta*delta2 term
 *       (vds.cu:1173-1180).
 *   D3  starve on coarse blocks: only threads 0..63 take part (the reference runs 512 threads
 *       that index past the 64-voxel block, vds.cu:1597-1649).
 *   D5  the reintegrate list is cleared every frame (the reference replays a stale list when a
 *       frame coarsens nothing, vds.cu:2040-2043 + :2091-2093, which can write out of bounds).
 *   D4  vertex merge with epsilon == 0 compares the three doubles bitwise (the reference
 *       hashes the bytes and compares with ==; they differ only for -0.0 / NaN).
 *   D6  LiDAR scans (integrate3DKernel, vds.cu:1215-1379): the reference updates a voxel with a
 *       non-atomic read-modify-write per point, so the points of one scan that cross the same voxel
 *       race.  Canonical: every voxel receives its updates in ascending point index (the sequential
 *       loop of mrh_integrate_points).  norm3df(x, y, z) is restated as sqrtf((x*x + y*y) + z*z).
 *   D8  spherical camera model (camera.cuh:93-101, :147-164, :184-201): sinf / cosf / atan2f / asinf are evaluated by ONE
 *       plain-fp32 implementation shared with the kernels (include/mrh_softmath.h, within ~2 ulp of the correctly
 *       rounded values) instead of CUDA's intrinsics, exactly as rsqrtf is restated as 1 / sqrtf.
 *   D7  3DGS splat seeds (subdivideKernel quad_tree.cu:102-167, processNodesKernel
 *       gaussian_data_structures.cu:5-56): the reference appends leaves, child nodes and seeds through
 *       atomic counters.  Canonical: leaves by tree level, inside a level in tree order (children in the
 *       order the reference writes them); seeds in leaf order.  Node errors keep the reference's
 *       summation order exactly (256 strided partial sums, then the halving tree of computeError).
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mrhash_hip.h"
#include "../include/mrh_mc_tables.h"
#include "../include/mrh_softmath.h"

/* ---- constants: params.h:4-38 ---------------------------------------------------------- */
#define LOCK_ENTRY (-1)
#define FREE_ENTRY (-2)
#define NO_OFFSET 0
#define P0 73856093u
#define P1 19349669u
#define P2 83492791u
#define SDF_BLOCK_SIZE 8
#define TOTAL_SDF_BLOCK_SIZE 512
#define FINEST_BLOCK_LOG2_DIM 3
#define OCTREE_BRANCHING_FACTOR 8
#define LINKED_LIST_SIZE 7u
#define HASH_BUCKET_SIZE 10u
#define MAX_DDA_ITERATION_COUNT 1024u
#define FLOAT_EPSILON 1e-6f

static const uint8_t MC_TRI[256][16] = MRH_MC_TRI_TABLE_INIT;

/* params.h:40-49 vert_offset */
static const int VERT_OFFSET[8][3] = {{0, 0, 0}, {0, 0, 7}, {0, 7, 0}, {0, 7, 7}, {7, 0, 0}, {7, 0, 7}, {7, 7, 0}, {7, 7, 7}};

typedef struct { int x, y, z; } i3;
typedef struct { float x, y, z; } f3;

/* vhu.cuh:27-38 */
typedef struct {
  i3 pos;
  unsigned offset;
  int ptr;
  int resolution;
} HashEntry;

typedef mrh_voxel Voxel; /* vhu.cuh:8-22, 12 bytes */

struct mrh_ctx {
  mrh_params p;
  /* camera: camera.cuh:13-40 */
  int has_camera;
  float fx, fy, ifx, ify, cx, cy;
  unsigned rows, cols;
  int row_threshold, col_threshold;
  float min_depth, max_depth;
  int model;
  float max_integration_distance; /* geowrapper.cpp:111 */
  /* pose (cam in world) and its inverse, cuda_algebra.cuh:137-143 */
  float R[9], t[3], Ri[9], ti[3];
  /* images */
  float* depth;
  uint8_t* rgb;
  int depth_rows, depth_cols, rgb_rows, rgb_cols;
  float* points;       /* sensor-frame xyz of the current scan (GeoWrapper::setPointCloud, geowrapper.cpp:345-405) */
  uint64_t num_points;
  float* normals;      /* one normal per point (mrh_upload_normals), NULL: none */
  uint64_t num_normals;
  f3* cloud;
  /* container: voxel_data_structures.cuh:63-100 */
  unsigned num_sdf_blocks, hash_num_buckets, total_size, low_blocks_to_allocate;
  HashEntry* table;
  HashEntry* compact;
  int* decision;
  int* mutex;
  unsigned* heap_high;
  unsigned* heap_low;
  int heap_counter_high, heap_counter_low;
  Voxel* blocks;
  unsigned current_occupied;
  i3* realloc_pos;
  int* realloc_res;
  unsigned num_realloc;
  unsigned* reintegrate;
  unsigned num_reintegrate;
  uint64_t* depth_buff;
  size_t depth_buff_n;
  uint64_t frames;
  int pending; /* 0 none, 1 after starve pass 0, 2 after pass 1 (sharded contexts) */
  /* mesh */
  mrh_triangle* tris;
  uint64_t ntris, cap_tris, max_triangles;
  int merge_on;            /* MeshExtractor::merge_mesh_ (mrh_mesh_merge_begin / _end) */
  uint64_t merge_total;    /* triangles of all extractions since begin */
  mrh_block_desc* tri_blocks;
  uint32_t* tri_counts;
  uint64_t n_tri_blocks;
  double* V;
  double* C;
  int32_t* F;
  uint64_t nv, nf;
  /* 3DGS splat seeds */
  mrh_qtree_leaf* qt_leaves;
  uint64_t n_qt_leaves;
  mrh_splat_seed* seeds;
  uint64_t n_seeds;
  /* stats */
  int profile;
  uint64_t last_updated, last_inserted, last_freed, total_updated, total_compact;
  uint32_t error_flags;
  /* multi-GPU block exchange (test counterpart of the HIP library's device buffers) */
  mrh_block_record* pack;
  uint64_t pack_cap;
  mrh_block_desc* halo;
  uint64_t n_halo, halo_cap;
  char err[256];
};

static char g_create_err[256] = "";

static int fail(mrh_ctx* c, int code, const char* msg) {
  char* dst = c ? c->err : g_create_err;
  snprintf(dst, 256, "%s", msg);
  return code;
}

/* ---- scalar helpers --------------------------------------------------------------------- */

/* float -> int as the arithmetic spec defines it (truncate toward zero, saturate, NaN -> 0);
 * identical to v_cvt_i32_f32 on gfx950 and to C for every in-range value. */
static inline int f2i(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return INT_MAX;
  if (v <= -2147483648.0f) return INT_MIN;
  return (int) v;
}
/* cuda_math.cuh:62-64 */
static inline int signi(float v) { return (0.f < v) - (v < 0.f); }
/* cuda_math.cuh:947-949 */
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }

/* cuda_algebra.cuh:71-75 + :146-148: rotation * point + translation */
static inline f3 se3_apply(const float* R, const float* t, f3 p) {
  f3 r;
  r.x = R[0] * p.x + R[1] * p.y + R[2] * p.z;
  r.y = R[3] * p.x + R[4] * p.y + R[5] * p.z;
  r.z = R[6] * p.x + R[7] * p.y + R[8] * p.z;
  r.x = r.x + t[0];
  r.y = r.y + t[1];
  r.z = r.z + t[2];
  return r;
}

/* cuda_algebra.cuh:45-57, 137-143: inverse = (R^T, -(R^T t)) */
static void se3_inverse(const float* R, const float* t, float* Ri, float* ti) {
  Ri[0] = R[0]; Ri[1] = R[3]; Ri[2] = R[6];
  Ri[3] = R[1]; Ri[4] = R[4]; Ri[5] = R[7];
  Ri[6] = R[2]; Ri[7] = R[5]; Ri[8] = R[8];
  float x = Ri[0] * t[0] + Ri[1] * t[1] + Ri[2] * t[2];
  float y = Ri[3] * t[0] + Ri[4] * t[1] + Ri[5] * t[2];
  float z = Ri[6] * t[0] + Ri[7] * t[1] + Ri[8] * t[2];
  ti[0] = -x; ti[1] = -y; ti[2] = -z;
}

/* ---- coordinates: vhu.cuh:66-165 -------------------------------------------------------- */

/* vhu.cuh:66-68 */
static inline f3 voxel_to_world(float vs, i3 v) { return mk3(v.x * vs, v.y * vs, v.z * vs); }

/* vhu.cuh:143-151 worldPointToVirtualVoxelPos */
static inline i3 world_to_voxel(float vs, f3 pt) {
  f3 p = mk3(pt.x / vs, pt.y / vs, pt.z / vs);
  const float epsilon = 1e-5;
  f3 a = mk3(p.x + (float) signi(p.x) * 0.5f, p.y + (float) signi(p.y) * 0.5f, p.z + (float) signi(p.z) * 0.5f);
  a.x = (a.x >= 0) ? floorf(a.x + epsilon) : ceilf(a.x - epsilon);
  a.y = (a.y >= 0) ? floorf(a.y + epsilon) : ceilf(a.y - epsilon);
  a.z = (a.z >= 0) ? floorf(a.z + epsilon) : ceilf(a.z - epsilon);
  i3 r = {f2i(a.x), f2i(a.y), f2i(a.z)};
  return r;
}

/* vhu.cuh:75-103 virtualVoxelPosToSDFBlock (goes through float world coordinates) */
static inline i3 voxel_to_block(i3 v, float vs, float extents) {
  const float epsilon = 1e-5;
  if (v.x < 0) v.x -= (SDF_BLOCK_SIZE - 1);
  if (v.y < 0) v.y -= (SDF_BLOCK_SIZE - 1);
  if (v.z < 0) v.z -= (SDF_BLOCK_SIZE - 1);
  const f3 pw = voxel_to_world(vs, v);
  const float mbs = (extents * (float) SDF_BLOCK_SIZE) * vs;
  i3 b;
  b.x = f2i((pw.x >= 0) ? floorf((pw.x + epsilon) / mbs) : ceilf((pw.x - epsilon) / mbs));
  b.y = f2i((pw.y >= 0) ? floorf((pw.y + epsilon) / mbs) : ceilf((pw.y - epsilon) / mbs));
  b.z = f2i((pw.z >= 0) ? floorf((pw.z + epsilon) / mbs) : ceilf((pw.z - epsilon) / mbs));
  return b;
}

/* vhu.cuh:157-161 worldPointToSDFBlock */
static inline i3 world_to_block(float vs, float extents, f3 pt) { return voxel_to_block(world_to_voxel(vs, pt), vs, extents); }

/* vhu.cuh:138-140 */
static inline i3 block_to_voxel(i3 b) { i3 r = {b.x * SDF_BLOCK_SIZE, b.y * SDF_BLOCK_SIZE, b.z * SDF_BLOCK_SIZE}; return r; }

/* vhu.cuh:106-108 */
static inline unsigned linearize(i3 p, int bs) { return (unsigned) (p.z * bs * bs + p.y * bs + p.x); }

/* vhu.cuh:130-136 */
static inline i3 delinearize(unsigned idx, int bs) {
  const unsigned size2 = (unsigned) (bs * bs);
  i3 r = {(int) (idx % (unsigned) bs), (int) ((idx % size2) / (unsigned) bs), (int) (idx / size2)};
  return r;
}

/* vhu.cuh:110-128 virtualVoxelPosToSDFBlockIndex, literal (stride = sdf_block_size). */
static inline unsigned voxel_to_block_index_ref(i3 v, int block_size) {
  const int scaling = SDF_BLOCK_SIZE / block_size;
  i3 l = {v.x % SDF_BLOCK_SIZE, v.y % SDF_BLOCK_SIZE, v.z % SDF_BLOCK_SIZE};
  if (l.x < 0) l.x += SDF_BLOCK_SIZE;
  if (l.y < 0) l.y += SDF_BLOCK_SIZE;
  if (l.z < 0) l.z += SDF_BLOCK_SIZE;
  l.x /= scaling; l.y /= scaling; l.z /= scaling;
  return linearize(l, SDF_BLOCK_SIZE);
}

/* Deviation D1: same local position, linearised with the block's own side so that a coarse
 * block is read where the integrate kernel wrote it (vds.cu:1114-1118,1162). Equal to the
 * literal form for fine blocks. */
static inline unsigned voxel_to_block_index(i3 v, int block_size) {
  const int scaling = SDF_BLOCK_SIZE / block_size;
  i3 l = {v.x % SDF_BLOCK_SIZE, v.y % SDF_BLOCK_SIZE, v.z % SDF_BLOCK_SIZE};
  if (l.x < 0) l.x += SDF_BLOCK_SIZE;
  if (l.y < 0) l.y += SDF_BLOCK_SIZE;
  if (l.z < 0) l.z += SDF_BLOCK_SIZE;
  l.x /= scaling; l.y /= scaling; l.z /= scaling;
  return linearize(l, block_size);
}

/* exported for the known-answer test: the literal reference forms */
unsigned orc_kat_voxel_to_block_index(int x, int y, int z, int block_size) {
  i3 v = {x, y, z};
  return voxel_to_block_index_ref(v, block_size);
}
void orc_kat_delinearize(unsigned idx, int block_size, int out[3]) {
  i3 r = delinearize(idx, block_size);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_kat_world_to_voxel(float vs, const float p[3], int out[3]) {
  i3 r = world_to_voxel(vs, mk3(p[0], p[1], p[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_kat_voxel_to_block(const int v[3], float vs, int out[3]) {
  i3 vv = {v[0], v[1], v[2]};
  i3 r = voxel_to_block(vv, vs, 1.0f);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* vhu.cuh:184-187 */
static inline float get_truncation(float z, float trunc, float scale) { return trunc + scale * z; }

/* ---- camera: camera.cuh:84-203 ---------------------------------------------------------- */

static inline f3 inverse_projection(const mrh_ctx* c, unsigned row, unsigned col, float d) {
  if (c->model == MRH_CAMERA_PINHOLE) {
    /* camera.cuh:88 */
    return mk3(d * (c->ifx * ((float) col - c->cx - 0.5f)), d * (c->ify * ((float) row - c->cy - 0.5f)), d * 1.f);
  } else {
    /* camera.cuh:91-99 */
    const float az = c->ifx * ((float) col - c->cx - 0.5f);
    const float el = c->ify * ((float) row - c->cy - 0.5f);
    float s0, c0, s1, c1; /* D8: one shared fp32 implementation instead of CUDA's sinf / cosf (include/mrh_softmath.h) */
    mrh_sincosf(az, &s0, &c0);
    mrh_sincosf(el, &s1, &c1);
    return mk3(d * (c0 * c1), d * (s0 * c1), d * s1);
  }
}

/* camera.cuh:120-129 */
static inline float get_depth(const mrh_ctx* c, f3 p) {
  if (c->model == MRH_CAMERA_PINHOLE) return p.z;
  return sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
}

/* camera.cuh:131-165 (approx = 0) and :167-203 (approx = 1).  pimg = (row, col). */
static inline int project_point(const mrh_ctx* c, f3 pc, int approx, int* row_out, int* col_out) {
  int row, col;
  if (c->model == MRH_CAMERA_PINHOLE) {
    if (pc.z <= c->min_depth || pc.z > c->max_depth) return 0;
    row = f2i((c->fy * pc.y / pc.z + c->cy) + 0.5f);
    col = f2i((c->fx * pc.x / pc.z + c->cx) + 0.5f);
  } else {
    const float range = sqrtf(pc.x * pc.x + pc.y * pc.y + pc.z * pc.z);
    if (range < c->min_depth || range > c->max_depth) return 0;
    const float px = mrh_atan2f(pc.y, pc.x); /* D8 */
    const float py = mrh_asinf(pc.z / range);
    row = f2i((c->fy * py + c->cy) + 0.5f);
    col = f2i((c->fx * px + c->cx) + 0.5f);
  }
  if (!approx) {
    if (row >= 0 && col >= 0 && row < (int) c->rows && col < (int) c->cols) { *row_out = row; *col_out = col; return 1; }
  } else {
    if (row >= -c->row_threshold && col >= -c->col_threshold && row < (int) (c->rows + c->row_threshold) &&
        col < (int) (c->cols + c->col_threshold)) { *row_out = row; *col_out = col; return 1; }
  }
  return 0;
}

/* camera.cuh:109-118 isInCameraFrustumApprox + vds.cu:66-77 isSDFBlockInCameraFrustumApprox */
static int block_in_frustum_approx(const mrh_ctx* c, i3 block) {
  const float vs = c->p.virtual_voxel_size;
  for (int i = 0; i < 8; i++) {
    i3 base = block_to_voxel(block);
    i3 v = {base.x + VERT_OFFSET[i][0], base.y + VERT_OFFSET[i][1], base.z + VERT_OFFSET[i][2]};
    f3 pw = voxel_to_world(vs, v);
    f3 pc = se3_apply(c->Ri, c->ti, pw);
    int r, cc;
    if (project_point(c, pc, 1, &r, &cc)) return 1;
  }
  return 0;
}

/* camera.cu:5-26 calculateCloudKernel + Camera::computeCloud */
static void compute_cloud(mrh_ctx* c) {
  const size_t n = (size_t) c->rows * c->cols;
  memset(c->cloud, 0, n * sizeof(f3));
  for (unsigned row = 0; row < c->rows; row++)
    for (unsigned col = 0; col < c->cols; col++) {
      const float d = c->depth[(size_t) row * c->cols + col];
      if (d <= c->min_depth || d > c->max_depth) continue;
      c->cloud[(size_t) row * c->cols + col] = inverse_projection(c, row, col, d);
    }
}

/* ---- hash table + heaps: vds.cu:33-62, 80-127, 151-160, 502-755, 1727-1824 --------------- */

/* vds.cu:151-160 */
static inline unsigned calc_hash(const mrh_ctx* c, i3 b) {
  unsigned x = (unsigned) b.x, y = (unsigned) b.y, z = (unsigned) b.z;
  int res = (int) (((x * P0) ^ (y * P1) ^ (z * P2)) % c->hash_num_buckets);
  if (res < 0) res += (int) c->hash_num_buckets;
  return (unsigned) res;
}

/* vds.cu:33-40 / :43-50 (atomicSub returns the old value) */
static int consume_heap_high(mrh_ctx* c) {
  int addr = c->heap_counter_high; c->heap_counter_high -= 1;
  if (addr < 0) return -1;
  return (int) c->heap_high[addr];
}
static int consume_heap_low(mrh_ctx* c) {
  int addr = c->heap_counter_low; c->heap_counter_low -= 1;
  if (addr < 0) return -1;
  return (int) c->heap_low[addr];
}
/* vds.cu:53-62 */
static void append_heap_high(mrh_ctx* c, unsigned ptr) { int addr = c->heap_counter_high; c->heap_counter_high += 1; c->heap_high[addr + 1] = ptr; }
static void append_heap_low(mrh_ctx* c, unsigned ptr) { int addr = c->heap_counter_low; c->heap_counter_low += 1; c->heap_low[addr + 1] = ptr; }

/* voxel_data_structures.cpp:148-161 */
static int heap_high_free(const mrh_ctx* c) { return (int) ((unsigned) c->heap_counter_high + 1u); }
static int heap_low_free(const mrh_ctx* c) { return c->heap_counter_low + 1; }

static inline int num_voxels_of(int resolution) { const int s = 1 << (FINEST_BLOCK_LOG2_DIM - resolution); return s * s * s; } /* vds.cu:208-212 */

static inline void delete_hash_entry(HashEntry* e) { e->pos.x = e->pos.y = e->pos.z = 0; e->offset = NO_OFFSET; e->ptr = FREE_ENTRY; } /* vhu.cuh:190-194: resolution untouched */
static inline void delete_voxel(Voxel* v) { v->sdf = 0.f; v->rgb[0] = v->rgb[1] = v->rgb[2] = 0; v->sum_squared = 0.f; v->weight = 0; } /* vhu.cuh:197-203 */

static inline int same_pos(const HashEntry* e, i3 p) { return e->pos.x == p.x && e->pos.y == p.y && e->pos.z == p.z && e->ptr != FREE_ENTRY; }

/* vds.cu:80-127 getHashEntry */
static HashEntry get_hash_entry(const mrh_ctx* c, i3 block) {
  HashEntry entry;
  entry.pos = block; entry.ptr = FREE_ENTRY; entry.offset = 0; entry.resolution = 0;
  const uint64_t h = calc_hash(c, block);
  for (unsigned i = 0; i < HASH_BUCKET_SIZE; ++i) {
    const HashEntry* curr = &c->table[h * HASH_BUCKET_SIZE + i];
    if (same_pos(curr, block)) return *curr;
  }
  const unsigned last = (unsigned) ((h + 1) * HASH_BUCKET_SIZE - 1);
  unsigned i = last;
  unsigned it = 0;
  while (it < LINKED_LIST_SIZE) {
    const HashEntry curr = c->table[i];
    if (same_pos(&curr, block)) return curr;
    if (curr.offset == 0) break;
    i = last + curr.offset;
    i %= (HASH_BUCKET_SIZE * c->hash_num_buckets);
    it++;
  }
  return entry;
}

/* atomicExch(&mutex[h], LOCK_ENTRY) */
static inline int mutex_exch(mrh_ctx* c, unsigned h) { int prev = c->mutex[h]; c->mutex[h] = LOCK_ENTRY; return prev; }

/* vds.cu:502-624 allocBlock (is_realloc = 0) and :627-755 reallocBlock (is_realloc = 1): the two
 * differ only in their return values. */
static int alloc_block(mrh_ctx* c, i3 pos, int resolution, int is_realloc) {
  unsigned h = calc_hash(c, pos);
  const unsigned hp = h * HASH_BUCKET_SIZE;
  int first_empty = -1;
  for (unsigned j = 0; j < HASH_BUCKET_SIZE; ++j) {
    const unsigned i = hp + j;
    const HashEntry* curr = &c->table[i];
    if (same_pos(curr, pos)) return -1;
    if (first_empty == -1 && curr->ptr == FREE_ENTRY) first_empty = (int) i;
  }
  const unsigned last = (h + 1) * HASH_BUCKET_SIZE - 1;
  unsigned i = last;
  HashEntry curr;
  curr.offset = 0;
  unsigned it = 0;
  while (it < LINKED_LIST_SIZE) {
    curr = c->table[i];
    if (same_pos(&curr, pos)) return -1;
    if (curr.offset == 0) break;
    i = last + curr.offset;
    i %= (HASH_BUCKET_SIZE * c->hash_num_buckets);
    it++;
  }
  if (first_empty != -1) {
    const int prev = mutex_exch(c, h);
    if (prev != LOCK_ENTRY) {
      HashEntry* e = &c->table[first_empty];
      e->pos = pos; e->offset = NO_OFFSET; e->resolution = resolution;
      int ptr_idx = -1;
      if (resolution == 0) ptr_idx = consume_heap_high(c);
      else if (resolution == 1) ptr_idx = consume_heap_low(c);
      if (ptr_idx < 0) { c->error_flags |= 1u; return -1; } /* "mem size exceed, not inserting hash entry" */
      e->ptr = ptr_idx * num_voxels_of(resolution);
      if (c->profile) c->last_inserted++;
      return first_empty;
    }
    return is_realloc ? LOCK_ENTRY : first_empty;
  }
  int offset = 0;
  it = 0;
  while (it < LINKED_LIST_SIZE) {
    offset++;
    i = (last + (unsigned) offset) % c->total_size;
    if (((unsigned) offset % HASH_BUCKET_SIZE) == 0) continue;
    curr = c->table[i];
    if (curr.ptr == FREE_ENTRY) {
      int prev = mutex_exch(c, h);
      if (prev != LOCK_ENTRY) {
        HashEntry last_entry = c->table[last];
        h = i / HASH_BUCKET_SIZE;
        prev = mutex_exch(c, h);
        if (prev != LOCK_ENTRY) {
          HashEntry* e = &c->table[i];
          e->pos = pos; e->offset = last_entry.offset; e->resolution = resolution;
          int ptr_idx = -1;
          if (resolution == 0) ptr_idx = consume_heap_high(c);
          else if (resolution == 1) ptr_idx = consume_heap_low(c);
          if (ptr_idx < 0) { c->error_flags |= 1u; return is_realloc ? LOCK_ENTRY : -1; }
          e->ptr = ptr_idx * num_voxels_of(resolution);
          if (c->profile) c->last_inserted++;
          last_entry.offset = (unsigned) offset;
          c->table[last] = last_entry;
        }
      }
      return is_realloc ? LOCK_ENTRY : -1;
    }
    it++;
  }
  return -1;
}

/* vds.cu:1727-1824 deleteHashEntryElement */
static int delete_hash_entry_element(mrh_ctx* c, i3 block) {
  const unsigned hash = calc_hash(c, block);
  const unsigned start = hash * HASH_BUCKET_SIZE;
  for (unsigned j = 0; j < HASH_BUCKET_SIZE; j++) {
    const unsigned i = start + j;
    const HashEntry curr = c->table[i];
    const int vbv = num_voxels_of(curr.resolution);
    if (same_pos(&curr, block)) {
      if (curr.offset != 0) {
        const int prev = mutex_exch(c, hash);
        if (prev == LOCK_ENTRY) return 0;
        if (curr.resolution == 0) append_heap_high(c, (unsigned) (curr.ptr / vbv));
        if (curr.resolution == 1) append_heap_low(c, (unsigned) (curr.ptr / vbv));
        const unsigned next_idx = (i + curr.offset) % c->total_size;
        c->table[i] = c->table[next_idx];
        delete_hash_entry(&c->table[next_idx]);
        return 1;
      } else {
        if (curr.resolution == 0) append_heap_high(c, (unsigned) (curr.ptr / vbv));
        if (curr.resolution == 1) append_heap_low(c, (unsigned) (curr.ptr / vbv));
        delete_hash_entry(&c->table[i]);
        return 1;
      }
    }
  }
  const unsigned last = (hash + 1) * HASH_BUCKET_SIZE - 1;
  int i = (int) last;
  HashEntry curr = c->table[i];
  int prev_idx = i;
  i = (int) ((last + curr.offset) % c->total_size);
  unsigned it = 0;
  while (it < LINKED_LIST_SIZE) {
    curr = c->table[i];
    const int vbv = num_voxels_of(curr.resolution);
    if (same_pos(&curr, block)) {
      const int prev = mutex_exch(c, hash);
      if (prev == LOCK_ENTRY) return 0;
      if (curr.resolution == 0) append_heap_high(c, (unsigned) (curr.ptr / vbv));
      if (curr.resolution == 1) append_heap_low(c, (unsigned) (curr.ptr / vbv));
      delete_hash_entry(&c->table[i]);
      HashEntry pe = c->table[prev_idx];
      pe.offset = curr.offset;
      c->table[prev_idx] = pe;
      return 1;
    }
    if (curr.offset == 0) return 0;
    prev_idx = i;
    i = (int) ((last + curr.offset) % c->total_size);
    it++;
  }
  return 0;
}

/* vds.cu:17-30 */
static void reset_mutex(mrh_ctx* c) { for (unsigned i = 0; i < c->hash_num_buckets; i++) c->mutex[i] = FREE_ENTRY; }

/* ---- voxel access: vds.cu:163-240 -------------------------------------------------------- */

static Voxel get_voxel_i(const mrh_ctx* c, i3 vpos, int* block_res) {
  Voxel v;
  const HashEntry e = get_hash_entry(c, voxel_to_block(vpos, c->p.virtual_voxel_size, (float) c->p.voxel_extents_scale));
  if (e.ptr == FREE_ENTRY) { delete_voxel(&v); return v; }
  const int scaling = 1 << e.resolution;
  if (block_res) *block_res = e.resolution;
  return c->blocks[(size_t) e.ptr + voxel_to_block_index(vpos, SDF_BLOCK_SIZE / scaling)]; /* D1 */
}
static Voxel get_voxel_f(const mrh_ctx* c, f3 pos, int* block_res) { return get_voxel_i(c, world_to_voxel(c->p.virtual_voxel_size, pos), block_res); }

/* vds.cu:236-240 getVoxelSize(float3) */
static float get_voxel_size_f(const mrh_ctx* c, f3 pos) {
  const HashEntry e = get_hash_entry(c, world_to_block(c->p.virtual_voxel_size, (float) c->p.voxel_extents_scale, pos));
  return c->p.virtual_voxel_size * (float) (1 << e.resolution);
}

/* vds.cu:260-338 trilinearInterpolation */
static int trilinear(const mrh_ctx* c, f3 pos, float* dist) {
  const float voxel_size = get_voxel_size_f(c, pos);
  const f3 pos_dual = mk3(pos.x - voxel_size * 0.5f, pos.y - voxel_size * 0.5f, pos.z - voxel_size * 0.5f);
  const HashEntry entry = get_hash_entry(c, world_to_block(voxel_size, (float) c->p.voxel_extents_scale, pos));
  const int base_resolution = entry.resolution;
  *dist = 0.f;
  const float pos_sdf = get_voxel_f(c, pos_dual, NULL).sdf;
  const float x0 = pos_dual.x, y0 = pos_dual.y, z0 = pos_dual.z;
  float x1 = x0, y1 = y0, z1 = z0;
  int resolution = 0;
  float sdf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 8; ++i) {
    const int dx = i & 1, dy = (i >> 1) & 1, dz = (i >> 2) & 1;
    const f3 vp = mk3(pos_dual.x + (float) dx * voxel_size, pos_dual.y + (float) dy * voxel_size, pos_dual.z + (float) dz * voxel_size);
    const Voxel v = get_voxel_f(c, vp, &resolution);
    if (!v.weight) return 0;
    if (resolution > base_resolution) {
      const float nvs = voxel_size * 2;
      const f3 np = mk3(pos.x - nvs * 0.5f + (float) dx * nvs, pos.y - nvs * 0.5f + (float) dy * nvs, pos.z - nvs * 0.5f + (float) dz * nvs);
      const float np_sdf = get_voxel_f(c, np, NULL).sdf;
      const float alpha = 0.5f;
      sdf[i] = (1 - alpha) * pos_sdf + alpha * np_sdf;
    } else {
      sdf[i] = v.sdf;
    }
    if (vp.x > x1) x1 = vp.x;
    if (vp.y > y1) y1 = vp.y;
    if (vp.z > z1) z1 = vp.z;
    resolution = 0;
  }
  const float ddx = (x1 - x0) > 1e-6f ? (pos.x - x0) / (x1 - x0) : 0.5f;
  const float ddy = (y1 - y0) > 1e-6f ? (pos.y - y0) / (y1 - y0) : 0.5f;
  const float ddz = (z1 - z0) > 1e-6f ? (pos.z - z0) / (z1 - z0) : 0.5f;
  const float cc[8] = {sdf[0],
                       (sdf[1] - sdf[0]),
                       (sdf[2] - sdf[0]),
                       (sdf[4] - sdf[0]),
                       (sdf[3] - sdf[2] - sdf[1] + sdf[0]),
                       (sdf[6] - sdf[4] - sdf[2] + sdf[0]),
                       (sdf[5] - sdf[4] - sdf[1] + sdf[0]),
                       (sdf[7] - sdf[6] - sdf[5] - sdf[3] + sdf[1] + sdf[4] + sdf[2] - sdf[0])};
  *dist = cc[0] + cc[1] * ddx + cc[2] * ddy + cc[3] * ddz + cc[4] * ddx * ddy + cc[5] * ddy * ddz + cc[6] * ddx * ddz +
          cc[7] * ddx * ddy * ddz;
  return 1;
}

/* multi-GPU tile ownership (no reference counterpart; mirrors mrh_device.h owns_block so that the sharded host
 * logic can be tested on the CPU with gloo) */
static int owns_block(const mrh_ctx* c, i3 b) {
  if (c->p.shard_count <= 1) return 1;
  const int sh = (c->p.shard_chunk_log2 > 0 && c->p.shard_chunk_log2 < 16) ? c->p.shard_chunk_log2 : 3;
  const uint32_t cx = (uint32_t) (b.x >> sh), cy = (uint32_t) (b.y >> sh), cz = (uint32_t) (b.z >> sh);
  const uint32_t h = (cx * P0) ^ (cy * P1) ^ (cz * P2);
  return (int) ((h ^ (h >> 15)) % (uint32_t) c->p.shard_count) == c->p.shard_rank;
}

/* ---- allocation: vds.cu:758-922 ----------------------------------------------------------- */

/* vds.cu:758-857 allocBlocksKernel, one pixel */
static void alloc_pixel(mrh_ctx* c, unsigned row, unsigned col) {
  const float vs = c->p.virtual_voxel_size;
  const float ext = (float) c->p.voxel_extents_scale;
  const float depth = get_depth(c, c->cloud[(size_t) row * c->cols + col]);
  if (depth == 0.f) return;
  const float t = get_truncation(depth, c->p.sdf_truncation, c->p.sdf_truncation_scale);
  const float min_depth = fminf(c->max_integration_distance, depth - t);
  const float max_depth = fminf(c->max_integration_distance, depth + t);
  if (min_depth >= max_depth) return;
  const f3 pcam_min = inverse_projection(c, row, col, min_depth);
  const f3 pcam_max = inverse_projection(c, row, col, max_depth);
  const f3 pw_min = se3_apply(c->R, c->t, pcam_min);
  const f3 pw_max = se3_apply(c->R, c->t, pcam_max);
  /* normalize (cuda_math.cuh:1075-1078) with rsqrtf restated as 1/sqrtf */
  f3 d = mk3(pw_max.x - pw_min.x, pw_max.y - pw_min.y, pw_max.z - pw_min.z);
  const float inv_len = 1.0f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
  const f3 dir = mk3(d.x * inv_len, d.y * inv_len, d.z * inv_len);

  i3 cur = world_to_block(vs, ext, pw_min);
  const i3 end = world_to_block(vs, ext, pw_max);
  const f3 step = mk3((float) signi(dir.x), (float) signi(dir.y), (float) signi(dir.z));
  const i3 nb = {cur.x + f2i(clampf(step.x, 0.0f, 1.f)), cur.y + f2i(clampf(step.y, 0.0f, 1.f)), cur.z + f2i(clampf(step.z, 0.0f, 1.f))};
  const f3 bw = voxel_to_world(vs, block_to_voxel(nb));
  const f3 boundary = mk3(bw.x - 0.5f * vs, bw.y - 0.5f * vs, bw.z - 0.5f * vs);
  f3 t_max = mk3((boundary.x - pw_min.x) / dir.x, (boundary.y - pw_min.y) / dir.y, (boundary.z - pw_min.z) / dir.z);
  f3 t_delta = mk3((step.x * (float) SDF_BLOCK_SIZE * vs) / dir.x, (step.y * (float) SDF_BLOCK_SIZE * vs) / dir.y,
                   (step.z * (float) SDF_BLOCK_SIZE * vs) / dir.z);
  const i3 bound = {f2i((float) end.x + step.x), f2i((float) end.y + step.y), f2i((float) end.z + step.z)};

  if (fabsf(dir.x) < FLOAT_EPSILON) { t_max.x = FLT_MAX; t_delta.x = FLT_MAX; }
  if (fabsf(boundary.x - dir.x) < FLOAT_EPSILON) { t_max.x = FLT_MAX; t_delta.x = FLT_MAX; }
  if (fabsf(dir.y) < FLOAT_EPSILON) { t_max.y = FLT_MAX; t_delta.y = FLT_MAX; }
  if (fabsf(boundary.y - dir.y) < FLOAT_EPSILON) { t_max.y = FLT_MAX; t_delta.y = FLT_MAX; }
  if (fabsf(dir.z) < FLOAT_EPSILON) { t_max.z = FLT_MAX; t_delta.z = FLT_MAX; }
  if (fabsf(boundary.z - dir.z) < FLOAT_EPSILON) { t_max.z = FLT_MAX; t_delta.z = FLT_MAX; }

  unsigned iter = 0;
  while (iter < MAX_DDA_ITERATION_COUNT) {
    if (owns_block(c, cur) && block_in_frustum_approx(c, cur)) (void) alloc_block(c, cur, 0, 0);
    if (t_max.x < t_max.y && t_max.x < t_max.z) {
      cur.x = f2i((float) cur.x + step.x);
      if (cur.x == bound.x) return;
      t_max.x += t_delta.x;
    } else if (t_max.z < t_max.y) {
      cur.z = f2i((float) cur.z + step.z);
      if (cur.z == bound.z) return;
      t_max.z += t_delta.z;
    } else {
      cur.y = f2i((float) cur.y + step.y);
      if (cur.y == bound.y) return;
      t_max.y += t_delta.y;
    }
    iter++;
  }
}

static void alloc_blocks_pass(mrh_ctx* c) {
  for (unsigned row = 0; row < c->rows; row++)
    for (unsigned col = 0; col < c->cols; col++) alloc_pixel(c, row, col);
}

/* vds.cu:860-871 allocateMemoryLow, n_blocks = low_blocks_to_allocate_ */
static void allocate_memory_low(mrh_ctx* c) {
  for (unsigned b = 0; b < c->low_blocks_to_allocate; b++) {
    const int addr_high = c->heap_counter_high; c->heap_counter_high -= 1;
    const int addr_low = c->heap_counter_low; c->heap_counter_low += OCTREE_BRANCHING_FACTOR;
    if (addr_high < 0) { c->error_flags |= 1u; continue; }
    for (int idx = 1; idx <= OCTREE_BRANCHING_FACTOR; idx++)
      c->heap_low[addr_low + idx] = c->heap_high[addr_high] * OCTREE_BRANCHING_FACTOR + OCTREE_BRANCHING_FACTOR - (unsigned) idx;
  }
}

/* vds.cu:874-922 allocBlocks (host loop) */
static void alloc_blocks(mrh_ctx* c) {
  int prev_free = heap_high_free(c) + heap_low_free(c);
  reset_mutex(c);
  if (c->p.sdf_var_threshold > 0.f && heap_low_free(c) < (int) c->low_blocks_to_allocate) allocate_memory_low(c);
  alloc_blocks_pass(c);
  for (;;) {
    reset_mutex(c);
    alloc_blocks_pass(c);
    const int cur_free = heap_high_free(c) + heap_low_free(c);
    if (prev_free == cur_free) break;
    prev_free = cur_free;
  }
}

/* ---- compaction: vds.cu:406-499 (canonical order C1) ------------------------------------- */

static int cmp_entry_pos(const void* a, const void* b) {
  const HashEntry* ea = (const HashEntry*) a;
  const HashEntry* eb = (const HashEntry*) b;
  if (ea->pos.x != eb->pos.x) return ea->pos.x < eb->pos.x ? -1 : 1;
  if (ea->pos.y != eb->pos.y) return ea->pos.y < eb->pos.y ? -1 : 1;
  if (ea->pos.z != eb->pos.z) return ea->pos.z < eb->pos.z ? -1 : 1;
  return 0;
}

static void flat_and_reduce(mrh_ctx* c, int use_camera) {
  /* resetCompactHashTableKernel touches all N entries; only [0, M) are ever read */
  for (unsigned i = 0; i < c->current_occupied; i++) delete_hash_entry(&c->compact[i]);
  unsigned n = 0;
  for (unsigned idx = 0; idx < c->total_size; idx++) {
    const HashEntry* e = &c->table[idx];
    if (e->ptr != FREE_ENTRY && (!use_camera || block_in_frustum_approx(c, e->pos))) c->compact[n++] = *e;
  }
  qsort(c->compact, n, sizeof(HashEntry), cmp_entry_pos);
  c->current_occupied = n;
}

/* ---- integration: vds.cu:1095-1212, vhu.cuh:167-181 -------------------------------------- */

/* returns 1 if the voxel was written.  with_variance = 0 restates reintegrateDepthMapKernel
 * (vds.cu:1942-2018), which is the same body without the delta / sum_squared lines. */
static inline int integrate_voxel(mrh_ctx* c, const HashEntry* entry, unsigned voxel_idx, int with_variance) {
  const float vs = c->p.virtual_voxel_size;
  const int scaling = 1 << entry->resolution;
  const i3 base = block_to_voxel(entry->pos);
  const i3 lc = delinearize(voxel_idx, SDF_BLOCK_SIZE / scaling);
  const i3 pi = {base.x + scaling * lc.x, base.y + scaling * lc.y, base.z + scaling * lc.z};
  const f3 pf = voxel_to_world(vs, pi);
  const f3 pcam = se3_apply(c->Ri, c->ti, pf);
  int row, col;
  if (!project_point(c, pcam, 0, &row, &col)) return 0;
  const float depth = get_depth(c, c->cloud[(size_t) row * c->cols + col]);
  if (depth == 0.f || depth > c->max_integration_distance) return 0;
  float sdf = depth - get_depth(c, pcam);
  const float truncation = get_truncation(depth, c->p.sdf_truncation, c->p.sdf_truncation_scale);
  if (sdf <= -truncation) return 0;
  if (sdf >= 0.f) sdf = fminf(truncation, sdf);
  else sdf = fmaxf(-truncation, sdf);
  const float weight_update = (float) (uint8_t) c->p.integration_weight_sample; /* uchar kernel arg, vds.cu:1101 */
  const uint8_t w1 = (uint8_t) weight_update;
  const uint8_t* px = &c->rgb[((size_t) row * c->cols + col) * 3];
  Voxel* dst = &c->blocks[(size_t) entry->ptr + voxel_idx];
  float curr_mean;
  if (dst->weight > 0) curr_mean = dst->sdf;
  else curr_mean = sdf;
  const float delta = (sdf - curr_mean) / (vs / 2);
  if (dst->weight == 0) { dst->rgb[0] = px[0]; dst->rgb[1] = px[1]; dst->rgb[2] = px[2]; }
  /* combineVoxel, vhu.cuh:167-181 */
  Voxel m;
  m.sum_squared = 0.f; /* Voxel() default, vhu.cuh:9-15 */
  m.rgb[0] = (uint8_t) f2i((0.5f * (float) dst->rgb[0] + 0.5f * (float) px[0]) + 0.5f);
  m.rgb[1] = (uint8_t) f2i((0.5f * (float) dst->rgb[1] + 0.5f * (float) px[1]) + 0.5f);
  m.rgb[2] = (uint8_t) f2i((0.5f * (float) dst->rgb[2] + 0.5f * (float) px[2]) + 0.5f);
  m.sdf = (dst->sdf * (float) dst->weight + sdf * (float) w1) / (float) ((int) dst->weight + (int) w1);
  {
    const int wsum = (int) dst->weight + (int) w1;
    const int wmax = (int) (uint8_t) c->p.integration_weight_max;
    m.weight = (uint8_t) (wsum < wmax ? wsum : wmax);
  }
  *dst = m;
  if (with_variance) {
    const float delta2 = (sdf - dst->sdf) / (vs / 2);
    dst->sum_squared = dst->sum_squared + delta * delta2; /* atomicAdd onto the just-stored 0 */
  }
  return 1;
}

static void integrate_depth_map(mrh_ctx* c) {
  uint64_t updated = 0;
  const long n = (long) c->current_occupied;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : updated)
#endif
  for (long e = 0; e < n; e++) {
    const HashEntry* entry = &c->compact[e];
    if (entry->ptr == FREE_ENTRY) continue;
    const unsigned nv = (unsigned) num_voxels_of(entry->resolution);
    for (unsigned v = 0; v < nv; v++) updated += (uint64_t) integrate_voxel(c, entry, v, 1);
  }
  c->last_updated = updated;
  c->total_updated += updated;
}

/* ---- variance-adaptive resolution: vds.cu:1857-1939, 2021-2107 -------------------------- */

static void check_var_sdf(mrh_ctx* c) {
  if (c->current_occupied == 0) return;
  reset_mutex(c);
  c->num_realloc = 0;
  for (unsigned b = 0; b < c->current_occupied; b++) {
    const HashEntry entry = c->compact[b];
    if (entry.resolution >= 1) continue;
    float s_sum[64], s_w[64];
    for (int tid = 0; tid < 64; tid++) {
      float local_sum_sq = 0.f, local_weight = 0.f;
      const int gx = (tid % 4) * 2, gy = ((tid / 4) % 4) * 2, gz = (tid / 16) * 2;
      for (int dz = 0; dz < 2; ++dz)
        for (int dy = 0; dy < 2; ++dy)
          for (int dx = 0; dx < 2; ++dx) {
            const int x = gx + dx, y = gy + dy, z = gz + dz;
            const int local_idx = z * 64 + y * 8 + x;
            const Voxel* v = &c->blocks[(size_t) entry.ptr + local_idx];
            if (v->weight > 0) { local_sum_sq += v->sum_squared; local_weight += (float) v->weight; }
          }
      s_sum[tid] = local_sum_sq;
      s_w[tid] = local_weight;
    }
    for (int stride = 32; stride > 0; stride /= 2)
      for (int t = 0; t < stride; t++) { s_sum[t] += s_sum[t + stride]; s_w[t] += s_w[t + stride]; }
    if (s_w[0] < 2) continue;
    const double avg_var = (double) (s_sum[0] / (s_w[0] - 1));
    if ((s_w[0] - 1) > 1e-6f && avg_var > 0.f && avg_var < (double) c->p.sdf_var_threshold) {
      if (delete_hash_entry_element(c, entry.pos)) {
        for (int i = 0; i < TOTAL_SDF_BLOCK_SIZE; ++i) delete_voxel(&c->blocks[(size_t) entry.ptr + i]);
        c->realloc_pos[c->num_realloc] = entry.pos;
        c->realloc_res[c->num_realloc] = entry.resolution + 1;
        c->num_realloc++;
      }
    }
  }
}

static void realloc_blocks_pass(mrh_ctx* c) {
  for (unsigned i = 0; i < c->num_realloc; i++) {
    const int idx = alloc_block(c, c->realloc_pos[i], c->realloc_res[i], 1);
    if (idx >= 0) c->reintegrate[c->num_reintegrate++] = (unsigned) idx;
  }
}

/* vds.cu:2037-2069 */
static void realloc_blocks(mrh_ctx* c) {
  /* D5: the reference clears d_num_reintegrate_ only when num_reallocate > 0 (vds.cu:2040-2043), so a
   * frame that coarsens nothing replays the previous frame's slot list against whatever now sits in
   * those slots (possibly free entries with ptr = -2: an out-of-bounds write).  No canonical value
   * exists for that; the list is cleared every frame here. */
  c->num_reintegrate = 0;
  if (c->num_realloc == 0) return;
  int prev_free = heap_high_free(c) + heap_low_free(c);
  reset_mutex(c);
  realloc_blocks_pass(c);
  for (;;) {
    reset_mutex(c);
    realloc_blocks_pass(c);
    const int cur_free = heap_high_free(c) + heap_low_free(c);
    if (prev_free == cur_free) break;
    prev_free = cur_free;
  }
}

/* vds.cu:2087-2107 + kernel :1942-2018; D2: blockDim = (16,1,1), gridDim.y = 32 -> voxel 0..31 */
static void reintegrate_depth_map(mrh_ctx* c) {
  for (unsigned i = 0; i < c->num_reintegrate; i++) {
    const HashEntry entry = c->table[c->reintegrate[i]];
    const unsigned nv = (unsigned) num_voxels_of(entry.resolution);
    for (unsigned v = 0; v < 32 && v < nv; v++) (void) integrate_voxel(c, &entry, v, 0);
  }
}

/* ---- garbage collection: voxel_data_structures.cpp:137-145, vds.cu:1583-1713, 1827-1844 -- */

/* vds.cu:1583-1585 pack(tid, depth) = (depth bits << 32) + tid.  The thread id is the race-ordered compact index
 * in the reference; the canonical tie-break (C1) is (block position, voxel index), which needs 63 + 9 bits, so the
 * key is split over two min-buffers: hi = depth bits << 32 | key72 >> 40, lo = key72 & (2^40 - 1), with
 * key72 = packed block position << 9 | voxel index.  For one map this orders candidates exactly like
 * pack(512 * sorted_block_index + voxel, depth); for tile shards it is also independent of which rank holds a
 * block, so an element-wise MIN over ranks gives the single-map z-buffer. */
static inline uint64_t pack_block_key(i3 b) {
  return ((uint64_t) (uint32_t) (b.x + (1 << 20)) << 42) | ((uint64_t) (uint32_t) (b.y + (1 << 20)) << 21) | (uint64_t) (uint32_t) (b.z + (1 << 20));
}

/* pass 0: zbuf0 = min hi; pass 1: among hi-winners zbuf1 = min lo; pass 2: the winner loses one weight */
static void starve_pass(mrh_ctx* c, int pass) {
  const size_t npix = (size_t) c->rows * c->cols;
  if (pass == 0) {
    if (c->depth_buff_n < npix) {
      free(c->depth_buff);
      c->depth_buff = (uint64_t*) malloc(2 * npix * sizeof(uint64_t));
      c->depth_buff_n = npix;
    }
    for (size_t i = 0; i < 2 * npix; i++) c->depth_buff[i] = (uint64_t) INT64_MAX;
  }
  uint64_t* z0 = c->depth_buff;
  uint64_t* z1 = c->depth_buff + npix;
  const float vs = c->p.virtual_voxel_size;
  for (unsigned b = 0; b < c->current_occupied; b++) {
    const HashEntry* entry = &c->compact[b];
    const unsigned nthreads = entry->resolution == 0 ? 512u : 64u; /* D3 */
    const i3 base = block_to_voxel(entry->pos);
    const uint64_t key = pack_block_key(entry->pos);
    for (unsigned i = 0; i < nthreads; i++) {
      const i3 lc = delinearize(i, SDF_BLOCK_SIZE);
      const i3 pi = {base.x + lc.x, base.y + lc.y, base.z + lc.z};
      const f3 pcam = se3_apply(c->Ri, c->ti, voxel_to_world(vs, pi));
      const float depth = get_depth(c, pcam);
      if (depth < c->min_depth) continue;
      int row, col;
      if (!project_point(c, pcam, 0, &row, &col)) continue;
      uint32_t dbits;
      memcpy(&dbits, &depth, 4);
      const uint64_t hi = ((uint64_t) dbits << 32) | (key >> 31);
      const uint64_t lo = ((key & 0x7FFFFFFFull) << 9) | (uint64_t) i;
      const size_t pix = (size_t) row * c->cols + col;
      if (pass == 0) {
        if (hi < z0[pix]) z0[pix] = hi;
      } else if (pass == 1) {
        if (z0[pix] == hi && lo < z1[pix]) z1[pix] = lo;
      } else if (z0[pix] == hi && z1[pix] == lo) {
        Voxel* v = &c->blocks[(size_t) entry->ptr + i];
        const int w = (int) v->weight - 1;
        v->weight = (uint8_t) (w > 0 ? w : 0);
      }
    }
  }
}

static void gc_identify(mrh_ctx* c) {
  const float thr = get_truncation(c->max_depth, c->p.sdf_truncation, c->p.sdf_truncation_scale);
  const long n = (long) c->current_occupied;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16)
#endif
  for (long b = 0; b < n; b++) {
    const HashEntry* entry = &c->compact[b];
    const int nv = num_voxels_of(entry->resolution);
    float min_sdf = FLT_MAX;
    int max_w = 0;
    for (int i = 0; i < nv; i++) {
      const Voxel* v = &c->blocks[(size_t) entry->ptr + i];
      const float s = (v->weight == 0) ? FLT_MAX : fabsf(v->sdf);
      min_sdf = fminf(min_sdf, s);
      if ((int) v->weight > max_w) max_w = v->weight;
    }
    c->decision[b] = (min_sdf >= thr || max_w == 0) ? 1 : 0;
  }
}

static void gc_free(mrh_ctx* c) {
  for (unsigned idx = 0; idx < c->current_occupied; idx++) {
    if (c->decision[idx] == 0) continue;
    const HashEntry entry = c->compact[idx];
    const int vbv = num_voxels_of(entry.resolution);
    if (delete_hash_entry_element(c, entry.pos)) {
      for (int i = 0; i < vbv; ++i) delete_voxel(&c->blocks[(size_t) entry.ptr + i]);
      if (c->profile) c->last_freed++;
    }
  }
}

static int is_starve_frame(const mrh_ctx* c, int max_num_frames) {
  return max_num_frames > 0 && c->frames > 0 && c->frames % (uint64_t) max_num_frames == 0;
}

/* garbageCollect after the (optional) starve: identify, resetHashBucketMutex, free */
static void gc_tail(mrh_ctx* c) {
  gc_identify(c);
  reset_mutex(c);
  gc_free(c);
}

/* ---- marching cubes: marching_cubes.cu:7-305, mesh_extractor.cu:6-55 --------------------- */

/* mesh_extractor.cu:6-36 vertexInterp */
static mrh_vertex vertex_interp(float isolevel, f3 p1, f3 p2, float d1, float d2, const uint8_t* c1, const uint8_t* c2) {
  mrh_vertex r1, r2, res;
  r1.p[0] = p1.x; r1.p[1] = p1.y; r1.p[2] = p1.z;
  r1.c[0] = (float) c1[0] / 255.f; r1.c[1] = (float) c1[1] / 255.f; r1.c[2] = (float) c1[2] / 255.f;
  r2.p[0] = p2.x; r2.p[1] = p2.y; r2.p[2] = p2.z;
  r2.c[0] = (float) c2[0] / 255.f; r2.c[1] = (float) c2[1] / 255.f; r2.c[2] = (float) c2[2] / 255.f;
  if (fabsf(isolevel - d1) < 0.00001f) return r1;
  if (fabsf(isolevel - d2) < 0.00001f) return r2;
  if (fabsf(d1 - d2) < 0.00001f) return r1;
  const float mu = (isolevel - d1) / (d2 - d1);
  res.p[0] = p1.x + mu * (p2.x - p1.x);
  res.p[1] = p1.y + mu * (p2.y - p1.y);
  res.p[2] = p1.z + mu * (p2.z - p1.z);
  res.c[0] = (float) c1[0] + mu * (float) ((int) c2[0] - (int) c1[0]) / 255.f;
  res.c[1] = (float) c1[1] + mu * (float) ((int) c2[1] - (int) c1[1]) / 255.f;
  res.c[2] = (float) c1[2] + mu * (float) ((int) c2[2] - (int) c1[2]) / 255.f;
  return res;
}

static int push_triangle(mrh_ctx* c, const mrh_triangle* t) {
  if (c->ntris >= c->max_triangles) { c->error_flags |= 8u; return 0; } /* mesh_extractor.cu:42-45 */
  if (c->ntris == c->cap_tris) {
    uint64_t ncap = c->cap_tris ? c->cap_tris * 2 : 4096;
    if (ncap > c->max_triangles) ncap = c->max_triangles;
    mrh_triangle* nt = (mrh_triangle*) realloc(c->tris, ncap * sizeof(mrh_triangle));
    if (!nt) { c->error_flags |= 8u; return 0; }
    c->tris = nt;
    c->cap_tris = ncap;
  }
  c->tris[c->ntris++] = *t;
  return 1;
}

/* marching_cubes.cu:7-69 checkVertexVoxels */
static void check_vertex_voxels(const mrh_ctx* c, f3 pf, f3* sP, f3* sM) {
  const float vvs = get_voxel_size_f(c, pf);
  float vs;
  vs = get_voxel_size_f(c, mk3(pf.x + sP->x, pf.y + 0.0f, pf.z + 0.0f));
  if (vs > 0 && vs < 1 && vs != vvs) sP->x *= 0.499f;
  vs = get_voxel_size_f(c, mk3(pf.x + sM->x, pf.y + 0.0f, pf.z + 0.0f));
  if (vs > 0 && vs < 1 && vs != vvs) sM->x *= 0.499f;
  vs = get_voxel_size_f(c, mk3(pf.x + 0.0f, pf.y + sP->y, pf.z + 0.0f));
  if (vs > 0 && vs < 1 && vs != vvs) sP->y *= 0.499f;
  vs = get_voxel_size_f(c, mk3(pf.x + 0.0f, pf.y + sM->y, pf.z + 0.0f));
  if (vs > 0 && vs < 1 && vs != vvs) sM->y *= 0.499f;
  vs = get_voxel_size_f(c, mk3(pf.x + 0.0f, pf.y + 0.0f, pf.z + sP->z));
  if (vs > 0 && vs < 1 && vs != vvs) sP->z *= 0.499f;
  vs = get_voxel_size_f(c, mk3(pf.x + 0.0f, pf.y + 0.0f, pf.z + sM->z));
  if (vs > 0 && vs < 1 && vs != vvs) sM->z *= 0.499f;
}

/* marching_cubes.cu:72-261 extractIsoSurfaceAtPosition (positive/negative scaling = 1); the voxel's triangles go to out[0 .. n),
 * n <= 5 is returned (the caller appends them in canonical order: blocks are evaluated in parallel, appended in sequence) */
static int extract_at_position(const mrh_ctx* c, f3 pf, mrh_triangle* out) {
  const float isolevel = 0.f;
  const float vvs = get_voxel_size_f(c, pf);
  const float P = vvs * 0.5f;
  const float M = -P;
  f3 sP = mk3(P * 1.f, P * 1.f, P * 1.f);
  f3 sM = mk3(M * 1.f, M * 1.f, M * 1.f);
  check_vertex_voxels(c, pf, &sP, &sM);

  f3 p[8];
  float dist[8];
  Voxel vox[8];
  /* corner k = x + 2y + 4z; evaluation order 000,001,010,011,100,101,110,111 with "001" = +x */
  for (int k = 0; k < 8; k++) {
    p[k] = mk3(pf.x + ((k & 1) ? sP.x : sM.x), pf.y + ((k & 2) ? sP.y : sM.y), pf.z + ((k & 4) ? sP.z : sM.z));
    const int valid = trilinear(c, p[k], &dist[k]);
    vox[k] = get_voxel_f(c, p[k], NULL);
    if (!valid) {
      if (vox[k].weight < c->p.min_weight_threshold) return 0;
      dist[k] = vox[k].sdf;
    }
  }
  unsigned cube_index = 0;
  for (int k = 0; k < 8; k++)
    if (dist[k] < isolevel) cube_index += (1u << k);

  const float thr = c->p.marching_cubes_threshold;
  for (unsigned k = 0; k < 8; k++)
    for (unsigned l = 0; l < 8; l++) {
      if (dist[k] * dist[l] < 0.f) {
        if (fabsf(dist[k]) + fabsf(dist[l]) > thr) return 0;
      } else {
        if (fabsf(dist[k] - dist[l]) > thr) return 0;
      }
    }
  for (int k = 0; k < 8; k++)
    if (fabsf(dist[k]) > thr) return 0;

  const uint8_t* row = MC_TRI[cube_index];
  const int ntri = row[0];
  for (int j = 0; j < ntri; j++) {
    mrh_triangle t;
    for (int k = 0; k < 3; k++) {
      const int code = row[1 + 3 * j + k];
      const int a = code >> 4, b = code & 0xF;
      t.v[k] = vertex_interp(isolevel, p[a], p[b], dist[a], dist[b], vox[a].rgb, vox[b].rgb);
    }
    out[j] = t;
  }
  return ntri;
}

/* mesh_extractor.cpp:95-98 + marching_cubes.cu:264-305 */
static void extract_iso_surface(mrh_ctx* c) {
  flat_and_reduce(c, 0);
  c->ntris = 0;
  free(c->tri_blocks); free(c->tri_counts);
  c->n_tri_blocks = c->current_occupied;
  c->tri_blocks = (mrh_block_desc*) calloc(c->current_occupied ? c->current_occupied : 1, sizeof(mrh_block_desc));
  c->tri_counts = (uint32_t*) calloc(c->current_occupied ? c->current_occupied : 1, sizeof(uint32_t));
  const float vs = c->p.virtual_voxel_size;
  const long nblk = (long) c->current_occupied;
  /* per block: its triangles in voxel order (blocks are independent: evaluated in parallel, appended below in list order) */
  mrh_triangle** per_block = (mrh_triangle**) calloc(nblk ? (size_t) nblk : 1, sizeof(mrh_triangle*));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8)
#endif
  for (long e = 0; e < nblk; e++) {
    const HashEntry* entry = &c->compact[e];
    const int scaling = 1 << entry->resolution;
    const unsigned nv = (unsigned) num_voxels_of(entry->resolution);
    const i3 base = block_to_voxel(entry->pos);
    uint32_t n = 0;
    mrh_triangle* buf = NULL;
    if (owns_block(c, entry->pos)) { /* halo blocks of other shards are looked up but emit nothing */
      buf = (mrh_triangle*) malloc((size_t) nv * 5 * sizeof(mrh_triangle));
      for (unsigned v = 0; v < nv; v++) {
        const i3 lc = delinearize(v, SDF_BLOCK_SIZE / scaling);
        const i3 pi = {base.x + scaling * lc.x, base.y + scaling * lc.y, base.z + scaling * lc.z};
        n += (uint32_t) extract_at_position(c, voxel_to_world(vs, pi), buf + n);
      }
    }
    per_block[e] = buf;
    c->tri_blocks[e].x = entry->pos.x; c->tri_blocks[e].y = entry->pos.y; c->tri_blocks[e].z = entry->pos.z;
    c->tri_blocks[e].resolution = entry->resolution;
    c->tri_counts[e] = n;
  }
  for (long e = 0; e < nblk; e++) {
    uint32_t kept = 0;
    for (uint32_t j = 0; j < c->tri_counts[e]; j++) kept += (uint32_t) push_triangle(c, &per_block[e][j]);
    c->tri_counts[e] = kept;
    free(per_block[e]);
  }
  free(per_block);
}

/* ---- host mesh assembly: mesh_extractor.cpp:9-76, 156-259 --------------------------------- */

typedef struct { uint64_t k[3]; int idx; int used; } VSlot;
typedef struct { int32_t f[3]; int used; } FSlot;

static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

/* Exported so the mesh post-process can be tested on hand-made inputs exactly like
 * tests/test_marching_cubes.cpp does.  in_v: [n,3] f64, in_f: [nf,3] i32.
 * out_map [n], out_v [<=n,3], out_f [nf,3]; returns number of unique vertices. */
int64_t orc_remove_duplicate_vertices(const double* in_v, int64_t n, const int32_t* in_f, int64_t nf, double epsilon,
                                      double* out_v, int32_t* out_f, int32_t* out_map) {
  size_t cap = 16;
  while (cap < (size_t) n * 2 + 1) cap <<= 1;
  VSlot* tab = (VSlot*) calloc(cap, sizeof(VSlot));
  int64_t nu = 0;
  const double inv_eps = epsilon != 0.0 ? 1.0 / epsilon : 0.0;
  for (int64_t i = 0; i < n; i++) {
    uint64_t k[3];
    for (int a = 0; a < 3; a++) {
      if (epsilon == 0.0) memcpy(&k[a], &in_v[i * 3 + a], 8); /* D4 */
      else { const int32_t q = (int32_t) floor(in_v[i * 3 + a] * inv_eps); k[a] = (uint64_t) (uint32_t) q; }
    }
    size_t h = (size_t) (mix64(k[0] ^ mix64(k[1] ^ mix64(k[2]))) & (cap - 1));
    /* a vertex with a NaN coordinate never compares equal (Vector3dEqual uses ==): always a new vertex */
    if (in_v[i * 3] != in_v[i * 3] || in_v[i * 3 + 1] != in_v[i * 3 + 1] || in_v[i * 3 + 2] != in_v[i * 3 + 2]) {
      out_v[nu * 3 + 0] = in_v[i * 3 + 0]; out_v[nu * 3 + 1] = in_v[i * 3 + 1]; out_v[nu * 3 + 2] = in_v[i * 3 + 2];
      out_map[i] = (int32_t) nu;
      nu++;
      continue;
    }
    for (;;) {
      if (!tab[h].used) {
        tab[h].used = 1; tab[h].k[0] = k[0]; tab[h].k[1] = k[1]; tab[h].k[2] = k[2]; tab[h].idx = (int) nu;
        out_v[nu * 3 + 0] = in_v[i * 3 + 0]; out_v[nu * 3 + 1] = in_v[i * 3 + 1]; out_v[nu * 3 + 2] = in_v[i * 3 + 2];
        out_map[i] = (int32_t) nu;
        nu++;
        break;
      }
      if (tab[h].k[0] == k[0] && tab[h].k[1] == k[1] && tab[h].k[2] == k[2]) { out_map[i] = tab[h].idx; break; }
      h = (h + 1) & (cap - 1);
    }
  }
  free(tab);
  for (int64_t i = 0; i < nf * 3; i++) out_f[i] = out_map[in_f[i]];
  return nu;
}

/* mesh_extractor.cpp:156-178: keeps first occurrences; returns the number kept */
int64_t orc_remove_duplicate_faces(const int32_t* in_f, int64_t nf, int32_t* out_f) {
  size_t cap = 16;
  while (cap < (size_t) nf * 2 + 1) cap <<= 1;
  FSlot* tab = (FSlot*) calloc(cap, sizeof(FSlot));
  int64_t nu = 0;
  for (int64_t i = 0; i < nf; i++) {
    const int32_t* f = &in_f[i * 3];
    size_t h = (size_t) (mix64((uint64_t) (uint32_t) f[0] ^ mix64((uint64_t) (uint32_t) f[1] ^ mix64((uint64_t) (uint32_t) f[2]))) & (cap - 1));
    int dup = 0;
    for (;;) {
      if (!tab[h].used) { tab[h].used = 1; tab[h].f[0] = f[0]; tab[h].f[1] = f[1]; tab[h].f[2] = f[2]; break; }
      if (tab[h].f[0] == f[0] && tab[h].f[1] == f[1] && tab[h].f[2] == f[2]) { dup = 1; break; }
      h = (h + 1) & (cap - 1);
    }
    if (!dup) { out_f[nu * 3 + 0] = f[0]; out_f[nu * 3 + 1] = f[1]; out_f[nu * 3 + 2] = f[2]; nu++; }
  }
  free(tab);
  return nu;
}

/* mesh_extractor.cpp:9-76 processTriangles.  merge_mesh_ == false, or an empty running mesh: the new soup alone (:24-31).
 * merge_mesh_ == true with a running mesh: `combine` appends the soup's vertices and its faces (shifted by the number of
 * running vertices) to the running, already de-duplicated mesh, colours likewise (:32-37), and the whole goes through the
 * vertex merge, first-assigned colours, degenerate-face and repeated-face removal again (:39-75) - restated literally,
 * incrementally, as the reference runs it. */
static void process_triangles(mrh_ctx* c) {
  const int64_t nt = (int64_t) c->ntris;
  const int merging = c->merge_on && (c->nv || c->nf);
  const int64_t pv = merging ? (int64_t) c->nv : 0, pf = merging ? (int64_t) c->nf : 0; /* running mesh */
  const int64_t n = pv + nt * 3, nfa = pf + nt;
  if (!merging) {
    free(c->V); free(c->C); free(c->F);
    c->V = c->C = NULL; c->F = NULL; c->nv = c->nf = 0;
  }
  if (nt == 0) return;
  double* nv = (double*) malloc((size_t) n * 3 * sizeof(double));
  double* nc = (double*) malloc((size_t) n * 3 * sizeof(double));
  int32_t* nf = (int32_t*) malloc((size_t) nfa * 3 * sizeof(int32_t));
  if (merging) {
    memcpy(nv, c->V, (size_t) pv * 3 * sizeof(double));
    memcpy(nc, c->C, (size_t) pv * 3 * sizeof(double));
    memcpy(nf, c->F, (size_t) pf * 3 * sizeof(int32_t));
    free(c->V); free(c->C); free(c->F);
    c->V = c->C = NULL; c->F = NULL; c->nv = c->nf = 0;
  }
  for (int64_t i = 0; i < nt; i++)
    for (int k = 0; k < 3; k++) {
      for (int a = 0; a < 3; a++) {
        nv[(pv + i * 3 + k) * 3 + a] = (double) c->tris[i].v[k].p[a];
        nc[(pv + i * 3 + k) * 3 + a] = (double) c->tris[i].v[k].c[a];
      }
      nf[(pf + i) * 3 + k] = (int32_t) (pv + i * 3 + k);
    }
  double* uv = (double*) malloc((size_t) n * 3 * sizeof(double));
  int32_t* uf = (int32_t*) malloc((size_t) nfa * 3 * sizeof(int32_t));
  int32_t* map = (int32_t*) malloc((size_t) n * sizeof(int32_t));
  const int64_t nu = orc_remove_duplicate_vertices(nv, n, nf, nfa, (double) c->p.vertices_merging_threshold, uv, uf, map);
  double* uc = (double*) malloc((size_t) (nu ? nu : 1) * 3 * sizeof(double));
  uint8_t* assigned = (uint8_t*) calloc((size_t) (nu ? nu : 1), 1);
  for (int64_t i = 0; i < n; i++) {
    const int32_t ni = map[i];
    if (!assigned[ni]) { uc[ni * 3 + 0] = nc[i * 3 + 0]; uc[ni * 3 + 1] = nc[i * 3 + 1]; uc[ni * 3 + 2] = nc[i * 3 + 2]; assigned[ni] = 1; }
  }
  int64_t kept = 0;
  for (int64_t i = 0; i < nfa; i++) {
    const int32_t a = uf[i * 3], b = uf[i * 3 + 1], d = uf[i * 3 + 2];
    if (a != b && a != d && b != d) { nf[kept * 3] = a; nf[kept * 3 + 1] = b; nf[kept * 3 + 2] = d; kept++; }
  }
  int32_t* ff = (int32_t*) malloc((size_t) (kept ? kept : 1) * 3 * sizeof(int32_t));
  const int64_t nfu = orc_remove_duplicate_faces(nf, kept, ff);
  c->V = uv; c->C = uc; c->F = ff; c->nv = (uint64_t) nu; c->nf = (uint64_t) nfu;
  free(nv); free(nc); free(nf); free(uf); free(map); free(assigned);
}

/* ---- C ABI -------------------------------------------------------------------------------- */

const char* mrh_version(void) {
#ifdef _OPENMP
  return "mrh_oracle abi1 cpu-restatement openmp";
#else
  return "mrh_oracle abi1 cpu-restatement serial";
#endif
}

const char* mrh_last_error(const mrh_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

static void init_buffers(mrh_ctx* c) {
  c->n_halo = 0;
  /* voxel_data_structures.cpp:58-87 resetBuffers + ctor counters (voxel_data_structures.cuh:88-96) */
  for (unsigned i = 0; i < c->num_sdf_blocks; i++) {
    for (unsigned j = 0; j < 8; j++) c->heap_low[(size_t) i * 8 + j] = c->num_sdf_blocks * 8;
    c->heap_high[i] = c->num_sdf_blocks - 1 - i;
  }
  memset(c->blocks, 0, (size_t) c->num_sdf_blocks * TOTAL_SDF_BLOCK_SIZE * sizeof(Voxel));
  for (unsigned i = 0; i < c->total_size; i++) {
    c->table[i].pos.x = c->table[i].pos.y = c->table[i].pos.z = 0;
    c->table[i].offset = NO_OFFSET; c->table[i].ptr = FREE_ENTRY; c->table[i].resolution = 0;
    c->compact[i] = c->table[i];
    c->decision[i] = 0;
  }
  for (unsigned i = 0; i < c->hash_num_buckets; i++) c->mutex[i] = FREE_ENTRY;
  c->heap_counter_high = (int) c->num_sdf_blocks - 1;
  c->heap_counter_low = -1;
  c->current_occupied = 0;
  c->num_realloc = c->num_reintegrate = 0;
  c->frames = 0;
  c->ntris = 0;
  c->last_updated = c->last_inserted = c->last_freed = c->total_updated = c->total_compact = 0;
  c->error_flags = 0;
}

int mrh_create(const mrh_params* p, mrh_ctx** out) {
  if (!p || !out) return fail(NULL, MRH_ERR_INVALID_ARG, "mrh_create: null argument");
  if (p->abi_version != MRH_ABI_VERSION) return fail(NULL, MRH_ERR_INVALID_ARG, "mrh_create: abi_version mismatch");
  if (!(p->virtual_voxel_size > 0.f)) return fail(NULL, MRH_ERR_INVALID_ARG, "mrh_create: virtual_voxel_size must be > 0");
  if (!(p->sdf_truncation >= 0.f) || !(p->sdf_truncation_scale >= 0.f))
    return fail(NULL, MRH_ERR_INVALID_ARG, "mrh_create: sdf_truncation and sdf_truncation_scale must be >= 0");
  mrh_ctx* c = (mrh_ctx*) calloc(1, sizeof(mrh_ctx));
  if (!c) return fail(NULL, MRH_ERR_DEVICE, "mrh_create: out of host memory");
  c->p = *p;
  if (c->p.integration_weight_max == 0) c->p.integration_weight_max = 255;
  if (c->p.voxel_extents_scale == 0) c->p.voxel_extents_scale = 1;
  c->num_sdf_blocks = p->num_sdf_blocks ? (unsigned) p->num_sdf_blocks : 65536u;
  /* geowrapper.cpp:50: hash_num_buckets = num_sdf_blocks */
  c->hash_num_buckets = p->hash_slots ? (unsigned) (p->hash_slots / HASH_BUCKET_SIZE) : c->num_sdf_blocks;
  if (c->hash_num_buckets == 0) c->hash_num_buckets = 1;
  c->total_size = c->hash_num_buckets * HASH_BUCKET_SIZE;
  c->low_blocks_to_allocate = (unsigned) ((float) c->num_sdf_blocks * 0.1f);
  c->max_triangles = p->max_triangles ? p->max_triangles : (uint64_t) 1 << 26;
  c->table = (HashEntry*) malloc((size_t) c->total_size * sizeof(HashEntry));
  c->compact = (HashEntry*) malloc((size_t) c->total_size * sizeof(HashEntry));
  c->decision = (int*) malloc((size_t) c->total_size * sizeof(int));
  c->mutex = (int*) malloc((size_t) c->hash_num_buckets * sizeof(int));
  c->heap_high = (unsigned*) malloc(((size_t) c->num_sdf_blocks + 1) * sizeof(unsigned));
  c->heap_low = (unsigned*) malloc(((size_t) c->num_sdf_blocks * 8 + 9) * sizeof(unsigned));
  c->blocks = (Voxel*) malloc((size_t) c->num_sdf_blocks * TOTAL_SDF_BLOCK_SIZE * sizeof(Voxel));
  c->realloc_pos = (i3*) malloc((size_t) c->num_sdf_blocks * sizeof(i3));
  c->realloc_res = (int*) malloc((size_t) c->num_sdf_blocks * sizeof(int));
  c->reintegrate = (unsigned*) malloc((size_t) c->num_sdf_blocks * sizeof(unsigned));
  if (!c->table || !c->compact || !c->decision || !c->mutex || !c->heap_high || !c->heap_low || !c->blocks || !c->realloc_pos ||
      !c->realloc_res || !c->reintegrate) {
    mrh_destroy(c);
    return fail(NULL, MRH_ERR_DEVICE, "mrh_create: out of host memory");
  }
  init_buffers(c);
  /* geowrapper.cpp:80 placeholder camera */
  mrh_set_camera(c, 1.f, 1.f, 0.f, 0.f, 1, 1, p->min_depth, p->max_depth, MRH_CAMERA_SPHERICAL);
  c->has_camera = 0;
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const float z[3] = {0, 0, 0};
  mrh_set_pose(c, I, z);
  *out = c;
  return MRH_OK;
}

int mrh_destroy(mrh_ctx* c) {
  if (!c) return MRH_OK;
  free(c->table); free(c->compact); free(c->decision); free(c->mutex); free(c->heap_high); free(c->heap_low);
  free(c->blocks); free(c->realloc_pos); free(c->realloc_res); free(c->reintegrate); free(c->depth_buff);
  free(c->depth); free(c->rgb); free(c->points); free(c->normals); free(c->cloud); free(c->tris); free(c->tri_blocks); free(c->tri_counts); free(c->V); free(c->C); free(c->F);
  free(c->qt_leaves); free(c->seeds); free(c->pack); free(c->halo);
  free(c);
  return MRH_OK;
}

int mrh_reset(mrh_ctx* c) {
  if (!c) return MRH_ERR_INVALID_ARG;
  init_buffers(c);
  return MRH_OK;
}

int mrh_set_camera(mrh_ctx* c, float fx, float fy, float cx, float cy, int rows, int cols, float min_depth, float max_depth,
                   int model) {
  if (!c) return MRH_ERR_INVALID_ARG;
  if (rows <= 0 || cols <= 0 || (model != MRH_CAMERA_PINHOLE && model != MRH_CAMERA_SPHERICAL))
    return fail(c, MRH_ERR_INVALID_ARG, "mrh_set_camera: bad rows/cols/model");
  /* camera.cuh:19-34 */
  c->fx = fx; c->fy = fy; c->ifx = 1.f / fx; c->ify = 1.f / fy; c->cx = cx; c->cy = cy;
  c->rows = (unsigned) rows; c->cols = (unsigned) cols;
  c->row_threshold = (int) ((float) c->rows * 0.5f);
  c->col_threshold = (int) ((float) c->cols * 0.5f);
  c->min_depth = min_depth; c->max_depth = max_depth; c->model = model;
  c->max_integration_distance = max_depth; /* geowrapper.cpp:111 */
  free(c->cloud);
  c->cloud = (f3*) calloc((size_t) rows * cols, sizeof(f3));
  c->has_camera = 1;
  return MRH_OK;
}

int mrh_set_pose(mrh_ctx* c, const float R[9], const float t[3]) {
  if (!c || !R || !t) return MRH_ERR_INVALID_ARG;
  memcpy(c->R, R, 36); memcpy(c->t, t, 12);
  se3_inverse(c->R, c->t, c->Ri, c->ti);
  return MRH_OK;
}

int mrh_upload_depth(mrh_ctx* c, const float* depth, int rows, int cols) {
  if (!c || !depth || rows <= 0 || cols <= 0) return fail(c, MRH_ERR_INVALID_ARG, "mrh_upload_depth: bad argument");
  free(c->depth);
  c->depth = (float*) malloc((size_t) rows * cols * sizeof(float));
  memcpy(c->depth, depth, (size_t) rows * cols * sizeof(float));
  c->depth_rows = rows; c->depth_cols = cols;
  return MRH_OK;
}

int mrh_upload_rgb(mrh_ctx* c, const uint8_t* rgb, int rows, int cols) {
  if (!c || !rgb || rows <= 0 || cols <= 0) return fail(c, MRH_ERR_INVALID_ARG, "mrh_upload_rgb: bad argument");
  free(c->rgb);
  c->rgb = (uint8_t*) malloc((size_t) rows * cols * 3);
  memcpy(c->rgb, rgb, (size_t) rows * cols * 3);
  c->rgb_rows = rows; c->rgb_cols = cols;
  return MRH_OK;
}

int mrh_set_depth_device(mrh_ctx* c, const float* d, int rows, int cols) { return mrh_upload_depth(c, d, rows, cols); }
int mrh_set_rgb_device(mrh_ctx* c, const uint8_t* d, int rows, int cols) { return mrh_upload_rgb(c, d, rows, cols); }

/* ---- LiDAR point-cloud integration: allocBlocks3DKernel vds.cu:925-1033, integrate3DKernel
 * vds.cu:1215-1379, VoxelContainer::integrate(point_cloud, ...) voxel_data_structures.cpp:112-135 ------------------
 * Scope: the shipped LiDAR configurations (vbr / maicity / newer_college .cfg: projective_sdf = true, no normals,
 * n_frames_invalidate_voxels = 0, sdf_var_threshold = 0).  norm3df(x, y, z) is restated as sqrtf((x*x + y*y) + z*z)
 * and normalize() as a multiplication by 1 / sqrtf(dot) (the arithmetic spec of this header).
 * D6 (canonicalisation): integrate3DKernel updates a voxel with a non-atomic read-modify-write per point, so points
 * of one scan that traverse the same voxel race in the reference.  Canonical result: every voxel receives its
 * updates in ascending point index, each point's walk run to completion — i.e. this sequential loop. */
static inline float norm3(f3 p) { return sqrtf((p.x * p.x + p.y * p.y) + p.z * p.z); }
static inline f3 normalize3(f3 p) {
  const float inv = 1.0f / sqrtf((p.x * p.x + p.y * p.y) + p.z * p.z);
  return mk3(p.x * inv, p.y * inv, p.z * inv);
}

/* the DDA set-up shared by both kernels (vds.cu:966-1005 / :1257-1296); unit = voxels per step (8 for blocks, 1 for voxels) */
typedef struct { i3 cur, bound; f3 step, t_max, t_delta; } dda3;
static dda3 dda_setup(const mrh_ctx* c, f3 pw_min, f3 pw_max, int blocks) {
  const float vs = c->p.virtual_voxel_size;
  const float ext = (float) c->p.voxel_extents_scale;
  dda3 r;
  const f3 dir = normalize3(mk3(pw_max.x - pw_min.x, pw_max.y - pw_min.y, pw_max.z - pw_min.z));
  i3 end;
  if (blocks) { r.cur = world_to_block(vs, ext, pw_min); end = world_to_block(vs, ext, pw_max); }
  else { r.cur = world_to_voxel(vs, pw_min); end = world_to_voxel(vs, pw_max); }
  r.step = mk3((float) signi(dir.x), (float) signi(dir.y), (float) signi(dir.z));
  const i3 nb = {r.cur.x + f2i(clampf(r.step.x, 0.0f, 1.f)), r.cur.y + f2i(clampf(r.step.y, 0.0f, 1.f)), r.cur.z + f2i(clampf(r.step.z, 0.0f, 1.f))};
  const f3 bw = voxel_to_world(vs, blocks ? block_to_voxel(nb) : nb);
  const f3 boundary = mk3(bw.x - 0.5f * vs, bw.y - 0.5f * vs, bw.z - 0.5f * vs);
  const float unit = blocks ? (float) SDF_BLOCK_SIZE : 1.0f;
  r.t_max = mk3((boundary.x - pw_min.x) / dir.x, (boundary.y - pw_min.y) / dir.y, (boundary.z - pw_min.z) / dir.z);
  if (blocks) r.t_delta = mk3((r.step.x * unit * vs) / dir.x, (r.step.y * unit * vs) / dir.y, (r.step.z * unit * vs) / dir.z);
  else r.t_delta = mk3((r.step.x * vs) / dir.x, (r.step.y * vs) / dir.y, (r.step.z * vs) / dir.z);
  r.bound.x = f2i((float) end.x + r.step.x); r.bound.y = f2i((float) end.y + r.step.y); r.bound.z = f2i((float) end.z + r.step.z);
  if (fabsf(dir.x) < FLOAT_EPSILON) { r.t_max.x = FLT_MAX; r.t_delta.x = FLT_MAX; }
  if (fabsf(boundary.x - dir.x) < FLOAT_EPSILON) { r.t_max.x = FLT_MAX; r.t_delta.x = FLT_MAX; }
  if (fabsf(dir.y) < FLOAT_EPSILON) { r.t_max.y = FLT_MAX; r.t_delta.y = FLT_MAX; }
  if (fabsf(boundary.y - dir.y) < FLOAT_EPSILON) { r.t_max.y = FLT_MAX; r.t_delta.y = FLT_MAX; }
  if (fabsf(dir.z) < FLOAT_EPSILON) { r.t_max.z = FLT_MAX; r.t_delta.z = FLT_MAX; }
  if (fabsf(boundary.z - dir.z) < FLOAT_EPSILON) { r.t_max.z = FLT_MAX; r.t_delta.z = FLT_MAX; }
  return r;
}
/* one traversal step; returns 0 when the walk ends */
static inline int dda_step(dda3* r) {
  if (r->t_max.x < r->t_max.y && r->t_max.x < r->t_max.z) {
    r->cur.x = f2i((float) r->cur.x + r->step.x);
    if (r->cur.x == r->bound.x) return 0;
    r->t_max.x += r->t_delta.x;
  } else if (r->t_max.z < r->t_max.y) {
    r->cur.z = f2i((float) r->cur.z + r->step.z);
    if (r->cur.z == r->bound.z) return 0;
    r->t_max.z += r->t_delta.z;
  } else {
    r->cur.y = f2i((float) r->cur.y + r->step.y);
    if (r->cur.y == r->bound.y) return 0;
    r->t_max.y += r->t_delta.y;
  }
  return 1;
}

/* allocBlocks3DKernel for one point (vds.cu:925-1033).  Normals: one per point (the reference indexes a 3-vectors-per-point
 * eigenvector array by 3 * point, vds.cu:937; only that first vector, the normal, is used) */
static void alloc_point(mrh_ctx* c, uint64_t i) {
  const f3 pcam = mk3(c->points[3 * i], c->points[3 * i + 1], c->points[3 * i + 2]);
  const float range = norm3(pcam);
  if (range == 0.f) return;
  f3 cam_dir = normalize3(pcam);
  if (!c->p.projective_sdf) cam_dir = normalize3(mk3(c->normals[3 * i], c->normals[3 * i + 1], c->normals[3 * i + 2]));  /* vds.cu:959-962: norm_dir */
  const float t = get_truncation(range, c->p.sdf_truncation, c->p.sdf_truncation_scale);
  const float min_depth = fminf(c->max_integration_distance, range - t);
  const float max_depth = fminf(c->max_integration_distance, range + t);
  if (min_depth >= max_depth) return;
  const float a = min_depth - range, b = max_depth - range;
  const f3 pcam_min = mk3(pcam.x + cam_dir.x * a, pcam.y + cam_dir.y * a, pcam.z + cam_dir.z * a);
  const f3 pcam_max = mk3(pcam.x + cam_dir.x * b, pcam.y + cam_dir.y * b, pcam.z + cam_dir.z * b);
  dda3 r = dda_setup(c, se3_apply(c->R, c->t, pcam_min), se3_apply(c->R, c->t, pcam_max), 1);
  for (unsigned iter = 0; iter < MAX_DDA_ITERATION_COUNT; iter++) {
    if (owns_block(c, r.cur)) (void) alloc_block(c, r.cur, 0, 0);
    if (!dda_step(&r)) return;
  }
}
/* allocBlocks3D host loop (same retry shape as allocBlocks, vds.cu:1036-1092) */
static void alloc_blocks_3d(mrh_ctx* c) {
  int prev_free = heap_high_free(c) + heap_low_free(c);
  reset_mutex(c);
  if (c->p.sdf_var_threshold > 0.f && heap_low_free(c) < (int) c->low_blocks_to_allocate) allocate_memory_low(c);  /* vds.cu:1048-1054 */
  for (uint64_t i = 0; i < c->num_points; i++) alloc_point(c, i);
  for (;;) {
    reset_mutex(c);
    for (uint64_t i = 0; i < c->num_points; i++) alloc_point(c, i);
    const int cur_free = heap_high_free(c) + heap_low_free(c);
    if (prev_free == cur_free) break;
    prev_free = cur_free;
  }
}

/* integrate3DKernel for one point (vds.cu:1215-1379): projective or normal-direction SDF, fine and coarse entries */
static void integrate_point(mrh_ctx* c, uint64_t i) {
  const float vs = c->p.virtual_voxel_size;
  const float ext = (float) c->p.voxel_extents_scale;
  const f3 pcam = mk3(c->points[3 * i], c->points[3 * i + 1], c->points[3 * i + 2]);
  const float range = norm3(pcam);
  if (range < 1e-6 || range > c->max_integration_distance) return;
  const f3 cam_dir = normalize3(pcam);
  const int projective = c->p.projective_sdf != 0;
  f3 norm_dir = mk3(0.f, 0.f, 0.f);
  if (!projective) norm_dir = normalize3(mk3(c->normals[3 * i], c->normals[3 * i + 1], c->normals[3 * i + 2]));
  const float truncation = get_truncation(range, c->p.sdf_truncation, c->p.sdf_truncation_scale);
  const float min_depth = fminf(c->max_integration_distance, range - truncation);
  const float max_depth = fminf(c->max_integration_distance, range + truncation);
  if (min_depth >= max_depth) return;
  f3 pcam_min, pcam_max;
  if (projective) {
    pcam_min = mk3(pcam.x - cam_dir.x * truncation, pcam.y - cam_dir.y * truncation, pcam.z - cam_dir.z * truncation);
    pcam_max = mk3(pcam.x + cam_dir.x * truncation, pcam.y + cam_dir.y * truncation, pcam.z + cam_dir.z * truncation);
  } else {
    const float a = min_depth - range, b = max_depth - range;
    pcam_min = mk3(pcam.x + norm_dir.x * a, pcam.y + norm_dir.y * a, pcam.z + norm_dir.z * a);
    pcam_max = mk3(pcam.x + norm_dir.x * b, pcam.y + norm_dir.y * b, pcam.z + norm_dir.z * b);
  }
  dda3 r = dda_setup(c, se3_apply(c->R, c->t, pcam_min), se3_apply(c->R, c->t, pcam_max), 0);
  const uint8_t w1 = (uint8_t) (float) (uint8_t) c->p.integration_weight_sample;  /* weight_update = integration_weight_sample (uchar -> float -> uchar) */
  for (unsigned iter = 0; iter < MAX_DDA_ITERATION_COUNT; iter++) {
    const i3 block = voxel_to_block(r.cur, vs, ext);
    const HashEntry entry = get_hash_entry(c, block);
    if (entry.ptr != FREE_ENTRY) {
      const int scale = 1 << entry.resolution;
      const i3 aprox = {r.cur.x / scale, r.cur.y / scale, r.cur.z / scale};  /* C division: truncates toward zero (vds.cu:1306-1307) */
      const f3 voxel_pos = voxel_to_world(vs * (float) scale, aprox);       /* getVoxelSize(entry) = vs * (1 << resolution) */
      const f3 pc = se3_apply(c->Ri, c->ti, voxel_pos);
      float sdf;
      if (projective) sdf = range - norm3(pc);
      else sdf = ((pc.x - pcam.x) * norm_dir.x + (pc.y - pcam.y) * norm_dir.y) + (pc.z - pcam.z) * norm_dir.z;  /* dot, cuda_math.cuh */
      if (sdf <= -truncation) break;
      if (sdf >= 0.f) sdf = fminf(truncation, sdf);
      else sdf = fmaxf(-truncation, sdf);
      Voxel* dst = &c->blocks[(size_t) entry.ptr + voxel_to_block_index(r.cur, SDF_BLOCK_SIZE / scale)];  /* D1: dense index on coarse blocks */
      float curr_mean = 0.f;
      if (dst->weight > 0) curr_mean = dst->sdf;
      const float delta = (sdf - curr_mean) / (vs / 2);
      Voxel m;   /* combineVoxel(stored, curr = {sdf, weight_update, rgb 0}), vhu.cuh:167-181 */
      m.sum_squared = 0.f;
      m.rgb[0] = (uint8_t) f2i((0.5f * (float) dst->rgb[0] + 0.5f * 0.f) + 0.5f);
      m.rgb[1] = (uint8_t) f2i((0.5f * (float) dst->rgb[1] + 0.5f * 0.f) + 0.5f);
      m.rgb[2] = (uint8_t) f2i((0.5f * (float) dst->rgb[2] + 0.5f * 0.f) + 0.5f);
      m.sdf = (dst->sdf * (float) dst->weight + sdf * (float) w1) / (float) ((int) dst->weight + (int) w1);
      {
        const int wsum = (int) dst->weight + (int) w1;
        const int wmax = (int) (uint8_t) c->p.integration_weight_max;
        m.weight = (uint8_t) (wsum < wmax ? wsum : wmax);
      }
      *dst = m;
      const float delta2 = (sdf - dst->sdf) / (vs / 2);
      dst->sum_squared = dst->sum_squared + delta * delta2;
      c->total_updated++;
    }
    if (!dda_step(&r)) return;
  }
}

int mrh_upload_points(mrh_ctx* c, const float* xyz, uint64_t n) {
  if (!c || (n && !xyz)) return fail(c, MRH_ERR_INVALID_ARG, "mrh_upload_points: bad argument");
  free(c->points);
  c->points = n ? (float*) malloc((size_t) n * 3 * sizeof(float)) : NULL;
  if (n) memcpy(c->points, xyz, (size_t) n * 3 * sizeof(float));
  c->num_points = n;
  return MRH_OK;
}
int mrh_set_points_device(mrh_ctx* c, const float* xyz, uint64_t n) { return mrh_upload_points(c, xyz, n); }
/* a hint for the order in which the HIP path takes the beams; the restatement walks the points one by one in index order */
int mrh_set_scan_layout(mrh_ctx* c, int row_len) { (void) row_len; return c ? MRH_OK : MRH_ERR_INVALID_ARG; }
int mrh_detect_scan_layout(const float* xyz, uint64_t n) { (void) xyz; (void) n; return 0; }
int mrh_upload_normals(mrh_ctx* c, const float* nxyz, uint64_t n) {
  if (!c || (n && !nxyz)) return fail(c, MRH_ERR_INVALID_ARG, "mrh_upload_normals: bad argument");
  free(c->normals);
  c->normals = n ? (float*) malloc((size_t) n * 3 * sizeof(float)) : NULL;
  if (n) memcpy(c->normals, nxyz, (size_t) n * 3 * sizeof(float));
  c->num_normals = n;
  return MRH_OK;
}

/* VoxelContainer::integrate(point_cloud, normals, weights, camera, max_num_frames), voxel_data_structures.cpp:112-135 */
static int is_starve_frame(const mrh_ctx* c, int max_num_frames);
static void gc_tail(mrh_ctx* c);
int mrh_integrate_points(mrh_ctx* c, int n_frames_invalidate) {
  if (!c) return MRH_ERR_INVALID_ARG;
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_integrate_points: an exchange is pending (call mrh_integrate_resume)");
  if (c->n_halo) return fail(c, MRH_ERR_STATE, "mrh_integrate_points: halo blocks of other shards are present (call mrh_drop_blocks(MRH_DROP_HALO))");
  if (!c->has_camera) return fail(c, MRH_ERR_STATE, "mrh_integrate_points: set_camera has not been called");
  const int max_num_frames = n_frames_invalidate < 0 ? c->p.n_frames_invalidate_voxels : n_frames_invalidate;
  if (!c->p.projective_sdf && c->num_normals != c->num_points)
    return fail(c, MRH_ERR_STATE, "mrh_integrate_points: the normal-direction SDF needs one normal per point (mrh_upload_normals)");
  c->last_inserted = c->last_freed = 0;
  alloc_blocks_3d(c);
  flat_and_reduce(c, 0);
  for (uint64_t i = 0; i < c->num_points; i++) integrate_point(c, i);
  if (c->p.sdf_var_threshold > 0.f && c->frames > 0) {
    check_var_sdf(c);
    realloc_blocks(c);
    flat_and_reduce(c, 0);
    /* reintegrate3D launches integrate3DKernel, not reintegrate3DKernel (vds.cu:1561-1580): the whole scan a second time */
    for (uint64_t i = 0; i < c->num_points; i++) integrate_point(c, i);
  }
  if (is_starve_frame(c, max_num_frames)) {
    starve_pass(c, 0);
    if (c->p.shard_count > 1) { c->pending = 1; return MRH_PENDING_EXCHANGE; }
    starve_pass(c, 1);
    starve_pass(c, 2);
  }
  if (max_num_frames > 0) gc_tail(c);
  c->frames++;
  return MRH_OK;
}

/* voxel_data_structures.cpp:90-110 VoxelContainer::integrate (+ camera.cu:21-26) */
int mrh_integrate(mrh_ctx* c, int n_frames_invalidate) {
  if (!c) return MRH_ERR_INVALID_ARG;
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_integrate: an exchange is pending (call mrh_integrate_resume)");
  if (c->n_halo) return fail(c, MRH_ERR_STATE, "mrh_integrate: halo blocks of other shards are present (call mrh_drop_blocks(MRH_DROP_HALO))");
  if (!c->has_camera) return fail(c, MRH_ERR_STATE, "mrh_integrate: set_camera has not been called");
  if (!c->depth || !c->rgb) return fail(c, MRH_ERR_STATE, "mrh_integrate: depth and rgb images are required");
  if (c->depth_rows != (int) c->rows || c->depth_cols != (int) c->cols || c->rgb_rows != (int) c->rows || c->rgb_cols != (int) c->cols)
    return fail(c, MRH_ERR_INVALID_ARG, "mrh_integrate: image shape does not match the camera");
  const int max_num_frames = n_frames_invalidate < 0 ? c->p.n_frames_invalidate_voxels : n_frames_invalidate;
  c->last_inserted = c->last_freed = 0;
  compute_cloud(c);
  alloc_blocks(c);
  flat_and_reduce(c, 1);
  integrate_depth_map(c);
  c->total_compact += c->current_occupied;
  if (c->p.sdf_var_threshold > 0.f && c->frames > 0) {
    check_var_sdf(c);
    realloc_blocks(c);
    flat_and_reduce(c, 1);
    reintegrate_depth_map(c);
  }
  if (is_starve_frame(c, max_num_frames)) {
    starve_pass(c, 0);
    if (c->p.shard_count > 1) { c->pending = 1; return MRH_PENDING_EXCHANGE; }
    starve_pass(c, 1);
    starve_pass(c, 2);
  }
  if (max_num_frames > 0) gc_tail(c);
  c->frames++;
  return MRH_OK;
}

int mrh_integrate_resume(mrh_ctx* c) {
  if (!c) return MRH_ERR_INVALID_ARG;
  if (c->pending == 1) { starve_pass(c, 1); c->pending = 2; return MRH_PENDING_EXCHANGE; }
  if (c->pending == 2) {
    starve_pass(c, 2);
    gc_tail(c);
    c->frames++;
    c->pending = 0;
    return MRH_OK;
  }
  return fail(c, MRH_ERR_STATE, "mrh_integrate_resume: no exchange is pending");
}

int mrh_exchange_buffer(mrh_ctx* c, void** ptr, uint64_t* n, int* is_device) {
  if (!c || !ptr || !n) return MRH_ERR_INVALID_ARG;
  if (c->pending == 0) return fail(c, MRH_ERR_STATE, "mrh_exchange_buffer: no exchange is pending");
  const size_t npix = (size_t) c->rows * c->cols;
  *ptr = c->pending == 1 ? (void*) c->depth_buff : (void*) (c->depth_buff + npix);
  *n = npix;
  if (is_device) *is_device = 0;
  return MRH_OK;
}

int mrh_sync(mrh_ctx* c) {
  if (!c) return MRH_ERR_INVALID_ARG;
  if (c->error_flags & 1u) return fail(c, MRH_ERR_CAPACITY, "block pool exhausted");
  return MRH_OK;
}

/* geowrapper.cpp:157-161 (the running mesh starts empty; D10: colors_ too) ... :181 ... end of the chunk loop */
int mrh_mesh_merge_begin(mrh_ctx* c) {
  if (!c) return MRH_ERR_INVALID_ARG;
  free(c->V); free(c->C); free(c->F);
  c->V = c->C = NULL; c->F = NULL; c->nv = c->nf = 0;
  c->merge_on = 1;
  c->merge_total = 0;
  return MRH_OK;
}

int mrh_mesh_merge_end(mrh_ctx* c, uint64_t* out_total_triangles) {
  if (!c) return MRH_ERR_INVALID_ARG;
  if (!c->merge_on) return fail(c, MRH_ERR_STATE, "mrh_mesh_merge_end: no merge in progress");
  c->merge_on = 0;
  if (out_total_triangles) *out_total_triangles = c->merge_total;
  return MRH_OK;
}

int mrh_extract_triangles(mrh_ctx* c, const mrh_triangle** out, uint64_t* out_n) {
  if (!c || !out_n) return MRH_ERR_INVALID_ARG; /* out == NULL: count + mesh only (include/mrhash_hip.h) */
  extract_iso_surface(c);
  if (c->merge_on) {
    c->merge_total += c->ntris;
    if (c->ntris > 0) process_triangles(c); /* geowrapper.cpp:181-183 */
  } else {
    process_triangles(c);
  }
  if (out) *out = c->tris;
  *out_n = c->ntris;
  return (c->error_flags & 8u) ? fail(c, MRH_ERR_CAPACITY, "triangle buffer full") : MRH_OK;
}

int mrh_extract_mesh(mrh_ctx* c, const double** v, uint64_t* nv, const int32_t** f, uint64_t* nf, const double** col) {
  if (!c || !v || !nv || !f || !nf || !col) return MRH_ERR_INVALID_ARG;
  *v = c->V; *nv = c->nv; *f = c->F; *nf = c->nf; *col = c->C;
  return MRH_OK;
}

int mrh_get_stats(mrh_ctx* c, mrh_stats* s) {
  if (!c || !s) return MRH_ERR_INVALID_ARG;
  memset(s, 0, sizeof(*s));
  s->frames_integrated = c->frames;
  s->num_sdf_blocks = c->num_sdf_blocks;
  for (unsigned i = 0; i < c->total_size; i++)
    if (c->table[i].ptr != FREE_ENTRY) { if (c->table[i].resolution == 0) s->occupied_fine++; else s->occupied_coarse++; }
  s->free_fine = heap_high_free(c);
  s->free_coarse = heap_low_free(c);
  s->last_compact_blocks = c->current_occupied;
  s->last_updated_voxels = c->last_updated;
  s->last_inserted_blocks = c->last_inserted;
  s->last_freed_blocks = c->last_freed;
  s->total_updated_voxels = c->total_updated;
  s->total_compact_blocks = c->total_compact;
  s->last_triangles = c->ntris;
  s->error_flags = c->error_flags;
  return MRH_OK;
}

int mrh_set_profile(mrh_ctx* c, int enabled) { if (!c) return MRH_ERR_INVALID_ARG; c->profile = enabled; return MRH_OK; }

int mrh_dump_blocks(mrh_ctx* c, mrh_block_desc* descs, mrh_voxel* voxels, uint64_t capacity, uint64_t* out_n) {
  if (!c || !out_n) return MRH_ERR_INVALID_ARG;
  uint64_t n = 0;
  for (unsigned i = 0; i < c->total_size; i++) {
    const HashEntry* e = &c->table[i];
    if (e->ptr == FREE_ENTRY) continue;
    if (descs) {
      if (n >= capacity) return fail(c, MRH_ERR_CAPACITY, "mrh_dump_blocks: capacity too small");
      descs[n].x = e->pos.x; descs[n].y = e->pos.y; descs[n].z = e->pos.z; descs[n].resolution = e->resolution;
      if (voxels) {
        const int nv = num_voxels_of(e->resolution);
        memset(&voxels[n * 512], 0, 512 * sizeof(mrh_voxel));
        memcpy(&voxels[n * 512], &c->blocks[(size_t) e->ptr], (size_t) nv * sizeof(mrh_voxel));
      }
    }
    n++;
  }
  *out_n = n;
  return MRH_OK;
}

/* streamer.cu:11-60 integrateFromGlobalHashPass1Kernel (RESOLVE_COLLISION branch: entries are unlinked with
 * deleteHashEntryElement) + Pass2 (payload copy, slot cleared); canonical output order = block position */
static int cmp_desc_pos(const void* a, const void* b) {
  const mrh_block_desc* x = (const mrh_block_desc*) a;
  const mrh_block_desc* y = (const mrh_block_desc*) b;
  if (x->x != y->x) return x->x < y->x ? -1 : 1;
  if (x->y != y->y) return x->y < y->y ? -1 : 1;
  if (x->z != y->z) return x->z < y->z ? -1 : 1;
  return 0;
}
int mrh_stream_out(mrh_ctx* c, const float center[3], float radius, mrh_block_desc* descs, mrh_voxel* voxels, uint64_t capacity, uint64_t* out_n) {
  if (!c || !out_n || !center) return MRH_ERR_INVALID_ARG;
  const float vs = c->p.virtual_voxel_size;
  uint64_t n = 0;
  for (unsigned i = 0; i < c->total_size; i++) {
    const HashEntry* e = &c->table[i];
    if (e->ptr == FREE_ENTRY) continue;
    const f3 pw = voxel_to_world(vs, block_to_voxel(e->pos));
    const float dx = pw.x - center[0], dy = pw.y - center[1], dz = pw.z - center[2];
    const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
    if (radius >= 0.f && !(d >= radius)) continue;
    if (descs) {
      if (n >= capacity) return fail(c, MRH_ERR_CAPACITY, "mrh_stream_out: capacity too small");
      descs[n].x = e->pos.x; descs[n].y = e->pos.y; descs[n].z = e->pos.z; descs[n].resolution = e->resolution;
    }
    n++;
  }
  *out_n = n;
  if (!descs) return MRH_OK;
  qsort(descs, (size_t) n, sizeof(mrh_block_desc), cmp_desc_pos);
  for (uint64_t k = 0; k < n; k++) {
    const i3 pos = {descs[k].x, descs[k].y, descs[k].z};
    const HashEntry e = get_hash_entry(c, pos);
    const int nv = num_voxels_of(e.resolution);
    if (voxels) {
      memset(&voxels[k * 512], 0, 512 * sizeof(mrh_voxel));
      memcpy(&voxels[k * 512], &c->blocks[(size_t) e.ptr], (size_t) nv * sizeof(mrh_voxel));
    }
    reset_mutex(c);  /* canonical: every qualifying block leaves (the reference's kernel can lose a bucket race and leave one for the next pass) */
    if (delete_hash_entry_element(c, pos))
      for (int i = 0; i < nv; ++i) delete_voxel(&c->blocks[(size_t) e.ptr + i]);
  }
  return MRH_OK;
}

int mrh_get_free_blocks(mrh_ctx* c, int64_t* out_free_fine, int64_t* out_free_coarse) {
  if (!c) return MRH_ERR_INVALID_ARG;
  if (out_free_fine) *out_free_fine = heap_high_free(c);
  if (out_free_coarse) *out_free_coarse = heap_low_free(c);
  return MRH_OK;
}

/* ---- 3DGS splat seeds: src/gs/quad_tree.cu:6-223, gaussian_data_structures.cu:5-84 ------------------------ */

#define QT_THREADS 256          /* params.h:18 n_threads_subdivide: the block size fixes the summation order */
#define QT_MAX_NODES 1000000u   /* params.h:20-23 max_num_qtree_nodes == qtree_leaves_capacity */

/* CUDANode::computeError, quad_tree.cu:6-90: thread t sums the pixels t, t + 256, ... of the node (row-major inside
 * the node), the 256 partial sums are folded by a halving tree; the same again for the squared deviations. */
static void qt_tree_fold(float s[3][QT_THREADS]) {
  for (int stride = QT_THREADS / 2; stride > 0; stride >>= 1)
    for (int t = 0; t < stride; t++)
      for (int k = 0; k < 3; k++) s[k][t] += s[k][t + stride];
}

static float qtree_node_error(const mrh_ctx* c, mrh_qtree_leaf n) {
  float s[3][QT_THREADS];
  const int count = n.width * n.height;
  const size_t cols = (size_t) c->rgb_cols;
  for (int t = 0; t < QT_THREADS; t++) {
    float r_sum = 0.f, g_sum = 0.f, b_sum = 0.f;
    for (int idx = t; idx < count; idx += QT_THREADS) {
      const int x = n.x0 + idx % n.width, y = n.y0 + idx / n.width;
      const uint8_t* pix = c->rgb + ((size_t) y * cols + (size_t) x) * 3;
      r_sum += (float) pix[0]; g_sum += (float) pix[1]; b_sum += (float) pix[2];
    }
    s[0][t] = r_sum; s[1][t] = g_sum; s[2][t] = b_sum;
  }
  qt_tree_fold(s);
  const float r_mean = s[0][0] / count, g_mean = s[1][0] / count, b_mean = s[2][0] / count;
  for (int t = 0; t < QT_THREADS; t++) {
    float r_mse = 0.f, g_mse = 0.f, b_mse = 0.f;
    for (int idx = t; idx < count; idx += QT_THREADS) {
      const int x = n.x0 + idx % n.width, y = n.y0 + idx / n.width;
      const uint8_t* pix = c->rgb + ((size_t) y * cols + (size_t) x) * 3;
      const float r_diff = (float) pix[0] - r_mean, g_diff = (float) pix[1] - g_mean, b_diff = (float) pix[2] - b_mean;
      r_mse += r_diff * r_diff; g_mse += g_diff * g_diff; b_mse += b_diff * b_diff;
    }
    s[0][t] = r_mse; s[1][t] = g_mse; s[2][t] = b_mse;
  }
  qt_tree_fold(s);
  const float r_fin = s[0][0] / count, g_fin = s[1][0] / count, b_fin = s[2][0] / count;
  const float error = r_fin * 0.2989f + g_fin * 0.5870f + b_fin * 0.1140f;
  return error * (c->rgb_cols * c->rgb_rows) / 90000000.0f;
}

/* CUDAQTree::subdivide + subdivideKernel, quad_tree.cu:102-223, level by level (D7 order). */
static int qtree_subdivide(mrh_ctx* c, float threshold, int min_pixel_size) {
  const size_t npix = (size_t) c->rgb_rows * c->rgb_cols;
  mrh_qtree_leaf* in = (mrh_qtree_leaf*) malloc(npix * sizeof *in);
  mrh_qtree_leaf* out = (mrh_qtree_leaf*) malloc(npix * sizeof *out);
  free(c->qt_leaves);
  c->qt_leaves = (mrh_qtree_leaf*) malloc(npix * sizeof *c->qt_leaves); /* leaves tile the image: never more than pixels */
  c->n_qt_leaves = 0;
  size_t n_in = 1;
  in[0].x0 = 0; in[0].y0 = 0; in[0].width = c->rgb_cols; in[0].height = c->rgb_rows;
  int rc = MRH_OK;
  while (n_in > 0) {
    size_t n_out = 0;
    for (size_t i = 0; i < n_in; i++) {
      const mrh_qtree_leaf node = in[i];
      const float err = qtree_node_error(c, node);
      const int w1 = node.width / 2, w2 = node.width - w1, h1 = node.height / 2, h2 = node.height - h1;
      if (err <= threshold || w1 <= min_pixel_size || h1 <= min_pixel_size) { c->qt_leaves[c->n_qt_leaves++] = node; continue; }
      const mrh_qtree_leaf k0 = {node.x0, node.y0, w1, h1}, k1 = {node.x0, node.y0 + h1, w1, h2};
      const mrh_qtree_leaf k2 = {node.x0 + w1, node.y0, w2, h1}, k3 = {node.x0 + w1, node.y0 + h1, w2, h2};
      out[n_out++] = k0; out[n_out++] = k1; out[n_out++] = k2; out[n_out++] = k3;
    }
    if (n_out > QT_MAX_NODES) { rc = MRH_ERR_CAPACITY; break; }
    mrh_qtree_leaf* tmp = in; in = out; out = tmp;
    n_in = n_out;
  }
  free(in); free(out);
  if (rc == MRH_OK && c->n_qt_leaves > QT_MAX_NODES) rc = MRH_ERR_CAPACITY;
  return rc;
}

/* processNodesKernel, gaussian_data_structures.cu:5-56 */
static void process_nodes(mrh_ctx* c) {
  free(c->seeds);
  c->seeds = (mrh_splat_seed*) malloc((c->n_qt_leaves ? c->n_qt_leaves : 1) * sizeof *c->seeds);
  c->n_seeds = 0;
  for (uint64_t i = 0; i < c->n_qt_leaves; i++) {
    const mrh_qtree_leaf node = c->qt_leaves[i];
    const float p2x = (float) node.x0 + 0.5f * (float) node.width, p2y = (float) node.y0 + 0.5f * (float) node.height;
    const int px = f2i(p2x + 0.5f), py = f2i(p2y + 0.5f);
    if (px < 0 || py < 0 || px >= (int) c->cols || py >= (int) c->rows) continue;
    const float depth_value = c->depth[(size_t) py * c->depth_cols + px];
    if (depth_value < c->min_depth) continue;
    const f3 center = se3_apply(c->R, c->t, inverse_projection(c, (unsigned) py, (unsigned) px, depth_value));
    if (get_voxel_f(c, center, NULL).weight != 1) continue;
    const float half_w = 0.5f * (float) node.width, half_h = 0.5f * (float) node.height;
    const float scale = (depth_value * sqrtf(half_w * half_w + half_h * half_h)) / c->fx;
    if (scale <= 0.0f) continue; /* as written: a NaN scale passes */
    mrh_splat_seed* s = &c->seeds[c->n_seeds++];
    s->p[0] = center.x; s->p[1] = center.y; s->p[2] = center.z;
    s->scale = scale;
    const uint8_t* rgb = c->rgb + ((size_t) py * c->rgb_cols + px) * 3;
    s->rgb[0] = rgb[0]; s->rgb[1] = rgb[1]; s->rgb[2] = rgb[2]; s->pad = 0;
  }
}

int mrh_splat_seeds(mrh_ctx* c, float qtree_thresh, int qtree_min_pixel_size, const mrh_splat_seed** out, uint64_t* out_n) {
  if (!c || !out || !out_n) return fail(c, MRH_ERR_INVALID_ARG, "mrh_splat_seeds: null argument");
  if (!c->has_camera) return fail(c, MRH_ERR_STATE, "mrh_splat_seeds: set_camera has not been called");
  if (c->model != MRH_CAMERA_PINHOLE) return fail(c, MRH_ERR_UNSUPPORTED, "mrh_splat_seeds: pinhole camera only");
  if (qtree_min_pixel_size < 0 || qtree_thresh != qtree_thresh) return fail(c, MRH_ERR_INVALID_ARG, "mrh_splat_seeds: bad quad-tree parameter");
  if (!c->depth || !c->rgb) return fail(c, MRH_ERR_STATE, "mrh_splat_seeds: no depth / colour image");
  if (c->depth_rows != (int) c->rows || c->depth_cols != (int) c->cols || c->rgb_rows != (int) c->rows || c->rgb_cols != (int) c->cols)
    return fail(c, MRH_ERR_INVALID_ARG, "mrh_splat_seeds: image shape differs from the camera");
  if ((uint64_t) c->rows * c->cols > (1ull << 22)) return fail(c, MRH_ERR_CAPACITY, "mrh_splat_seeds: image above 2^22 pixels");
  const int rc = qtree_subdivide(c, qtree_thresh, qtree_min_pixel_size);
  if (rc) return fail(c, rc, "mrh_splat_seeds: quad-tree above the reference's node capacity");
  process_nodes(c);
  *out = c->seeds; *out_n = c->n_seeds;
  return MRH_OK;
}

int mrh_get_qtree_leaves(mrh_ctx* c, const mrh_qtree_leaf** out, uint64_t* out_n) {
  if (!c || !out || !out_n) return MRH_ERR_INVALID_ARG;
  *out = c->qt_leaves; *out_n = c->n_qt_leaves;
  return MRH_OK;
}

int mrh_peek_free_blocks(mrh_ctx* c, int64_t* out_free_fine, int64_t* out_free_coarse, uint64_t* out_frames_behind) {
  if (out_frames_behind) *out_frames_behind = 0; /* the oracle is synchronous */
  return mrh_get_free_blocks(c, out_free_fine, out_free_coarse);
}

int mrh_get_voxel(mrh_ctx* c, int32_t vx, int32_t vy, int32_t vz, mrh_voxel* out, int* found) {
  if (!c || !out) return MRH_ERR_INVALID_ARG;
  i3 v = {vx, vy, vz};
  const HashEntry e = get_hash_entry(c, voxel_to_block(v, c->p.virtual_voxel_size, (float) c->p.voxel_extents_scale));
  if (found) *found = e.ptr != FREE_ENTRY;
  *out = get_voxel_i(c, v, NULL);
  return MRH_OK;
}

/* room on the coarse free list for `need` coarse blocks, in allocateMemoryLow's portions (vds.cu:860-871): an import or a merge
 * into a context whose frames have not refilled the list yet (vds.cu:885-891 does it at the start of a frame) */
static void ensure_coarse_units(mrh_ctx* c, int need) {
  if (!(c->p.sdf_var_threshold > 0.f)) return;
  while (heap_low_free(c) < need && c->low_blocks_to_allocate > 0 && heap_high_free(c) > (int) c->low_blocks_to_allocate) allocate_memory_low(c);
}

int mrh_import_blocks(mrh_ctx* c, const mrh_block_desc* descs, const mrh_voxel* voxels, uint64_t n) {
  if (!c || (n && (!descs || !voxels))) return MRH_ERR_INVALID_ARG;
  {
    int need = 0;
    for (uint64_t k = 0; k < n; k++) need += descs[k].resolution != 0;
    ensure_coarse_units(c, need);
  }
  for (uint64_t k = 0; k < n; k++) {
    const i3 pos = {descs[k].x, descs[k].y, descs[k].z};
    HashEntry e = get_hash_entry(c, pos);
    if (e.ptr == FREE_ENTRY) {
      int prev_free = heap_high_free(c) + heap_low_free(c);
      for (;;) { /* allocBlock's retry protocol, vds.cu:901-921 */
        reset_mutex(c);
        (void) alloc_block(c, pos, descs[k].resolution, 0);
        const int cur = heap_high_free(c) + heap_low_free(c);
        if (cur == prev_free) break;
        prev_free = cur;
      }
      e = get_hash_entry(c, pos);
      if (e.ptr == FREE_ENTRY) return fail(c, MRH_ERR_CAPACITY, "mrh_import_blocks: could not insert block");
    }
    memcpy(&c->blocks[(size_t) e.ptr], &voxels[k * 512], (size_t) num_voxels_of(e.resolution) * sizeof(Voxel));
  }
  return MRH_OK;
}

/* ---- multi-GPU block exchange: no reference counterpart (the reference is single-GPU).  Same semantics as the HIP
 * library (include/mrhash_hip.h), host memory; MRH_UNPACK_MERGE restates combineVoxel (vhu.cuh:167-181). */
int mrh_peek_error_flags(mrh_ctx* c, uint32_t* out_new_flags) {
  if (!c || !out_new_flags) return MRH_ERR_INVALID_ARG;
  *out_new_flags = 0;
  return MRH_OK;
}

int mrh_set_sharding(mrh_ctx* c, int shard_rank, int shard_count, int shard_chunk_log2) {
  if (!c || shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count) return MRH_ERR_INVALID_ARG;
  c->p.shard_rank = shard_rank; c->p.shard_count = shard_count; c->p.shard_chunk_log2 = shard_chunk_log2;
  return MRH_OK;
}

static int owner_rank(const mrh_ctx* c, i3 b) {
  if (c->p.shard_count <= 1) return 0;
  const int sh = (c->p.shard_chunk_log2 > 0 && c->p.shard_chunk_log2 < 16) ? c->p.shard_chunk_log2 : 3;
  const uint32_t cx = (uint32_t) (b.x >> sh), cy = (uint32_t) (b.y >> sh), cz = (uint32_t) (b.z >> sh);
  const uint32_t h = (cx * P0) ^ (cy * P1) ^ (cz * P2);
  return (int) ((h ^ (h >> 15)) % (uint32_t) c->p.shard_count);
}

int mrh_pack_blocks(mrh_ctx* c, int mode, int rank_arg, const mrh_block_record** out_records, uint64_t* out_n, int* out_is_device_memory) {
  if (!c || !out_records || !out_n) return MRH_ERR_INVALID_ARG;
  if (out_is_device_memory) *out_is_device_memory = 0;
  const int sh = (c->p.shard_chunk_log2 > 0 && c->p.shard_chunk_log2 < 16) ? c->p.shard_chunk_log2 : 3;
  const int side = 1 << sh;
  uint64_t n = 0;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      if (n > c->pack_cap) {
        free(c->pack);
        c->pack = (mrh_block_record*) malloc((size_t) n * sizeof(mrh_block_record));
        if (!c->pack) { c->pack_cap = 0; return fail(c, MRH_ERR_CAPACITY, "mrh_pack_blocks: out of memory"); }
        c->pack_cap = n;
      }
      n = 0;
    }
    for (unsigned i = 0; i < c->total_size; i++) {
      const HashEntry* e = &c->table[i];
      if (e->ptr == FREE_ENTRY) continue;
      const int owner = owner_rank(c, e->pos);
      int keep;
      if (mode == MRH_PACK_HALO) {
        const int lx = e->pos.x & (side - 1), ly = e->pos.y & (side - 1), lz = e->pos.z & (side - 1);
        keep = owner == c->p.shard_rank && (lx == 0 || lx == side - 1 || ly == 0 || ly == side - 1 || lz == 0 || lz == side - 1);
      } else if (mode == MRH_PACK_OWNER) {
        keep = owner == rank_arg;
      } else {
        return fail(c, MRH_ERR_INVALID_ARG, "mrh_pack_blocks: bad mode");
      }
      if (!keep) continue;
      if (pass == 1) {
        mrh_block_record* r = &c->pack[n];
        r->desc.x = e->pos.x; r->desc.y = e->pos.y; r->desc.z = e->pos.z; r->desc.resolution = e->resolution;
        memset(r->voxels, 0, sizeof r->voxels);
        memcpy(r->voxels, &c->blocks[(size_t) e->ptr], (size_t) num_voxels_of(e->resolution) * sizeof(mrh_voxel));
      }
      n++;
    }
  }
  *out_n = n;
  *out_records = n ? c->pack : NULL;
  return MRH_OK;
}

static void drop_position(mrh_ctx* c, i3 pos);
int mrh_unpack_blocks(mrh_ctx* c, int mode, const mrh_block_record* records, uint64_t n, int is_device_memory, uint64_t* out_taken) {
  if (!c || (n && !records)) return MRH_ERR_INVALID_ARG;
  (void) is_device_memory; /* everything is host memory here */
  if (mode != MRH_UNPACK_HALO && mode != MRH_UNPACK_MERGE) return fail(c, MRH_ERR_INVALID_ARG, "mrh_unpack_blocks: bad mode");
  uint64_t taken = 0;
  {
    int need = 0;
    for (uint64_t k = 0; k < n; k++) need += records[k].desc.resolution != 0;
    ensure_coarse_units(c, need);
  }
  for (uint64_t k = 0; k < n; k++) {
    const mrh_block_record* r = &records[k];
    const i3 pos = {r->desc.x, r->desc.y, r->desc.z};
    if (mode == MRH_UNPACK_HALO) {
      int wanted = 0;
      if (!owns_block(c, pos))
        for (int j = 0; j < 27 && !wanted; j++) {
          const i3 q = {pos.x + (j % 3) - 1, pos.y + ((j / 3) % 3) - 1, pos.z + (j / 9) - 1};
          wanted = owns_block(c, q);
        }
      if (!wanted) continue;
    }
    HashEntry e = get_hash_entry(c, pos);
    /* One position at two resolutions (variance-adaptive sub-maps; no rule in the reference, which is single-GPU): the COARSE
     * side wins, the fine side's observations go the way reallocBlock sends them (vds.cu:627-755 frees the fine block and starts
     * the coarse one from the current frame: nothing of the fine payload survives a coarsening).  So a fine record onto a coarse
     * block is dropped, and a coarse record onto a fine block replaces it.  Independent of the order of the sub-maps. */
    if (mode == MRH_UNPACK_MERGE && e.ptr != FREE_ENTRY && e.resolution != r->desc.resolution) {
      if (e.resolution > r->desc.resolution) continue; /* map coarse, record fine */
      drop_position(c, pos);                           /* map fine, record coarse */
      e = get_hash_entry(c, pos);
    }
    const int fresh = e.ptr == FREE_ENTRY;
    if (fresh) {
      int prev_free = heap_high_free(c) + heap_low_free(c);
      for (;;) { /* allocBlock's retry protocol, vds.cu:901-921 */
        reset_mutex(c);
        (void) alloc_block(c, pos, r->desc.resolution, 0);
        const int cur = heap_high_free(c) + heap_low_free(c);
        if (cur == prev_free) break;
        prev_free = cur;
      }
      e = get_hash_entry(c, pos);
      if (e.ptr == FREE_ENTRY) return fail(c, MRH_ERR_CAPACITY, "mrh_unpack_blocks: could not insert block");
      if (mode == MRH_UNPACK_HALO) {
        if (c->n_halo == c->halo_cap) {
          c->halo_cap = c->halo_cap ? 2 * c->halo_cap : 1024;
          c->halo = (mrh_block_desc*) realloc(c->halo, (size_t) c->halo_cap * sizeof(mrh_block_desc));
        }
        c->halo[c->n_halo++] = r->desc;
      }
    } else if (e.resolution != r->desc.resolution) {
      return fail(c, MRH_ERR_CAPACITY, "mrh_unpack_blocks: block exists at another resolution");
    }
    const int nv = num_voxels_of(e.resolution);
    Voxel* dst = &c->blocks[(size_t) e.ptr];
    if (mode == MRH_UNPACK_HALO) {
      memcpy(dst, r->voxels, (size_t) nv * sizeof(Voxel));
    } else {
      for (int i = 0; i < nv; i++) {
        const Voxel v0 = dst[i], v1 = r->voxels[i];
        if (v1.weight == 0) continue;
        if (v0.weight == 0) { dst[i] = v1; continue; }
        Voxel out = v1; /* sum_squared: the later sub-map's term */
        for (int ch = 0; ch < 3; ch++) out.rgb[ch] = (uint8_t) ((0.5f * (float) v0.rgb[ch] + 0.5f * (float) v1.rgb[ch]) + 0.5f);
        out.sdf = (v0.sdf * (float) v0.weight + v1.sdf * (float) v1.weight) / (float) ((int) v0.weight + (int) v1.weight);
        const int wsum = (int) v0.weight + (int) v1.weight;
        const int wmax = c->p.integration_weight_max & 0xFF;
        out.weight = (uint8_t) (wsum < wmax ? wsum : wmax);
        dst[i] = out;
      }
    }
    taken++;
  }
  if (out_taken) *out_taken = taken;
  return MRH_OK;
}

static void drop_position(mrh_ctx* c, i3 pos) {
  const HashEntry e = get_hash_entry(c, pos);
  if (e.ptr == FREE_ENTRY) return;
  const int nv = num_voxels_of(e.resolution);
  reset_mutex(c);
  if (delete_hash_entry_element(c, pos))
    for (int i = 0; i < nv; ++i) delete_voxel(&c->blocks[(size_t) e.ptr + i]);
}

int mrh_drop_blocks(mrh_ctx* c, int mode, uint64_t* out_dropped) {
  if (!c) return MRH_ERR_INVALID_ARG;
  uint64_t n = 0;
  if (mode == MRH_DROP_HALO) {
    for (uint64_t k = 0; k < c->n_halo; k++) { const i3 pos = {c->halo[k].x, c->halo[k].y, c->halo[k].z}; drop_position(c, pos); n++; }
  } else if (mode == MRH_DROP_FOREIGN || mode == MRH_DROP_ALL) {
    uint64_t cap = 0;
    for (unsigned i = 0; i < c->total_size; i++) cap += c->table[i].ptr != FREE_ENTRY;
    i3* list = (i3*) malloc((size_t) (cap ? cap : 1) * sizeof(i3));
    for (unsigned i = 0; i < c->total_size; i++) {
      const HashEntry* e = &c->table[i];
      if (e->ptr == FREE_ENTRY) continue;
      if (mode == MRH_DROP_ALL || !owns_block(c, e->pos)) list[n++] = e->pos;
    }
    for (uint64_t k = 0; k < n; k++) drop_position(c, list[k]);
    free(list);
  } else {
    return fail(c, MRH_ERR_INVALID_ARG, "mrh_drop_blocks: bad mode");
  }
  c->n_halo = 0;
  if (out_dropped) *out_dropped = n;
  return MRH_OK;
}

int mrh_get_triangle_blocks(mrh_ctx* c, const mrh_block_desc** d, const uint32_t** n_per_block, uint64_t* n) {
  if (!c || !d || !n_per_block || !n) return MRH_ERR_INVALID_ARG;
  *d = c->tri_blocks; *n_per_block = c->tri_counts; *n = c->n_tri_blocks;
  return MRH_OK;
}

int mrh_get_triangles_device(mrh_ctx* c, const mrh_triangle** out, uint64_t* out_n, int* out_is_device_memory) {
  if (!c || !out || !out_n) return MRH_ERR_INVALID_ARG;
  *out = c->ntris ? c->tris : NULL;
  *out_n = c->ntris;
  if (out_is_device_memory) *out_is_device_memory = 0;
  return MRH_OK;
}

typedef struct { mrh_block_desc d; uint32_t count; uint64_t src; } TriRun;
static int cmp_run_pos(const void* a, const void* b) {
  const TriRun* x = (const TriRun*) a;
  const TriRun* y = (const TriRun*) b;
  if (x->d.x != y->d.x) return x->d.x < y->d.x ? -1 : 1;
  if (x->d.y != y->d.y) return x->d.y < y->d.y ? -1 : 1;
  if (x->d.z != y->d.z) return x->d.z < y->d.z ? -1 : 1;
  return x->src < y->src ? -1 : (x->src > y->src ? 1 : 0);
}
/* rank 0 of a sharded extraction (no reference counterpart): per-block runs -> canonical block order -> processTriangles */
int mrh_process_triangle_runs(mrh_ctx* c, const mrh_block_desc* descs, const uint32_t* counts, uint64_t n_blocks, const mrh_triangle* tris,
                              uint64_t n_tris, int is_device_memory) {
  if (!c || (n_blocks && (!descs || !counts)) || (n_tris && !tris)) return MRH_ERR_INVALID_ARG;
  (void) is_device_memory;
  TriRun* runs = (TriRun*) malloc((size_t) (n_blocks ? n_blocks : 1) * sizeof(TriRun));
  uint64_t total = 0;
  for (uint64_t i = 0; i < n_blocks; i++) { runs[i].d = descs[i]; runs[i].count = counts[i]; runs[i].src = total; total += counts[i]; }
  if (total != n_tris) { free(runs); return fail(c, MRH_ERR_INVALID_ARG, "mrh_process_triangle_runs: counts and triangles disagree"); }
  qsort(runs, (size_t) n_blocks, sizeof(TriRun), cmp_run_pos);
  mrh_triangle* merged = (mrh_triangle*) malloc((size_t) (n_tris ? n_tris : 1) * sizeof(mrh_triangle));
  free(c->tri_blocks); free(c->tri_counts);
  c->tri_blocks = (mrh_block_desc*) calloc(n_blocks ? n_blocks : 1, sizeof(mrh_block_desc));
  c->tri_counts = (uint32_t*) calloc(n_blocks ? n_blocks : 1, sizeof(uint32_t));
  c->n_tri_blocks = n_blocks;
  uint64_t dst = 0;
  for (uint64_t k = 0; k < n_blocks; k++) {
    memcpy(merged + dst, tris + runs[k].src, (size_t) runs[k].count * sizeof(mrh_triangle));
    c->tri_blocks[k] = runs[k].d;
    c->tri_counts[k] = runs[k].count;
    dst += runs[k].count;
  }
  free(runs);
  free(c->tris);
  c->tris = merged; c->ntris = n_tris; c->cap_tris = n_tris ? n_tris : 1;
  process_triangles(c);
  return MRH_OK;
}

int mrh_process_triangles(mrh_ctx* c, const mrh_triangle* tris, uint64_t n) {
  if (!c || (n && !tris)) return MRH_ERR_INVALID_ARG;
  mrh_triangle* copy = (mrh_triangle*) malloc((n ? n : 1) * sizeof(mrh_triangle));
  if (n) memcpy(copy, tris, n * sizeof(mrh_triangle));
  free(c->tris);
  c->tris = copy; c->ntris = n; c->cap_tris = n ? n : 1;
  process_triangles(c);
  return MRH_OK;
}

int mrh_selftest_division(mrh_ctx* c, uint64_t samples, uint64_t seed, uint64_t* out) {
  (void) c; (void) samples; (void) seed;
  if (out) *out = 0; /* the oracle divides with the C '/' operator: nothing to self-test */
  return MRH_OK;
}

/* test hook for include/mrh_softmath.h (tests/test_softmath.py): op 0 sin, 1 cos, 2 atan2(a, b), 3 asin */
float orc_softmath(int op, float a, float b) {
  float s, c;
  switch (op) {
    case 0: mrh_sincosf(a, &s, &c); return s;
    case 1: mrh_sincosf(a, &s, &c); return c;
    case 2: return mrh_atan2f(a, b);
    default: return mrh_asinf(a);
  }
}

/* threads the order-independent loops (integrate, GC identify) run on; 1 when built without OpenMP */
#ifdef _OPENMP
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
#else
int orc_num_threads(void) { return 1; }
#endif

/* ---- test-only introspection of the literal reference structures -------------------------- */
/* (lets tests restate tests/test_hash_utils.cu:306-526 — BufferInitialization / HeapSanityCheck) */
const void* orc_hash_table(const mrh_ctx* c, uint64_t* n) { *n = c->total_size; return c->table; }
const unsigned* orc_heap_high(const mrh_ctx* c, uint64_t* n, int* counter) { *n = c->num_sdf_blocks; *counter = c->heap_counter_high; return c->heap_high; }
const int* orc_bucket_mutex(const mrh_ctx* c, uint64_t* n) { *n = c->hash_num_buckets; return c->mutex; }
const void* orc_compact_table(const mrh_ctx* c, uint64_t* n) { *n = c->total_size; return c->compact; }
const int* orc_decisions(const mrh_ctx* c, uint64_t* n) { *n = c->total_size; return c->decision; }
/* camera helpers for restating tests/test_projections.cu */
void orc_inverse_projection(const mrh_ctx* c, unsigned row, unsigned col, float d, float out[3]) {
  const f3 p = inverse_projection(c, row, col, d); out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
int orc_project_point(const mrh_ctx* c, const float pc[3], int approx, int out[2]) {
  int r = 0, cc = 0;
  const int ok = project_point(c, mk3(pc[0], pc[1], pc[2]), approx, &r, &cc);
  out[0] = r; out[1] = cc;
  return ok;
}
/* zero every weight (test_hash_utils.cu:243-251 does this before garbageCollect) */
void orc_zero_all_weights(mrh_ctx* c) {
  for (size_t i = 0; i < (size_t) c->num_sdf_blocks * TOTAL_SDF_BLOCK_SIZE; i++) c->blocks[i].weight = 0;
}
