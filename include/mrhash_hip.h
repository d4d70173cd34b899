/*
 * mrhash_hip.h — C ABI of the MI355X-native voxel-hash TSDF fusion engine.
 *
 * This is the drop-in boundary between a C++/Python host (the GeoWrapper facade) and the
 * gfx950 HIP layer (libmrhash_hip.so).  Every entry point replaces one C++ member call the
 * reference host makes on its CUDA subsystems; the citation after each declaration names the
 * reference interface it stands in for (paths relative to the reference checkout,
 * mrhash/src/sdf/...).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types cross this boundary
 *   - every function returns an int status: 0 = MRH_OK, negative = mrh_status error code;
 *     a human-readable message for the last failure on a context is available through
 *     mrh_last_error()
 *   - one host thread per context; calls are ordered on the context's HIP stream.  Calls
 *     that hand data back to the host (mrh_stats, mrh_dump_blocks, mrh_extract_triangles,
 *     mrh_extract_mesh, mrh_sync) block until the stream has drained; mrh_integrate only
 *     enqueues work.
 *   - all inputs are copied (or consumed) before the call returns unless the name says
 *     `_device`, in which case the pointer is a device pointer that must stay valid until
 *     the next mrh_sync()/blocking call.
 *
 * The same ABI is implemented by the CPU oracle (oracle/mrh_oracle.c -> libmrh_oracle.so),
 * which exists only so that tests can drive both implementations through identical code.
 * The product library never links, loads or calls the oracle.
 */
#ifndef MRHASH_HIP_H
#define MRHASH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRH_ABI_VERSION 3

typedef enum mrh_status {
  MRH_OK                = 0,
  MRH_ERR_INVALID_ARG   = -1, /* null pointer, bad shape, bad enum                         */
  MRH_ERR_DEVICE        = -2, /* HIP runtime error (message carries hipGetErrorString)     */
  MRH_ERR_NO_DEVICE     = -3, /* no gfx950 device visible                                  */
  MRH_ERR_CAPACITY      = -4, /* block pool / hash table / triangle buffer exhausted       */
  MRH_ERR_STATE         = -5, /* call order violated (e.g. integrate before set_camera)    */
  MRH_ERR_UNSUPPORTED   = -6, /* feature outside this library's scope                      */
  MRH_ERR_OUT_OF_RANGE  = -7  /* block coordinate outside the packed-key range (+-2^20)    */
} mrh_status;

typedef enum mrh_camera_model {
  MRH_CAMERA_PINHOLE   = 0, /* camera.cuh:9 CameraModel::Pinhole   */
  MRH_CAMERA_SPHERICAL = 1  /* camera.cuh:9 CameraModel::Spherical */
} mrh_camera_model;

/* Construction parameters: the GeoWrapper constructor arguments that reach the fusion path
 * (geowrapper.cpp:9-81, pybind/pygeowrapper.cpp:14-29) plus explicit capacities (the
 * reference derives them from cudaMemGetInfo, geowrapper.cpp:37-54; 0 here means "apply the
 * same rule to the free HBM of the selected device"). */
typedef struct mrh_params {
  uint32_t abi_version;               /* must be MRH_ABI_VERSION                                         */
  float    sdf_truncation;            /* metres                                                          */
  float    sdf_truncation_scale;      /* truncation grows by scale * depth (vhu.cuh:184-187)             */
  int32_t  integration_weight_sample; /* per-observation weight (passed to kernels as u8, vds.cu:1101)   */
  int32_t  integration_weight_max;    /* weight clamp, params.h:24 = 255                                 */
  float    virtual_voxel_size;        /* metres, finest voxel                                            */
  int32_t  n_frames_invalidate_voxels;/* 0 = no GC; >0 = GC every frame, starve every n-th frame         */
  int32_t  voxel_extents_scale;       /* only 1 is coherent in the reference (vhu.cuh:90-92 vs :138-140) */
  float    marching_cubes_threshold;
  uint8_t  min_weight_threshold;
  uint8_t  projective_sdf;            /* kept for signature parity; RGB-D path is always projective      */
  uint8_t  reserved0[2];
  float    min_depth;                 /* initial camera thresholds (geowrapper.cpp:80)                   */
  float    max_depth;
  float    sdf_var_threshold;         /* >0 enables variance-adaptive fine->coarse blocks                */
  float    vertices_merging_threshold;
  uint64_t num_sdf_blocks;            /* capacity in fine (8^3) blocks; 0 = reference sizing rule        */
  uint64_t hash_slots;                /* open-address table slots (power of two); 0 = 4 x num_sdf_blocks */
  uint64_t max_triangles;             /* triangle buffer capacity; 0 = reference sizing rule             */
  int32_t  device_id;                 /* HIP device ordinal                                              */
  /* Multi-GPU tile sharding (new design, no reference counterpart): this context owns the
   * blocks whose chunk (cube of 2^shard_chunk_log2 blocks) hashes to shard_rank modulo shard_count.
   * shard_count 1 = own all.  Marching cubes then emits triangles for owned blocks only. */
  int32_t  shard_rank;
  int32_t  shard_count;
  int32_t  shard_chunk_log2;          /* ownership granularity: cubes of 2^k blocks per side; 0 = default 3     */
} mrh_params;

/* Reference `Voxel` (voxel_hash_utils.cuh:8-22): 12 bytes. Used only at the dump/restore
 * boundary; the device layout is private. */
typedef struct mrh_voxel {
  float   sdf;
  float   sum_squared;
  uint8_t rgb[3];
  uint8_t weight;
} mrh_voxel;

/* One allocated block as seen from outside: position in block units, resolution level
 * (0 = 8^3 voxels, 1 = 4^3 voxels at 2x spacing).  Mirrors the host-side `SDFBlockDesc`
 * (streamer.cuh:40-80) minus the private heap pointer. */
typedef struct mrh_block_desc {
  int32_t x, y, z;
  int32_t resolution;
} mrh_block_desc;

/* Reference `Vertex` / `Triangle` (voxel_hash_utils.cuh:46-64): 24 / 72 bytes. */
typedef struct mrh_vertex {
  float p[3];
  float c[3];
} mrh_vertex;

typedef struct mrh_triangle {
  mrh_vertex v[3];
} mrh_triangle;

typedef struct mrh_stats {
  uint64_t frames_integrated;     /* VoxelContainer::num_integrated_frames_                             */
  uint64_t num_sdf_blocks;        /* capacity                                                           */
  uint64_t occupied_fine;         /* live 8^3 blocks                                                    */
  uint64_t occupied_coarse;       /* live 4^3 blocks                                                    */
  int64_t  free_fine;             /* = getHeapHighFreeCount(), voxel_data_structures.cpp:148-153        */
  int64_t  free_coarse;           /* = getHeapLowFreeCount(),  voxel_data_structures.cpp:156-161        */
  uint64_t last_compact_blocks;   /* M: in-frustum blocks of the last frame (current_occupied_blocks_)  */
  uint64_t last_updated_voxels;   /* U: voxels written by the last integrate kernel (profile mode only) */
  uint64_t last_inserted_blocks;  /* blocks inserted by the last frame's allocation (profile mode only) */
  uint64_t last_freed_blocks;     /* blocks freed by the last frame's GC (profile mode only)            */
  uint64_t total_updated_voxels;  /* running sums of the two above since create/reset (profile mode)    */
  uint64_t total_compact_blocks;
  uint64_t last_triangles;        /* triangles produced by the last extraction                          */
  float    last_integrate_kernel_ms; /* HIP-event time of the last integrate kernel (profile mode)      */
  float    sum_integrate_kernel_ms;  /* running sum over frames since mrh_set_profile(ctx, 1)           */
  uint64_t n_integrate_kernel;       /* number of launches in that sum                                  */
  uint32_t error_flags;           /* device-side flags raised since create / reset: bit0 pool exhausted,
                                     bit1 table full, bit2 key out of range, bit3 triangle buffer full   */
  uint32_t reserved;
  /* open-address table upkeep (no reference counterpart: its buckets return slots to FREE, vds.cu:1727-1824) */
  uint64_t hash_slots;            /* table capacity                                                     */
  uint64_t tombstones;            /* erased slots not yet reused or rebuilt away (counted by this call) */
  uint32_t max_probe_length;      /* longest probe path of a live key, in slots (1 = home slot)         */
  uint32_t rehash_count;          /* table rebuilds since create / reset                                */
  /* marching cubes: HIP-event times of the two kernels of the last mrh_extract_triangles (0 if none ran) */
  float    last_mc_count_ms;
  float    last_mc_emit_ms;
  uint64_t last_mc_blocks;        /* blocks (fine + coarse) the last extraction walked                  */
  /* profile mode, two-launch fast path: HIP-event time of the allocation + sweep launch (k_front), as sum_integrate_kernel_ms
   * is that of the integrate launch (k_back)                                                                               */
  float    sum_front_kernel_ms;
  uint32_t reserved1;
  uint64_t n_front_kernel;
} mrh_stats;

typedef struct mrh_ctx mrh_ctx;

/* ---- lifetime ------------------------------------------------------------------------ */

/* Replaces GeoWrapper::GeoWrapper + VoxelContainer/MarchingCubesExtractor constructors
 * (geowrapper.cpp:9-81, voxel_data_structures.cuh:63-100, mesh_extractor.cuh:53-56). */
int mrh_create(const mrh_params* params, mrh_ctx** out_ctx);

/* Replaces the subsystem destructors (voxel_data_structures.cuh:117-170). NULL is a no-op. */
int mrh_destroy(mrh_ctx* ctx);

/* Replaces GeoWrapper::clearBuffers -> VoxelContainer::resetBuffers
 * (voxel_data_structures.cpp:58-87): empties table, pools and frame counter. */
int mrh_reset(mrh_ctx* ctx);

/* Message of the last failing call on this context ("" if none); never NULL.
 * With ctx == NULL returns the message of the last failing mrh_create on this thread.
 * Replaces CUDA_CHECK's print-and-exit (cuda_utils.cuh:9-17). */
const char* mrh_last_error(const mrh_ctx* ctx);

/* ---- per-frame inputs ------------------------------------------------------------------ */

/* Replaces GeoWrapper::setCamera -> Camera::Camera + setIntegrationDistance
 * (geowrapper.cpp:98-116, camera.cuh:13-40). */
int mrh_set_camera(mrh_ctx* ctx, float fx, float fy, float cx, float cy, int rows, int cols,
                   float min_depth, float max_depth, int camera_model);

/* Replaces Camera::setCamInWorld (camera.cuh:72). R is the row-major 3x3 rotation of the
 * camera in the world, t its translation (geowrapper.cpp:86-92 after toRotationMatrix). */
int mrh_set_pose(mrh_ctx* ctx, const float R_row_major[9], const float t[3]);

/* Replaces depth_img_.toDevice() (geowrapper.cpp:125): host float32 [rows, cols], row-major. */
int mrh_upload_depth(mrh_ctx* ctx, const float* depth, int rows, int cols);

/* Replaces rgb_img_.toDevice() (geowrapper.cpp:126): host uint8 [rows, cols, 3]. */
int mrh_upload_rgb(mrh_ctx* ctx, const uint8_t* rgb, int rows, int cols);

/* Both uploads return as soon as the image sits in pinned staging memory (the caller's buffer is free again); the
 * host-to-device copy runs on a copy stream of its image kind (depth and colour move side by side) into one of three
 * device slots, so the copies of frame N+1 overlap the kernels of frame N.  mrh_integrate / mrh_splat_seeds order themselves after the newest upload.
 *
 * Zero-copy variants: the images already live in HBM (e.g. a resident frame queue). The
 * pointers are used by the next mrh_integrate (and mrh_splat_seeds) and must stay valid until it has executed. */
int mrh_set_depth_device(mrh_ctx* ctx, const float* d_depth, int rows, int cols);
int mrh_set_rgb_device(mrh_ctx* ctx, const uint8_t* d_rgb, int rows, int cols);

/* ---- the hot path ---------------------------------------------------------------------- */

/* One frame of fusion = Camera::computeCloud + VoxelContainer::integrate
 * (camera.cu:21-26, voxel_data_structures.cpp:90-110): block allocation along every pixel
 * ray, frustum compaction, depth->TSDF integration, optional variance-driven coarsening,
 * optional starve + garbage collection.  n_frames_invalidate < 0 uses the constructor value.
 * Enqueues on the context stream and returns without waiting.  A frame whose images came through mrh_upload_* is checked
 * here (state, shapes) but its kernels are enqueued by the NEXT mrh_integrate — or by whichever other call needs the map
 * first —, with the pose and images it was issued under: its transfers have landed by then and its kernels need no
 * cross-stream wait.  A device error of such a frame is therefore returned one call late (MRH_DEFER_UPLOADS=0: at once). */
int mrh_integrate(mrh_ctx* ctx, int n_frames_invalidate);

/* Tile-sharded contexts (shard_count > 1) only.  On a starve frame (voxel_data_structures.cpp:139) the per-pixel
 * z-buffer of starveVoxelsKernel (vds.cu:1597-1649) must hold the minimum over the voxels of ALL shards, so the
 * frame is split at the two points where a reduction over shards is needed: mrh_integrate returns
 * MRH_PENDING_EXCHANGE (> 0, not an error) after filling the buffer; the host min-reduces the int64 buffer
 * returned by mrh_exchange_buffer across ranks (RCCL all-reduce MIN over xGMI; every value is < 2^63) and calls
 * mrh_integrate_resume, which may return MRH_PENDING_EXCHANGE once more before it returns MRH_OK.
 * Unsharded contexts never return MRH_PENDING_EXCHANGE. */
#define MRH_PENDING_EXCHANGE 1
int mrh_exchange_buffer(mrh_ctx* ctx, void** out_ptr, uint64_t* out_count_int64, int* out_is_device_memory);
int mrh_integrate_resume(mrh_ctx* ctx);

/* ---- LiDAR scans (BASELINE.json configs[4]) --------------------------------------------------
 * Replaces GeoWrapper::setPointCloud (geowrapper.cpp:345-405, pybind/pygeowrapper.cpp:66-76: the point matrix is
 * copied) and VoxelContainer::integrate(point_cloud, normals, weights, camera, max_num_frames)
 * (voxel_data_structures.cpp:112-135) = allocBlocks3D (vds.cu:925-1092) + integrate3D (vds.cu:1215-1410).
 * xyz: n points, sensor frame, float32 [n][3]; the pose is the one given to mrh_set_pose, the integration distance
 * the max_depth given to mrh_set_camera (either camera model; no image is involved).
 * Covers the projective SDF (every shipped LiDAR configuration) and the normal-direction SDF (mrh_upload_normals),
 * garbage collection on scans (n_frames_invalidate_voxels > 0: identify + free over every live block each scan; the
 * starve step every n-th scan projects through the camera given to mrh_set_camera — spherical for a LiDAR) and
 * variance-adaptive maps (sdf_var_threshold > 0: after coarsening the scan is integrated a second time, as
 * reintegrate3D does, vds.cu:1561-1580).
 * The reference updates a voxel with a non-atomic read-modify-write per point (a race between the points of a
 * scan); here every voxel receives its updates in ascending point index (oracle header, D6). */
int mrh_upload_points(mrh_ctx* ctx, const float* xyz, uint64_t n);
/* One normal per point of the current scan (sensor frame, any length: normalised on the device as vds.cu:1236 does), for
 * the normal-direction SDF (projective_sdf = 0, vds.cu:1248-1251, :1322-1326).  Replaces the first eigenvector the
 * reference takes from its MAD-tree (geowrapper.cpp:386-403: three eigenvectors per point, indexed by 3 * point,
 * vds.cu:1229); estimating normals is the caller's business here.  Copied before the call returns. */
int mrh_upload_normals(mrh_ctx* ctx, const float* nxyz, uint64_t n);
int mrh_set_points_device(mrh_ctx* ctx, const float* d_xyz, uint64_t n); /* zero-copy: device pointer, valid until the next integrate returns */
int mrh_integrate_points(mrh_ctx* ctx, int n_frames_invalidate);
/* How the caller's scans are laid out (no counterpart in the reference: setPointCloud takes an unordered matrix).  A LiDAR
 * driver usually delivers an ORGANISED cloud — rows of `row_len` points, row-major, neighbours in the array neighbours in
 * direction — and beams that leave side by side end in the same voxels: the integration then takes its beams in 16 x 16
 * patches instead of 256 in a row (3-4 x fewer distinct voxels a workgroup: 96 -> 88 us per 128 x 1024 scan at the time it was built).  row_len > 0: the caller's
 * scans have that many points per row; 0 (the default): clouds given as host memory (mrh_upload_points) are looked at —
 * a few dozen point pairs, when the cloud's size changes and every 64th upload —, clouds given as device memory are taken as unordered; < 0: never.  A performance hint only:
 * the map is the same bit for bit whatever is said here (every voxel still receives its updates in ascending point index). */
int mrh_set_scan_layout(mrh_ctx* ctx, int row_len);
/* Pure host code, no context: the row length mrh_upload_points would find for this cloud (0: not an organised scan). */
int mrh_detect_scan_layout(const float* xyz, uint64_t n);

/* ---- 3DGS splat seeds (BASELINE.json configs[4]) ----------------------------------------------
 * Replaces the initialisation half of GaussianContainer::runGS (gaussian_data_structures.cpp:140-157):
 * extractNodesQTree = CUDAQTree::subdivide (gaussian_data_structures.cpp:48-68, src/gs/quad_tree.cu:6-223) — a
 * quad-tree over the CURRENT colour image, a node is a leaf when its luma-weighted colour MSE x (rows*cols / 9e7)
 * is <= qtree_thresh or a half of it would be <= qtree_min_pixel_size pixels wide/high — and checkNodes =
 * processNodesKernel (gaussian_data_structures.cu:5-84): a leaf whose centre pixel has depth >= min_depth and whose
 * back-projected centre falls into a voxel of weight exactly 1 (surface seen for the first time) yields one seed
 * {world position, scale = depth * |half extent| / fx, colour of the centre pixel}.  Call after mrh_integrate of the
 * same frame (the images and pose of that frame are still the current ones).  Pinhole camera only.
 * The reference appends leaves and seeds through atomic counters (a race order); canonical order here: leaves by
 * tree level, then by position in the tree (children in the order the reference writes them: top-left, bottom-left,
 * top-right, bottom-right); seeds in leaf order.  The optimiser / rasteriser (src/gs) stay out of scope: the seeds
 * are what GaussianModel::Add_gaussians (src/gs/gaussian.cu:147) is handed.
 * Buffers are owned by ctx until the next call.  Blocks.  Images above 2^22 pixels or trees above the reference's
 * 1 000 000-leaf capacity (params.h:20-23) return MRH_ERR_CAPACITY. */
typedef struct mrh_splat_seed {
  float p[3];      /* gs::Point, world frame (gaussian_utils.cuh:112-116)  */
  float scale;     /* d_scales_                                             */
  uint8_t rgb[3];  /* gs::Color (gaussian_utils.cuh:118-122)               */
  uint8_t pad;
} mrh_splat_seed;  /* 20 bytes */

typedef struct mrh_qtree_leaf {
  int32_t x0, y0, width, height; /* gs::CUDANode (quad_tree.cuh:10-50) without the racy id */
} mrh_qtree_leaf;

int mrh_splat_seeds(mrh_ctx* ctx, float qtree_thresh, int qtree_min_pixel_size, const mrh_splat_seed** out_seeds, uint64_t* out_n);
/* The leaves of the last mrh_splat_seeds call (CUDAQTree::getAllNodes, quad_tree.cuh:82-87).  Test / debug helper: the leaves
 * stay on the device until this is called (blocks for the read-back). */
int mrh_get_qtree_leaves(mrh_ctx* ctx, const mrh_qtree_leaf** out_leaves, uint64_t* out_n);

/* Blocks until every enqueued frame has executed.  Device error flags raised since the last call that reported them
 * (pool exhausted, table full, key out of range) come back as MRH_ERR_CAPACITY / MRH_ERR_OUT_OF_RANGE ONCE and are
 * cleared: the frames that raised them skipped the affected blocks (what the reference does after its device printf,
 * vds.cu:566-569) and the map stays usable.  mrh_get_stats().error_flags keeps the union since create / reset. */
int mrh_sync(mrh_ctx* ctx);

/* The same flags WITHOUT waiting for the device, for a host loop that never syncs (GeoWrapper::compute): every frame
 * ends with a small copy of the counters into pinned host memory (the mechanism of mrh_peek_free_blocks); this returns
 * the flags of the newest report that has arrived and that no earlier peek has returned (0: nothing new).  Does not
 * clear anything on the device: a later mrh_sync still reports them. */
int mrh_peek_error_flags(mrh_ctx* ctx, uint32_t* out_new_flags);

/* Replaces MeshExtractor::extractMesh = flatAndReduceHashTable() + extractIsoSurface
 * (mesh_extractor.cpp:95-98, marching_cubes.cu:264-305): marching cubes over every live
 * block.  Triangles come back in canonical order (block position ascending in (x,y,z), then
 * voxel index, then triangle number); the buffer is owned by ctx until the next extraction.
 * out_triangles may be NULL: the triangle soup then stays on the device (72 bytes per triangle that never cross the
 * link) and only *out_n and the mesh behind mrh_extract_mesh are produced — what GeoWrapper::extractMesh needs. */
int mrh_extract_triangles(mrh_ctx* ctx, const mrh_triangle** out_triangles, uint64_t* out_n);

/* Replaces MeshExtractor::processTriangles (mesh_extractor.cpp:9-76) applied to the triangles
 * of the last mrh_extract_triangles: f64 vertices [V,3], i32 faces [F,3], f64 colours [V,3],
 * after vertex merge (exact, or quantised by vertices_merging_threshold), degenerate- and
 * duplicate-face removal.  Buffers are owned by ctx until the next extraction. */
int mrh_extract_mesh(mrh_ctx* ctx, const double** out_vertices, uint64_t* out_nv,
                     const int32_t** out_faces, uint64_t* out_nf, const double** out_colors);

/* Replaces MeshExtractor::merge_mesh_ = true + the running vertices_ / faces_ / colors_ of the chunk loop in
 * GeoWrapper::extractMesh (geowrapper.cpp:157-188; mesh_extractor.cpp:27-40 `combine`): between begin and end every
 * mrh_extract_triangles ADDS its triangles to the running mesh (processTriangles on running mesh + new soup) instead of
 * replacing it; an extraction without triangles leaves it alone (geowrapper.cpp:181).  After mrh_mesh_merge_end the merged
 * mesh is read with mrh_extract_mesh; *out_total_triangles = triangles of all extractions in between (repeats included).
 * begin also empties colors_ (the reference only empties vertices_ and faces_, geowrapper.cpp:157-158: a second extractMesh
 * call would pair stale colours with new vertices — canonical choice D10: start from an empty mesh). */
int mrh_mesh_merge_begin(mrh_ctx* ctx);
int mrh_mesh_merge_end(mrh_ctx* ctx, uint64_t* out_total_triangles);

/* ---- introspection --------------------------------------------------------------------- */

/* Replaces the scalar read-backs getHeapHighFreeCount / getHeapLowFreeCount /
 * current_occupied_blocks_ (voxel_data_structures.cpp:148-161, vds.cu:447). Blocks. */
int mrh_get_stats(mrh_ctx* ctx, mrh_stats* out);

/* The two scalar read-backs of the per-frame streaming test (geowrapper.cpp:137: getHeapHighFreeCount() <=
 * stream_threshold * num_sdf_blocks) without the full statistics pass.  Blocks. */
int mrh_get_free_blocks(mrh_ctx* ctx, int64_t* out_free_fine, int64_t* out_free_coarse);

/* The same two numbers WITHOUT waiting for the device: from the first call on, every frame ends with a two-word copy
 * of the free-list levels into pinned host memory; the call returns the newest report that has arrived and how many
 * frames it is behind (0 = the last enqueued frame has already finished).  The very first call, and a call with no
 * report in flight, answer through mrh_get_free_blocks.  Lets a host keep the reference's per-frame paging test
 * (geowrapper.cpp:137) without serialising upload and compute on it.  A host-fed frame that mrh_integrate has kept back
 * for one call (see mrh_integrate) is counted in *out_frames_behind; its pool level and error flags appear only after the
 * call that launches it (the next mrh_integrate, mrh_sync, or any call that reads the map): a loop that ONLY peeks after its
 * last frame never sees that frame — end such a loop with mrh_sync. */
int mrh_peek_free_blocks(mrh_ctx* ctx, int64_t* out_free_fine, int64_t* out_free_coarse, uint64_t* out_frames_behind);

/* 1 = bracket the integrate kernel with HIP events and count updated voxels / inserted /
 * freed blocks on the device (used by bench.py for the roofline figures); 0 = off. */
int mrh_set_profile(mrh_ctx* ctx, int enabled);

/* Streamer, device half: Streamer::streamOutToHostPass0 / streamAllOut (streamer.cpp:168-205,
 * :232-281; integrateFromGlobalHashPass1/2Kernel streamer.cu:11-160).  Every live block whose origin
 * (block position * 8 * voxel size) lies at distance >= radius from `center` is copied out — descs[i] and 512
 * reference-layout voxels at voxels[i*512 ...] (coarse blocks use the first 64), ordered by block position — and
 * removed from the device map (entry deleted, slot zeroed and returned to its free list).  radius < 0: every block
 * (streamAllOut).  With descs == NULL nothing is removed and only the count of blocks that WOULD leave is returned.
 * The host keeps the blocks (GeoWrapper's chunk grid) and brings them back with mrh_import_blocks
 * (Streamer::streamInToGPU, streamer.cpp:358-378). */
int mrh_stream_out(mrh_ctx* ctx, const float center[3], float radius, mrh_block_desc* descs, mrh_voxel* voxels,
                   uint64_t capacity, uint64_t* out_n);

/* Copies every live block out: descs[i] and 512 reference-layout voxels at
 * voxels[i*512 .. i*512+511] (coarse blocks use the first 64).  With descs == NULL only the
 * count is returned.  Replaces the device->host half of Streamer::streamAllOut
 * (streamer.cpp:250-281) as far as tests and checkpointing need it.  Order is unspecified. */
int mrh_dump_blocks(mrh_ctx* ctx, mrh_block_desc* descs, mrh_voxel* voxels, uint64_t capacity,
                    uint64_t* out_n);

/* Inserts n blocks given in the mrh_dump_blocks format (descs[i], 512 reference-layout voxels each; coarse blocks
 * use the first 64).  A block that already exists is overwritten.  Used to restore a dumped map and to bring
 * boundary ("halo") blocks of neighbouring shards in before mesh extraction.  Replaces the host->device half
 * of the streamer (Streamer::streamInToGPU, streamer.cpp:358-378) as far as this library needs it.  Blocks. */
int mrh_import_blocks(mrh_ctx* ctx, const mrh_block_desc* descs, const mrh_voxel* voxels, uint64_t n);

/* Per-block triangle counts of the last mrh_extract_triangles, in the same canonical block order as the
 * triangle buffer (blocks with zero triangles included).  Lets sharded ranks merge their buffers into the
 * single-GPU order.  Buffers are owned by ctx until the next extraction. */
int mrh_get_triangle_blocks(mrh_ctx* ctx, const mrh_block_desc** out_descs, const uint32_t** out_counts, uint64_t* out_n);

/* The triangle soup of the last mrh_extract_triangles / mrh_process_triangle_runs where the library keeps it (device memory
 * for the HIP library: *out_is_device_memory = 1), valid until the next extraction: what a rank hands to a collective. */
int mrh_get_triangles_device(mrh_ctx* ctx, const mrh_triangle** out_triangles, uint64_t* out_n, int* out_is_device_memory);

/* Rank 0 of a sharded extraction: per-block runs of triangles from all ranks — descs[i] / counts[i] in host memory (a few
 * bytes per block), the triangles of run i following those of run i - 1 in `triangles` (device memory iff
 * is_device_memory, e.g. the output of an all-gather) — are brought into the canonical single-GPU order (block position)
 * on the device and post-processed (MeshExtractor::processTriangles); results through mrh_extract_mesh,
 * mrh_get_triangle_blocks and mrh_get_triangles_device. */
int mrh_process_triangle_runs(mrh_ctx* ctx, const mrh_block_desc* descs, const uint32_t* counts, uint64_t n_blocks,
                              const mrh_triangle* triangles, uint64_t n_triangles, int is_device_memory);

/* MeshExtractor::processTriangles (mesh_extractor.cpp:9-76) on a caller-supplied triangle buffer; the result is
 * read back with mrh_extract_mesh.  Used by rank 0 on the merged buffer of all shards. */
int mrh_process_triangles(mrh_ctx* ctx, const mrh_triangle* triangles, uint64_t n);

/* ---- multi-GPU: block exchange between tile-sharded contexts (new design, no reference counterpart) --------------
 * One process per GPU; the collectives themselves (RCCL all-gather / all-to-all over xGMI) are the host's business
 * (mrhash_amd/parallel.py), these calls produce and consume the buffers they move — in DEVICE memory, so that nothing
 * is staged through the host.  A record is one block in the reference layout: */
typedef struct mrh_block_record {
  mrh_block_desc desc;
  mrh_voxel      voxels[512];   /* coarse blocks use the first 64 */
} mrh_block_record;             /* 6160 bytes */

/* Changes the tile ownership of a context (mrh_params.shard_rank / shard_count / shard_chunk_log2) after creation:
 * a frame-sharded rank fuses its frames owning everything (count 1), then takes its place in the tile partition for
 * the merge and the mesh extraction. */
int mrh_set_sharding(mrh_ctx* ctx, int shard_rank, int shard_count, int shard_chunk_log2);

typedef enum mrh_pack_mode {
  MRH_PACK_HALO  = 0, /* blocks this rank owns that lie on the surface of their chunk: what a neighbouring chunk's
                         marching cubes can read (corner samples reach <= 2 voxels into the next block)          */
  MRH_PACK_OWNER = 1  /* every live block owned by `rank_arg` (frame-sharded sub-maps on their way to the owner) */
} mrh_pack_mode;
/* Selects blocks, orders them by position and writes their records into a buffer owned by ctx (valid until the next
 * pack call).  *out_is_device_memory = 1 for the HIP library.  The map is not modified.  Blocks. */
int mrh_pack_blocks(mrh_ctx* ctx, int mode, int rank_arg, const mrh_block_record** out_records, uint64_t* out_n,
                    int* out_is_device_memory);

typedef enum mrh_unpack_mode {
  MRH_UNPACK_HALO  = 0, /* keep the records of blocks this rank does NOT own that are 26-adjacent to a block position it
                           owns; they become ordinary readable blocks, remembered as halo (mrh_drop_blocks)     */
  MRH_UNPACK_MERGE = 1  /* weighted merge into the map, voxel by voxel, with combineVoxel's arithmetic
                           (vhu.cuh:167-181): sdf = (s0 w0 + s1 w1) / (w0 + w1), weight = min(max, w0 + w1),
                           colour = u8(0.5 c0 + 0.5 c1 + 0.5); a voxel with weight 0 on one side takes the other
                           side unchanged; absent blocks are inserted.  Variance-adaptive maps (round 6): a position
                           that is fine on one side and coarse on the other ends up COARSE — the coarse side wins, the fine
                           side's observations are dropped, as reallocBlock drops them when it coarsens a block
                           (vds.cu:627-755); the result does not depend on the order of the sub-maps.  *out_taken counts
                           the records that were merged or inserted (a fine record onto a coarse block is not).           */
} mrh_unpack_mode;
/* `records` is a device pointer iff is_device_memory != 0 (what a RCCL collective leaves behind).  Blocks.
 * The records of ONE call must carry distinct block positions (the blocks of one rank's map do): every record is handled by
 * its own workgroup, and two records of the same position in one MRH_UNPACK_MERGE call would race on that block.  Fold
 * several sub-maps with one call per sub-map (mrh_comm_merge_submaps, mrhash_amd/parallel.py). */
int mrh_unpack_blocks(mrh_ctx* ctx, int mode, const mrh_block_record* records, uint64_t n, int is_device_memory,
                      uint64_t* out_taken);

typedef enum mrh_drop_mode {
  MRH_DROP_HALO    = 0, /* the blocks MRH_UNPACK_HALO brought in (after the extraction that needed them)  */
  MRH_DROP_FOREIGN = 1, /* every block this rank does not own                                               */
  MRH_DROP_ALL     = 2  /* every block (the frame counter and the configuration stay)                       */
} mrh_drop_mode;
int mrh_drop_blocks(mrh_ctx* ctx, int mode, uint64_t* out_dropped);

/* Looks one voxel up by integer voxel coordinate = VoxelContainer::getVoxel(int3)
 * (vds.cu:163-176); a miss returns a zero voxel and *out_found = 0. Test helper. */
int mrh_get_voxel(mrh_ctx* ctx, int32_t vx, int32_t vy, int32_t vz, mrh_voxel* out, int* out_found);

/* Device self-test of the arithmetic spec: the shared-reciprocal division the integrate kernel uses must be
 * bit-identical to a correctly rounded IEEE fp32 divide.  Runs `samples` pseudo-random operand pairs on the
 * context's device and returns the number of mismatches in *out_mismatches (0 expected). */
int mrh_selftest_division(mrh_ctx* ctx, uint64_t samples, uint64_t seed, uint64_t* out_mismatches);

/* Library build info: "mrhash_hip <abi> gfx950 ..." (or "mrh_oracle ..." for the oracle). */
const char* mrh_version(void);

#ifdef __cplusplus
}
#endif

#endif /* MRHASH_HIP_H */
