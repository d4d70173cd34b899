// geowrapper.cpp — host facade over the C ABI; mirrors pygeowrapper::GeoWrapper (geowrapper.cpp:9-577 of the
// reference) for the setDepthImage / compute() / extractMesh() path.  Errors that the reference turns into
// print-and-exit (cuda_utils.cuh:9-17) are raised as std::runtime_error instead.
#include "geowrapper.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <charconv>
#include <climits>
#include <chrono>
#include <memory>
#include <stdexcept>
#include <thread>

namespace pygeowrapper {

void GeoWrapper::check(int rc, const char* what) {
  if (rc != MRH_OK) throw std::runtime_error(std::string("GeoWrapper::") + what + " | " + mrh_last_error(ctx_));
}

GeoWrapper::GeoWrapper(float sdf_truncation, float sdf_truncation_scale, int integration_weight_sample, float virtual_voxel_size,
                       int n_frames_invalidate_voxels, int voxel_extents_scale, bool /*viewer_active*/, float marching_cubes_threshold,
                       uint8_t min_weight_threshold, float min_depth, float max_depth, const std::string& gs_optimization_param_path,
                       float sdf_var_threshold, float vertices_merging_threshold, bool projective_sdf)
    : sdf_truncation_(sdf_truncation), sdf_truncation_scale_(sdf_truncation_scale), integration_weight_sample_(integration_weight_sample),
      virtual_voxel_size_(virtual_voxel_size), n_frames_invalidate_voxels_(n_frames_invalidate_voxels), voxel_extents_scale_(voxel_extents_scale),
      min_weight_threshold_(min_weight_threshold), sdf_var_threshold_(sdf_var_threshold), vertices_merging_threshold_(vertices_merging_threshold) {
  if (!gs_optimization_param_path.empty()) {
    // geowrapper.cpp:74-78 builds a GaussianContainer from this file.  Here only its initialisation half exists (the
    // per-frame splat seeds, mrh_splat_seeds); the optimiser / rasteriser stay out of scope, so of the JSON only the two
    // quad-tree keys are read (src/gs/gaussian.cu:49-50; defaults gaussian.cuh:28-29).
    std::ifstream js(gs_optimization_param_path);
    if (!js.is_open()) throw std::runtime_error("GeoWrapper::GeoWrapper | cannot open " + gs_optimization_param_path);
    const std::string text((std::istreambuf_iterator<char>(js)), std::istreambuf_iterator<char>());
    auto number_after = [&](const char* key, double fallback) {
      const size_t k = text.find(std::string("\"") + key + "\"");
      if (k == std::string::npos) return fallback;
      const size_t colon = text.find(':', k);
      if (colon == std::string::npos) return fallback;
      return std::strtod(text.c_str() + colon + 1, nullptr);
    };
    qtree_thresh_ = (float) number_after("qtree_thresh", 0.1);
    qtree_min_pixel_size_ = (int) number_after("qtree_min_pixel_size", 1.0);
    gs_enabled_ = true;
    std::cerr << "GeoWrapper::GeoWrapper | splat seeds only (quad-tree threshold " << qtree_thresh_ << ", min pixel size " << qtree_min_pixel_size_
              << "); Gaussian-splatting optimisation is outside this library's scope" << std::endl;
  }
  mrh_params p;
  std::memset(&p, 0, sizeof p);
  p.abi_version = MRH_ABI_VERSION;
  p.sdf_truncation = sdf_truncation;
  p.sdf_truncation_scale = sdf_truncation_scale;
  p.integration_weight_sample = integration_weight_sample;
  p.integration_weight_max = integration_weight_max_;
  p.virtual_voxel_size = virtual_voxel_size;
  p.n_frames_invalidate_voxels = n_frames_invalidate_voxels;
  p.voxel_extents_scale = voxel_extents_scale;
  p.marching_cubes_threshold = marching_cubes_threshold;
  p.min_weight_threshold = min_weight_threshold;
  p.projective_sdf = projective_sdf ? 1 : 0;
  p.min_depth = min_depth;
  p.max_depth = max_depth;
  p.sdf_var_threshold = sdf_var_threshold;
  p.vertices_merging_threshold = vertices_merging_threshold;
  // capacities: 0 = the reference rule applied to the free HBM of the device (geowrapper.cpp:37-54);
  // MRHASH_NUM_SDF_BLOCKS / MRHASH_DEVICE override
  if (const char* e = std::getenv("MRHASH_NUM_SDF_BLOCKS")) p.num_sdf_blocks = std::strtoull(e, nullptr, 10);
  if (const char* e = std::getenv("MRHASH_DEVICE")) p.device_id = std::atoi(e);
  device_id_ = p.device_id;
  if (const char* e = std::getenv("MRHASH_STREAM")) streaming_enabled_ = std::atoi(e) != 0;
  if (const char* e = std::getenv("MRH_SYNC_COMPUTE")) sync_compute_ = std::atoi(e) != 0;
  p.shard_count = 1;
  int rc = mrh_create(&p, &ctx_);
  if (rc != MRH_OK) throw std::runtime_error(std::string("GeoWrapper::GeoWrapper | ") + mrh_last_error(nullptr));
  mrh_stats st;
  check(mrh_get_stats(ctx_, &st), "GeoWrapper");
  num_sdf_blocks_ = (int) st.num_sdf_blocks;
  hash_num_buckets_ = num_sdf_blocks_;  // geowrapper.cpp:50
  max_num_sdf_block_integrate_from_global_hash_ = (int) (st.num_sdf_blocks / 7);  // 0.10 / 0.70 of the block budget, :53-54
  pose_ = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  camera_in_lidar_ = pose_;
}

GeoWrapper::~GeoWrapper() {
  mrh_destroy(ctx_);  // detaches from the communicator
  mrh_comm_destroy(comm_);
}

// ---- multi-GPU: RCCL behind the C ABI (include/mrhash_comm.h) ------------------------------------------------------------
std::array<uint8_t, MRH_COMM_ID_BYTES> GeoWrapper::commUniqueId() {
  std::array<uint8_t, MRH_COMM_ID_BYTES> id;
  if (mrh_comm_unique_id(id.data()) != MRH_OK) throw std::runtime_error(std::string("GeoWrapper::commUniqueId | ") + mrh_comm_last_error(nullptr));
  return id;
}

void GeoWrapper::commInit(const std::array<uint8_t, MRH_COMM_ID_BYTES>& id, int rank, int world, int chunk_log2, bool tile_sharded) {
  if (comm_) throw std::runtime_error("GeoWrapper::commInit | a communicator is already attached");
  if (mrh_comm_create(id.data(), rank, world, device_id_, &comm_) != MRH_OK)
    throw std::runtime_error(std::string("GeoWrapper::commInit | ") + mrh_comm_last_error(nullptr));
  check(mrh_comm_attach(ctx_, comm_), "commInit");
  comm_rank_ = rank; comm_world_ = world; comm_chunk_log2_ = chunk_log2; tile_sharded_ = tile_sharded;
  if (tile_sharded) check(mrh_set_sharding(ctx_, rank, world, chunk_log2), "commInit");
  streaming_enabled_ = false;  // a sharded map is sized per rank; the host chunk grid and the exchange steps do not mix
}

void GeoWrapper::mergeSubmaps() {
  if (!comm_) throw std::runtime_error("GeoWrapper::mergeSubmaps | commInit has not been called");
  mrh_comm_merge_info info;
  check(mrh_comm_merge_submaps(ctx_, comm_chunk_log2_, &info), "mergeSubmaps");
  tile_sharded_ = true;
}

void GeoWrapper::setCurrPose(const std::array<float, 3>& t, const std::array<float, 4>& q) {
  // Eigen::Quaternionf(qw,qx,qy,qz).toRotationMatrix() (Eigen 3.4.0 Quaternion.h), float arithmetic, no normalisation
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  pose_ = {1.f - (tyy + tzz), txy - twz, txz + twy, t[0],
           txy + twz, 1.f - (txx + tzz), tyz - twx, t[1],
           txz - twy, tyz + twx, 1.f - (txx + tyy), t[2],
           0.f, 0.f, 0.f, 1.f};
}

void GeoWrapper::setCamera(float fx, float fy, float cx, float cy, int rows, int cols, float min_depth, float max_depth, int camera_model) {
  check(mrh_set_camera(ctx_, fx, fy, cx, cy, rows, cols, min_depth, max_depth, camera_model), "setCamera");
  max_depth_ = max_depth;
  // Streaming radius.  The reference uses max_depth itself (geowrapper.cpp:138), which is the z-range of a pinhole frame,
  // not its reach: a voxel at z = max_depth in an image corner is farther from the camera centre than that, so blocks
  // still inside the frustum can be paged out and then re-created empty.  Here the radius is the true reach of a frame:
  // (max_depth + truncation) along the most oblique pixel ray, plus one block diagonal.
  const float trunc = sdf_truncation_ + sdf_truncation_scale_ * max_depth;
  float sec = 1.f;
  if (camera_model == MRH_CAMERA_PINHOLE && fx > 0.f && fy > 0.f) {
    const float ux = std::max(cx + 0.5f, (float) cols - cx) / fx, uy = std::max(cy + 0.5f, (float) rows - cy) / fy;
    sec = std::sqrt(1.f + ux * ux + uy * uy);
  }
  reach_ = (max_depth + trunc) * sec + 8.f * virtual_voxel_size_ * std::sqrt(3.f);
}

// The reference copies the image into its own host buffer here and uploads it in compute() (geowrapper.cpp:300-321,
// :125-126).  Here the setter's copy IS the upload: mrh_upload_* copies into pinned staging (the numpy array is free on
// return, as with the reference) and starts the host-to-device copy on a second stream right away, under the previous
// frame's kernels; compute() only enqueues the frame.
void GeoWrapper::setDepthImage(const float* data, size_t rows, size_t cols) {
  check(mrh_upload_depth(ctx_, data, (int) rows, (int) cols), "setDepthImage");
  have_depth_ = true;
}

void GeoWrapper::setRGBImage(const uint8_t* data, size_t rows, size_t cols) {
  check(mrh_upload_rgb(ctx_, data, (int) rows, (int) cols), "setRGBImage");
  have_rgb_ = true;
}

void GeoWrapper::setPointCloud(const float* pts, size_t n, const float* normals_or_null) {
  point_cloud_.assign(pts, pts + 3 * n);
  if (normals_or_null) normals_.assign(normals_or_null, normals_or_null + 3 * n);
  else normals_.clear();
}

// ---- streamer, host side: chunk grid (streamer.cuh:251-352, streamer.cpp:214-247, :292-331) ---------------------------
namespace {
constexpr float kStreamThreshold = 0.15f;  // params.h:28 stream_threshold
}

std::array<int, 3> GeoWrapper::worldToChunks(const std::array<float, 3>& pw) const {
  std::array<int, 3> c;
  const float ext = (float) voxel_extents_scale_;  // chunk edge in metres (geowrapper.cpp:56)
  for (int a = 0; a < 3; a++) {
    const float p = pw[a] / ext;
    const float s = (float) ((0.f < p) - (p < 0.f));
    c[a] = (int) (p + s * 0.5f);
  }
  return c;
}

float GeoWrapper::chunkRadius() const {
  const float ext = (float) voxel_extents_scale_;
  return std::sqrt(3.f * ext * ext) / 2.0f;  // voxel_extents_.norm() / 2 (streamer.cuh:315-317)
}

// any part of the chunk within `radius` of `center`?  (The reference's isChunkInSphere, streamer.cuh:345-352, asks for
// the WHOLE chunk to be inside, which leaves blocks the next frame can touch on the host; see stream().)
bool GeoWrapper::chunkTouchesSphere(const std::array<int, 3>& chunk, const std::array<float, 3>& center, float radius) const {
  const float ext = (float) voxel_extents_scale_;
  const float dx = chunk[0] * ext - center[0], dy = chunk[1] * ext - center[1], dz = chunk[2] * ext - center[2];
  return std::sqrt(dx * dx + dy * dy + dz * dz) <= radius + chunkRadius();
}

void GeoWrapper::streamOutToGrid(const std::array<float, 3>& center, float radius) {
  uint64_t n = 0;
  check(mrh_stream_out(ctx_, center.data(), radius, nullptr, nullptr, 0, &n), "stream");
  if (n == 0) return;
  std::vector<mrh_block_desc> descs(n);
  std::vector<mrh_voxel> vox(n * 512);
  check(mrh_stream_out(ctx_, center.data(), radius, descs.data(), vox.data(), n, &n), "stream");
  const float bs = 8.f * virtual_voxel_size_;
  for (uint64_t k = 0; k < n; k++) {  // Streamer::integrateInChunkGrid
    const std::array<float, 3> pw = {(float) descs[k].x * bs, (float) descs[k].y * bs, (float) descs[k].z * bs};
    HostBlock b;
    b.desc = descs[k];
    b.voxels.assign(vox.begin() + k * 512, vox.begin() + (k + 1) * 512);
    grid_[worldToChunks(pw)].push_back(std::move(b));
  }
}

void GeoWrapper::streamInFromGrid(const std::array<float, 3>* center, float radius) {
  std::vector<mrh_block_desc> descs;
  std::vector<mrh_voxel> vox;
  std::vector<std::array<int, 3>> taken;
  for (const auto& kv : grid_) {
    if (center && !chunkTouchesSphere(kv.first, *center, radius)) continue;
    for (const HostBlock& b : kv.second) {
      descs.push_back(b.desc);
      vox.insert(vox.end(), b.voxels.begin(), b.voxels.end());
    }
    taken.push_back(kv.first);
  }
  if (descs.empty()) return;
  // import first, forget the host copies only once the device holds them: a failed import (pool full) must not lose blocks
  check(mrh_import_blocks(ctx_, descs.data(), vox.data(), descs.size()), "stream");
  for (const auto& k : taken) grid_.erase(k);  // "it lives on the device from now"
}

// Streamer::stream (streamer.cpp:333-354) with the two radii chosen so that paging is TRANSPARENT — a block is either on
// the device or in the host grid, and everything a frame can touch is on the device:
//   in : every chunk that touches the sphere of the frame's reach comes back (also called every frame by compute());
//   out: blocks farther than reach + 2 chunk radii leave — none of them lies in a chunk the next stream-in would take.
void GeoWrapper::stream(const std::array<float, 3>& camera_position, float radius) {
  streamOutToGrid(camera_position, radius + 2.f * chunkRadius() + 1e-3f);
  streamInFromGrid(&camera_position, radius);
}

size_t GeoWrapper::hostGridBlocks() const {
  size_t n = 0;
  for (const auto& kv : grid_) n += kv.second.size();
  return n;
}

void GeoWrapper::compute() {
  if (streaming_enabled_) {
    const std::array<float, 3> cam = {pose_[3], pose_[7], pose_[11]};
    if (!grid_.empty()) streamInFromGrid(&cam, reach_);  // host-only test unless a paged-out chunk is within reach
    // geowrapper.cpp:137-138 reads the free count back every frame; here it is the newest level the device has reported
    // (a frame or two old, no stall): paging is transparent in this library, so the trigger frame does not matter
    int64_t free_fine = 0;
    check(mrh_peek_free_blocks(ctx_, &free_fine, nullptr, nullptr), "compute");
    if ((float) free_fine <= kStreamThreshold * (float) num_sdf_blocks_) stream(cam, reach_);
  }
  const float R[9] = {pose_[0], pose_[1], pose_[2], pose_[4], pose_[5], pose_[6], pose_[8], pose_[9], pose_[10]};
  const float t[3] = {pose_[3], pose_[7], pose_[11]};
  check(mrh_set_pose(ctx_, R, t), "compute");
  if (have_depth_ && have_rgb_) {  // geowrapper.cpp:140
    check(mrh_integrate(ctx_, n_frames_invalidate_voxels_), "compute");
    // The reference reports an exhausted pool / a full table from the device (printf, vds.cu:566-569) and carries on without
    // the affected blocks.  Same here, without a stall: the flags of a frame that finished a moment ago arrive with the
    // pool-level report; each is announced once.
    uint32_t flags = 0;
    if (sync_compute_) {
      // MRH_SYNC_COMPUTE=1 / _setSyncCompute(true): compute() as the reference's — blocking (geowrapper.cpp:118-148 ends in
      // cudaDeviceSynchronize), the frame's own capacity flags announced in this call, a device error thrown by it.
      const int rc = mrh_sync(ctx_);
      if (rc != MRH_OK && rc != MRH_ERR_CAPACITY && rc != MRH_ERR_OUT_OF_RANGE) check(rc, "compute");
      mrh_stats st;
      check(mrh_get_stats(ctx_, &st), "compute");
      flags = st.error_flags & ~flags_announced_;
      flags_announced_ |= st.error_flags;
      last_compute_flags_ = flags;
    }
    if (sync_compute_ ? flags != 0 : (mrh_peek_error_flags(ctx_, &flags) == MRH_OK && flags)) {
      if (flags & 1u) std::cerr << "GeoWrapper::compute | SDF block pool exhausted: blocks of recent frames were skipped" << std::endl;
      if (flags & 2u) std::cerr << "GeoWrapper::compute | hash table probe limit reached: blocks of recent frames were skipped" << std::endl;
      if (flags & 4u) std::cerr << "GeoWrapper::compute | block coordinates left the +-2^20 key range" << std::endl;
    }
    if (gs_enabled_) {  // geowrapper.cpp:142-143 runGS -> extractNodesQTree + checkNodes; Add_gaussians keeps what they emit
      const mrh_splat_seed* seeds = nullptr;
      uint64_t n = 0;
      check(mrh_splat_seeds(ctx_, qtree_thresh_, qtree_min_pixel_size_, &seeds, &n), "compute");
      seeds_.insert(seeds_.end(), seeds, seeds + n);
    }
  }
  if (!point_cloud_.empty()) {  // geowrapper.cpp:146-147: VoxelContainer::integrate(point_cloud, eigenvectors, weights, ...)
    // Normals passed along with the points (one per point) drive the normal-direction SDF when the wrapper was built with
    // projective_sdf = false (vds.cu:1248-1251, :1322-1326); with the projective SDF (every shipped configuration and
    // runner) the reference only normalises them and never uses them (vds.cu:1236), so they are simply not needed.
    // Estimating normals (the reference's MAD-tree, geowrapper.cpp:377-403) is the caller's business here.
    if (!normals_.empty()) check(mrh_upload_normals(ctx_, normals_.data(), normals_.size() / 3), "compute");
    check(mrh_upload_points(ctx_, point_cloud_.data(), point_cloud_.size() / 3), "compute");
    check(mrh_integrate_points(ctx_, n_frames_invalidate_voxels_), "compute");
  }
}

// Streamer::isChunkInSphere (streamer.cuh:345-352), literally: the chunk CENTRE within |radius - chunk radius| of the sphere
// centre.  Used by the chunk loop of extractMesh only (stream() pages with its own, transparent radii).
bool GeoWrapper::chunkInSphereRef(const std::array<int, 3>& chunk, const std::array<float, 3>& center, float radius) const {
  const float ext = (float) voxel_extents_scale_;
  const float dx = chunk[0] * ext - center[0], dy = chunk[1] * ext - center[1], dz = chunk[2] * ext - center[2];
  return std::sqrt(dx * dx + dy * dy + dz * dz) <= std::fabs(radius - chunkRadius());
}

// Streamer::streamInToGPU(center, radius) as the chunk loop uses it (streamer.cpp:292-331, :358-378): every chunk of the host
// grid that lies in the sphere goes to the device and leaves the grid.  Refuses BEFORE touching anything when the sphere holds
// more blocks than the pool has free (the reference would run its heap into the ground, silently).
uint64_t GeoWrapper::streamInSphereRef(const std::array<float, 3>& center, float radius) {
  std::vector<std::array<int, 3>> taken;
  uint64_t n = 0;
  for (const auto& kv : grid_)
    if (chunkInSphereRef(kv.first, center, radius)) { taken.push_back(kv.first); n += kv.second.size(); }
  if (n == 0) return 0;
  int64_t free_fine = 0;
  check(mrh_get_free_blocks(ctx_, &free_fine, nullptr), "extractMesh");
  if ((int64_t) n > free_fine)
    throw std::runtime_error("GeoWrapper::extractMesh | a sphere of radius " + std::to_string(radius) + " m holds " + std::to_string(n) +
                             " blocks, the pool has " + std::to_string(free_fine) + " free (num_sdf_blocks " + std::to_string(num_sdf_blocks_) +
                             "): nothing was moved, the map is intact on the host grid");
  std::vector<mrh_block_desc> descs;
  std::vector<mrh_voxel> vox;
  descs.reserve(n);
  vox.reserve(n * 512);
  for (const auto& k : taken)
    for (const HostBlock& b : grid_[k]) {
      descs.push_back(b.desc);
      vox.insert(vox.end(), b.voxels.begin(), b.voxels.end());
    }
  check(mrh_import_blocks(ctx_, descs.data(), vox.data(), descs.size()), "extractMesh");
  for (const auto& k : taken) grid_.erase(k);
  return n;
}

void GeoWrapper::extractMesh(const std::string& filename) {
  const bool dbg = std::getenv("MRH_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  // the previous mesh goes first: if this extraction throws, the getters must not read buffers the library has already reused
  mesh_v_ = mesh_c_ = nullptr; mesh_f_ = nullptr; mesh_nv_ = mesh_nf_ = 0;
  V_.clear(); F_.clear(); C_.clear();
  mesh_cached_ = true;

  if (comm_ && comm_world_ > 1) {
    // sharded map: boundary blocks to every rank, every rank extracts what it owns, rank 0 merges the runs into the single-GPU
    // canonical order and post-processes; the halo goes again so that fusion can continue
    if (!tile_sharded_) throw std::runtime_error("GeoWrapper::extractMesh | frame-sharded sub-maps: call mergeSubmaps() first");
    if (!grid_.empty()) throw std::runtime_error("GeoWrapper::extractMesh | a sharded map cannot have blocks on the host grid");
    std::cout << "GeoWrapper::extractMesh | extracting (rank " << comm_rank_ << " of " << comm_world_ << ")..." << std::endl;
    uint64_t taken = 0, nt_all = 0;
    check(mrh_comm_exchange_halo(ctx_, &taken), "extractMesh");
    check(mrh_comm_gather_mesh(ctx_, 0, &nt_all), "extractMesh");
    check(mrh_drop_blocks(ctx_, MRH_DROP_HALO, nullptr), "extractMesh");
    if (comm_rank_ != 0) return;  // the mesh lives on rank 0
    std::cout << "MarchingCubesExtractor::extractIsoSurface | triangles extracted: " << nt_all << std::endl;
    writeMesh(filename, t0);
    return;
  }

  // ---- geowrapper.cpp:150-190.  The reference pages EVERYTHING out, then walks the chunk grid in steps of int(10 * max depth)
  // chunks: stream in the sphere of radius 10 * max depth around the step's chunk, marching cubes, merge into the running
  // mesh, page everything out again.  Whenever that loop has a single iteration whose sphere holds the whole map, and the map
  // fits the pool, its triangles are those of ONE extraction over the resident map — taken here without paging 6 KiB per block
  // out and in again (the only difference: the map stays on the device afterwards; the reference leaves it on the host).
  const float radius = 10.0f * max_depth_;  // params.h:35 radius_scale_chunk
  const int radiusi = std::max(1, (int) radius);  // (int) radius == 0 would never advance the reference's loops
  uint64_t n_dev = 0;
  check(mrh_dump_blocks(ctx_, nullptr, nullptr, 0, &n_dev), "extractMesh");
  std::vector<mrh_block_desc> dev_descs(n_dev ? n_dev : 1);
  if (n_dev) check(mrh_dump_blocks(ctx_, dev_descs.data(), nullptr, n_dev, &n_dev), "extractMesh");
  const float bs = 8.f * virtual_voxel_size_;
  std::array<int, 3> lo = {INT32_MAX, INT32_MAX, INT32_MAX}, hi = {INT32_MIN, INT32_MIN, INT32_MIN};
  auto widen = [&](const std::array<int, 3>& c) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); } };
  std::vector<std::array<int, 3>> dev_chunks(n_dev);
  for (uint64_t k = 0; k < n_dev; k++) {
    dev_chunks[k] = worldToChunks({(float) dev_descs[k].x * bs, (float) dev_descs[k].y * bs, (float) dev_descs[k].z * bs});
    widen(dev_chunks[k]);
  }
  for (const auto& kv : grid_) widen(kv.first);  // Streamer::computeBounds over what WOULD be on the host after streamAllOut
  uint64_t nt = 0;
  std::cout << "GeoWrapper::extractMesh | extracting..." << std::endl;
  if (n_dev + grid_.size() == 0) {
    check(mrh_extract_triangles(ctx_, nullptr, &nt), "extractMesh");
  } else {
    for (int a = 0; a < 3; a++) if (lo[a] == hi[a]) hi[a] += 1;  // geowrapper.cpp:165-173
    const std::array<float, 3> first_center = {(float) lo[0] * (float) voxel_extents_scale_, (float) lo[1] * (float) voxel_extents_scale_,
                                               (float) lo[2] * (float) voxel_extents_scale_};
    bool one_pass = true;
    for (int a = 0; a < 3; a++) one_pass = one_pass && (lo[a] + radiusi >= hi[a]);
    for (uint64_t k = 0; k < n_dev && one_pass; k++) one_pass = chunkInSphereRef(dev_chunks[k], first_center, radius);
    for (const auto& kv : grid_) { if (!one_pass) break; one_pass = chunkInSphereRef(kv.first, first_center, radius); }
    if (one_pass && !grid_.empty()) {
      int64_t free_fine = 0;
      check(mrh_get_free_blocks(ctx_, &free_fine, nullptr), "extractMesh");
      one_pass = (int64_t) hostGridBlocks() <= free_fine;
    }
    if (one_pass) {
      streamInFromGrid(nullptr, 0.f);  // whatever the streamer paged out takes part in the mesh
      check(mrh_extract_triangles(ctx_, nullptr, &nt), "extractMesh");  // the soup itself stays on the device
    } else {
      const std::array<float, 3> origin = {0.f, 0.f, 0.f};
      streamOutToGrid(origin, -1.f);  // Streamer::streamAllOut (geowrapper.cpp:153)
      check(mrh_mesh_merge_begin(ctx_), "extractMesh");
      try {
        for (int x = lo[0]; x < hi[0]; x += radiusi)
          for (int y = lo[1]; y < hi[1]; y += radiusi)
            for (int z = lo[2]; z < hi[2]; z += radiusi) {
              const float e = (float) voxel_extents_scale_;
              const std::array<float, 3> center = {(float) x * e, (float) y * e, (float) z * e};  // Streamer::chunkToWorld
              if (streamInSphereRef(center, radius) == 0) continue;  // an empty map: no triangles, nothing to merge (:181)
              uint64_t n_it = 0;
              check(mrh_extract_triangles(ctx_, nullptr, &n_it), "extractMesh");
              streamOutToGrid(origin, -1.f);  // streamAllOut (:186)
            }
      } catch (...) {
        uint64_t dummy = 0;
        (void) mrh_mesh_merge_end(ctx_, &dummy);
        const std::array<float, 3> o = {0.f, 0.f, 0.f};
        try { streamOutToGrid(o, -1.f); } catch (...) {}  // every block in exactly one place: the host grid
        throw;
      }
      check(mrh_mesh_merge_end(ctx_, &nt), "extractMesh");
    }
  }
  std::cout << "MarchingCubesExtractor::extractIsoSurface | triangles extracted: " << nt << std::endl;
  writeMesh(filename, t0);
}

// the mesh of the last extraction out of the library, and the ASCII PLY of geowrapper.cpp:194-227
void GeoWrapper::writeMesh(const std::string& filename, const double t0) {
  const bool dbg = std::getenv("MRH_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double *v = nullptr, *c = nullptr;
  const int32_t* f = nullptr;
  uint64_t nv = 0, nf = 0;
  check(mrh_extract_mesh(ctx_, &v, &nv, &f, &nf, &c), "extractMesh");
  const double t1 = now();
  mesh_v_ = v; mesh_c_ = c; mesh_f_ = f; mesh_nv_ = nv; mesh_nf_ = nf;
  mesh_cached_ = false;  // getVertices / getFaces / getColors copy on first use (cacheMesh)
  const double t2 = now();

  // ASCII PLY, same header and value formatting as geowrapper.cpp:194-227.  The reference streams every number through
  // an ofstream (`<<` on a double is printf's %g with precision 6); std::to_chars(general, 6) is defined to produce the
  // same characters, so the rows are formatted in parallel chunks and written in order: byte-identical file, a
  // fraction of the 0.9 s the iostream loop takes for a million triangles.
  std::ofstream ply(filename, std::ios::binary);
  if (!ply.is_open()) {
    std::cerr << "GeoWrapper::extractMesh | Failed to open file for writing: " << filename << std::endl;
    return;
  }
  ply << "ply\nformat ascii 1.0\nelement vertex " << nv << "\nproperty float x\nproperty float y\nproperty float z\n"
      << "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face " << nf
      << "\nproperty list uchar int vertex_indices\nend_header\n";
  const unsigned n_workers = (unsigned) std::min<uint64_t>(std::max(1u, std::min(32u, std::thread::hardware_concurrency())), (nv + nf) / 65536 + 1);
  struct Chunk { std::unique_ptr<char[]> buf; size_t len = 0; };
  std::vector<Chunk> chunks(2 * (size_t) n_workers);
  auto format_chunk = [&](unsigned k) {
    auto put_d = [](char* p, double v) { return std::to_chars(p, p + 32, v, std::chars_format::general, 6).ptr; };
    auto put_i = [](char* p, int v) { return std::to_chars(p, p + 16, v).ptr; };
    {
      const uint64_t lo = nv * k / n_workers, hi = nv * (k + 1) / n_workers;
      Chunk& c = chunks[k];
      c.buf.reset(new char[(hi - lo) * 64 + 1]);  // 3 x <= 13 chars + 3 x <= 3 chars + separators
      char* p = c.buf.get();
      for (uint64_t i = lo; i < hi; ++i) {
        for (int a = 0; a < 3; a++) { p = put_d(p, mesh_v_[i * 3 + a]); *p++ = ' '; }
        for (int a = 0; a < 3; a++) { p = put_i(p, (int) (unsigned char) (mesh_c_[i * 3 + a])); *p++ = a == 2 ? '\n' : ' '; }
      }
      c.len = (size_t) (p - c.buf.get());
    }
    {
      const uint64_t lo = nf * k / n_workers, hi = nf * (k + 1) / n_workers;
      Chunk& c = chunks[n_workers + k];
      c.buf.reset(new char[(hi - lo) * 40 + 1]);  // "3" + 3 x (space + <= 11 chars) + newline
      char* p = c.buf.get();
      for (uint64_t i = lo; i < hi; ++i) {
        *p++ = '3';
        for (int a = 0; a < 3; a++) { *p++ = ' '; p = put_i(p, mesh_f_[i * 3 + a]); }
        *p++ = '\n';
      }
      c.len = (size_t) (p - c.buf.get());
    }
  };
  if (n_workers == 1) format_chunk(0);
  else {
    std::vector<std::thread> workers;
    for (unsigned k = 0; k < n_workers; k++) workers.emplace_back(format_chunk, k);
    for (std::thread& w : workers) w.join();
  }
  const double t3 = now();
  for (const Chunk& c : chunks) ply.write(c.buf.get(), (std::streamsize) c.len);
  ply.close();
  if (dbg) std::cerr << "[geowrapper] extractMesh: library " << t1 - t0 << " ms, formatting " << t3 - t2
                     << ", writing " << now() - t3 << std::endl;
  std::cout << "GeoWrapper::extractMesh | written " << nv << " vertices and " << nf << " faces to " << filename << std::endl;
}

void GeoWrapper::cacheMesh() const {
  if (mesh_cached_) return;
  V_.assign(mesh_v_, mesh_v_ + mesh_nv_ * 3);
  C_.assign(mesh_c_, mesh_c_ + mesh_nv_ * 3);
  F_.assign(mesh_f_, mesh_f_ + mesh_nf_ * 3);
  mesh_cached_ = true;
}

void GeoWrapper::streamAllOut() {
  // The reference pages every block to the host chunk grid here (streamer.cpp:250-281) because its marching cubes
  // then walks the grid chunk by chunk.  With 288 GB of HBM the whole map is extracted in one pass instead
  // (extractMesh brings back whatever the streamer paged out), so this only drains the stream.
  check(mrh_sync(ctx_), "streamAllOut");
}

void GeoWrapper::clearBuffers() {
  std::cout << "clearing buffers..." << std::endl;
  cacheMesh();  // the mesh of the last extractMesh outlives the map, as the reference's host matrices do
  check(mrh_reset(ctx_), "clearBuffers");
  flags_announced_ = last_compute_flags_ = 0;
  grid_.clear();  // Streamer::clearGrid (geowrapper.cpp:555)
}

void GeoWrapper::serializeData(const std::string& filename_hash, const std::string& filename_voxel) {
  // PLY point clouds of block origins and weighted voxels (streamer.cpp:104-160): x y z r g b weight [sdf]
  uint64_t n = 0;
  check(mrh_dump_blocks(ctx_, nullptr, nullptr, 0, &n), "serializeData");
  std::vector<mrh_block_desc> descs(n ? n : 1);
  std::vector<mrh_voxel> vox((n ? n : 1) * 512);
  check(mrh_dump_blocks(ctx_, descs.data(), vox.data(), n, &n), "serializeData");
  struct Pt { float x, y, z, r, g, b, w, s; };
  std::vector<Pt> hash_pts, voxel_pts;
  for (uint64_t k = 0; k < n; ++k) {
    const mrh_block_desc& d = descs[k];
    const float bx = (float) (d.x * 8) * virtual_voxel_size_, by = (float) (d.y * 8) * virtual_voxel_size_, bz = (float) (d.z * 8) * virtual_voxel_size_;
    const int side = 8 >> d.resolution, sf = 1 << d.resolution;
    float wsum = 0.f;
    unsigned valid = 0;
    for (int l = 0; l < side * side * side; ++l) {
      const mrh_voxel& v = vox[k * 512 + l];
      if (v.weight == 0) continue;
      const int lx = (l % side) * sf, ly = ((l % (side * side)) / side) * sf, lz = (l / (side * side)) * sf;
      voxel_pts.push_back({bx + lx * virtual_voxel_size_, by + ly * virtual_voxel_size_, bz + lz * virtual_voxel_size_,
                           d.resolution == 0 ? 1.f : 0.f, d.resolution == 1 ? 1.f : 0.f, 0.f, (float) v.weight, v.sdf});
      wsum += (float) v.weight;
      valid++;
    }
    if (valid) hash_pts.push_back({bx, by, bz, d.resolution == 0 ? 1.f : 0.f, d.resolution == 1 ? 1.f : 0.f, 0.f, wsum / (float) valid, 0.f});
  }
  auto write = [](const std::string& fn, const std::vector<Pt>& pts, bool with_sdf) {
    std::ofstream o(fn);
    if (!o.is_open()) { std::cerr << "GeoWrapper::serializeData | cannot open " << fn << std::endl; return; }
    o << "ply\nformat ascii 1.0\nelement vertex " << pts.size() << "\nproperty float x\nproperty float y\nproperty float z\n"
      << "property float red\nproperty float green\nproperty float blue\nproperty float weight\n" << (with_sdf ? "property float sdf\n" : "") << "end_header\n";
    for (const Pt& p : pts) {
      o << p.x << " " << p.y << " " << p.z << " " << p.r << " " << p.g << " " << p.b << " " << p.w;
      if (with_sdf) o << " " << p.s;
      o << "\n";
    }
  };
  write(filename_hash, hash_pts, false);
  write(filename_voxel, voxel_pts, true);
  std::cout << "Streamer::serializeData | written " << hash_pts.size() << " hash points and " << voxel_pts.size() << " voxels to " << filename_hash
            << " and " << filename_voxel << std::endl;
}

void GeoWrapper::serializeGrid(const std::string& filename) {
  // flat documented format instead of cista (serializer.h:16-77): "MRHGRID1" u64 n, then n x {mrh_block_desc, 512 x mrh_voxel}
  uint64_t n = 0;
  check(mrh_dump_blocks(ctx_, nullptr, nullptr, 0, &n), "serializeGrid");
  std::vector<mrh_block_desc> descs(n ? n : 1);
  std::vector<mrh_voxel> vox((n ? n : 1) * 512);
  check(mrh_dump_blocks(ctx_, descs.data(), vox.data(), n, &n), "serializeGrid");
  std::ofstream o(filename, std::ios::binary);
  if (!o.is_open()) throw std::runtime_error("GeoWrapper::serializeGrid | cannot open " + filename);
  const uint64_t total = n + hostGridBlocks();  // device-resident blocks, then the streamer's host chunk grid
  o.write("MRHGRID1", 8);
  o.write((const char*) &total, 8);
  for (uint64_t k = 0; k < n; ++k) {
    o.write((const char*) &descs[k], sizeof(mrh_block_desc));
    o.write((const char*) &vox[k * 512], 512 * sizeof(mrh_voxel));
  }
  for (const auto& kv : grid_)
    for (const HostBlock& b : kv.second) {
      o.write((const char*) &b.desc, sizeof(mrh_block_desc));
      o.write((const char*) b.voxels.data(), 512 * sizeof(mrh_voxel));
    }
}

void GeoWrapper::deserializeGrid(const std::string& filename) {
  // inverse of serializeGrid: blocks are inserted (or overwritten) through mrh_import_blocks, the stream-in half of
  // the reference's streamer (Streamer::streamInToGPU, streamer.cpp:358-378) as far as a resident map needs it
  std::ifstream in(filename, std::ios::binary);
  if (!in.is_open()) throw std::runtime_error("GeoWrapper::deserializeGrid | cannot open " + filename);
  char magic[8];
  uint64_t n = 0;
  in.read(magic, 8);
  in.read((char*) &n, 8);
  if (!in || std::memcmp(magic, "MRHGRID1", 8) != 0) throw std::runtime_error("GeoWrapper::deserializeGrid | not a MRHGRID1 file: " + filename);
  std::vector<mrh_block_desc> descs(n ? n : 1);
  std::vector<mrh_voxel> vox((n ? n : 1) * 512);
  for (uint64_t k = 0; k < n; ++k) {
    in.read((char*) &descs[k], sizeof(mrh_block_desc));
    in.read((char*) &vox[k * 512], 512 * sizeof(mrh_voxel));
  }
  if (!in) throw std::runtime_error("GeoWrapper::deserializeGrid | truncated file: " + filename);
  check(mrh_import_blocks(ctx_, descs.data(), vox.data(), n), "deserializeGrid");
}

void GeoWrapper::GSSavePointCloud(const std::string& folder) {
  if (!gs_enabled_) {
    std::cerr << "GeoWrapper::GSSavePointCloud | GS container not initialized" << std::endl;  // geowrapper.cpp:233-236
    return;
  }
  // the reference writes the optimised model (GaussianModel::Save_ply, src/gs/gaussian.cu:260-282) to
  // <folder>/point_cloud.ply; here the same file holds the seeds the model is initialised from
  const std::string mk = "mkdir -p '" + folder + "'";
  if (std::system(mk.c_str()) != 0) throw std::runtime_error("GeoWrapper::GSSavePointCloud | cannot create " + folder);
  std::ofstream ply(folder + "/point_cloud.ply");
  if (!ply.is_open()) throw std::runtime_error("GeoWrapper::GSSavePointCloud | cannot write " + folder + "/point_cloud.ply");
  ply << "ply\nformat ascii 1.0\nelement vertex " << seeds_.size()
      << "\nproperty float x\nproperty float y\nproperty float z\nproperty float scale\n"
         "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n";
  for (const mrh_splat_seed& s : seeds_)
    ply << s.p[0] << " " << s.p[1] << " " << s.p[2] << " " << s.scale << " " << (int) s.rgb[0] << " " << (int) s.rgb[1] << " " << (int) s.rgb[2] << "\n";
  std::cout << "GeoWrapper::GSSavePointCloud | written " << seeds_.size() << " splat seeds to " << folder << std::endl;
}

void GeoWrapper::GSFinalOpt() {  // geowrapper.cpp:241-244: no-op without a GS container
  if (gs_enabled_) throw std::runtime_error("GeoWrapper::GSFinalOpt | Gaussian-splatting optimisation is outside this library's scope");
}

}  // namespace pygeowrapper
