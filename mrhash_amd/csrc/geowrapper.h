// geowrapper.h — C++ host facade with the reference's GeoWrapper surface (geowrapper.h:18-260 of
// rvp-group/mrhash), sitting on the C ABI of include/mrhash_hip.h.  No CUDA/HIP types appear here: the
// facade owns host-side copies of the inputs (the reference's setters copy too, geowrapper.cpp:246-321)
// and one opaque mrh_ctx.
#pragma once

#include <array>
#include <map>
#include <cstdint>
#include <string>
#include <vector>

#include "mrhash_hip.h"
#include "mrhash_comm.h"

namespace pygeowrapper {

class GeoWrapper {
public:
  // same 15 arguments, same order, same defaults as geowrapper.h:63-78 / pygeowrapper.cpp:14-29
  GeoWrapper(float sdf_truncation, float sdf_truncation_scale, int integration_weight_sample, float virtual_voxel_size,
             int n_frames_invalidate_voxels, int voxel_extents_scale, bool viewer_active, float marching_cubes_threshold,
             uint8_t min_weight_threshold, float min_depth, float max_depth, const std::string& gs_optimization_param_path = "",
             float sdf_var_threshold = 0.f, float vertices_merging_threshold = 0.f, bool projective_sdf = true);
  ~GeoWrapper();
  GeoWrapper(const GeoWrapper&) = delete;
  GeoWrapper& operator=(const GeoWrapper&) = delete;

  // getters / setters: like the reference they only touch the cached host copy (geowrapper.h:80-109)
  int getHashNumBuckets() const { return hash_num_buckets_; }
  int getNumSdfBlocks() const { return num_sdf_blocks_; }
  int getHashBucketSize() const { return hash_bucket_size_; }
  float getSdfTruncation() const { return sdf_truncation_; }
  float getSdfTruncationScale() const { return sdf_truncation_scale_; }
  int getIntegrationWeightSample() const { return integration_weight_sample_; }
  int getIntegrationWeightMax() const { return integration_weight_max_; }
  float getVirtualVoxelSize() const { return virtual_voxel_size_; }
  int getLinkedListSize() const { return linked_list_size_; }
  int getNFramesInvalidateVoxels() const { return n_frames_invalidate_voxels_; }
  int getMaxNumSdfBlockIntegrateFromGlobalHash() const { return max_num_sdf_block_integrate_from_global_hash_; }
  int getVoxelExtentsScale() const { return voxel_extents_scale_; }
  void setHashNumBuckets(int v) { hash_num_buckets_ = v; }
  void setNumSdfBlocks(int v) { num_sdf_blocks_ = v; }
  void setHashBucketSize(int v) { hash_bucket_size_ = v; }
  void setSdfTruncation(float v) { sdf_truncation_ = v; }
  void setSdfTruncationScale(float v) { sdf_truncation_scale_ = v; }
  void setIntegrationWeightSample(int v) { integration_weight_sample_ = v; }
  void setIntegrationWeightMax(int v) { integration_weight_max_ = v; }
  void setVirtualVoxelSize(float v) { virtual_voxel_size_ = v; }
  void setLinkedListSize(int v) { linked_list_size_ = v; }
  void setNFramesInvalidateVoxels(int v) { n_frames_invalidate_voxels_ = v; }
  void setMaxNumSdfBlockIntegrateFromGlobalHash(int v) { max_num_sdf_block_integrate_from_global_hash_ = v; }
  void setVoxelExtentsScale(int v) { voxel_extents_scale_ = v; }

  // translation <tx,ty,tz>, quaternion <qx,qy,qz,qw> (geowrapper.cpp:86-92)
  void setCurrPose(const std::array<float, 3>& t, const std::array<float, 4>& q);
  const std::array<float, 16>& getCurrPose() const { return pose_; }  // row-major 4x4
  void setCameraInLidar(const std::array<float, 16>& m) { camera_in_lidar_ = m; }
  void setCamera(float fx, float fy, float cx, float cy, int rows, int cols, float min_depth, float max_depth, int camera_model);

  // raw-pointer forms of the numpy setters; shape checks live in the binding (geowrapper.cpp:246-321)
  void setDepthImage(const float* data, size_t rows, size_t cols);
  void setRGBImage(const uint8_t* data, size_t rows, size_t cols);
  void setPointCloud(const float* pts, size_t n, const float* normals_or_null);
  const std::vector<float>& pointCloud() const { return point_cloud_; }
  const std::vector<float>& normals() const { return normals_; }

  void compute();                                // geowrapper.cpp:118-148
  void extractMesh(const std::string& filename);  // geowrapper.cpp:150-230
  // V / F / C of the last extractMesh.  The library keeps them (until the next extraction); the copies the getters hand out are
  // made on first use — extractMesh itself formats the PLY straight from the library's buffers (83 MB of copies, a third
  // of the call, when nobody asks for the arrays).
  const std::vector<double>& vertices() const { cacheMesh(); return V_; }
  const std::vector<int32_t>& faces() const { cacheMesh(); return F_; }
  const std::vector<double>& colors() const { cacheMesh(); return C_; }

  void streamAllOut();
  // Streamer::stream (streamer.cpp:333-354): blocks farther than `radius` from `camera_position` leave the device for the
  // host chunk grid, chunks inside the sphere come back.  compute() calls it when the pool runs low (geowrapper.cpp:137-138).
  void stream(const std::array<float, 3>& camera_position, float radius);
  size_t hostGridBlocks() const;
  void setSyncCompute(bool on) { sync_compute_ = on; }
  uint32_t lastComputeFlags() const { return last_compute_flags_; }  // sync mode: flags the last compute() raised (1 pool, 2 table, 4 key range)  // blocks currently held by the host chunk grid
  void clearBuffers();
  void serializeData(const std::string& filename_hash, const std::string& filename_voxel);
  void serializeGrid(const std::string& filename);
  void deserializeGrid(const std::string& filename);
  void GSSavePointCloud(const std::string& folder);
  void GSFinalOpt();
  const std::vector<mrh_splat_seed>& splatSeeds() const { return seeds_; }  // accumulated over compute() calls

  // ---- multi-GPU (include/mrhash_comm.h; no reference counterpart): one GeoWrapper per process / GPU.
  // commUniqueId() on one rank, the 128 bytes handed to the others by the launcher, then commInit on every rank.
  //   tile_sharded = true : the result-identical mode — every rank is given every frame, compute() fuses the tiles this rank
  //                         owns (starve frames reduce over the ranks inside the library), extractMesh() exchanges the boundary
  //                         blocks, gathers the per-rank extractions on rank 0 and writes the file there (other ranks: no file,
  //                         empty V / F / C).
  //   tile_sharded = false: frame sharding — every rank fuses its own frames into its own sub-map; mergeSubmaps() folds them
  //                         into one tile-sharded map, after which extractMesh() works as above.
  static std::array<uint8_t, MRH_COMM_ID_BYTES> commUniqueId();
  void commInit(const std::array<uint8_t, MRH_COMM_ID_BYTES>& id, int rank, int world, int chunk_log2 = 3, bool tile_sharded = true);
  void mergeSubmaps();
  int commRank() const { return comm_rank_; }
  int commWorld() const { return comm_world_; }

  mrh_ctx* ctx() { return ctx_; }

private:
  void check(int rc, const char* what);
  // host side of the streamer (streamer.cuh:40-80, :251-352): chunk grid of streamed-out blocks
  struct HostBlock {
    mrh_block_desc desc;
    std::vector<mrh_voxel> voxels;  // 512, reference layout
  };
  std::array<int, 3> worldToChunks(const std::array<float, 3>& pw) const;
  float chunkRadius() const;
  bool chunkTouchesSphere(const std::array<int, 3>& chunk, const std::array<float, 3>& center, float radius) const;
  void streamOutToGrid(const std::array<float, 3>& center, float radius);
  void streamInFromGrid(const std::array<float, 3>* center, float radius);  // center == nullptr: everything
  // the chunk loop of extractMesh (geowrapper.cpp:162-188): the reference's own sphere test and stream-in
  bool chunkInSphereRef(const std::array<int, 3>& chunk, const std::array<float, 3>& center, float radius) const;
  uint64_t streamInSphereRef(const std::array<float, 3>& center, float radius);
  std::map<std::array<int, 3>, std::vector<HostBlock>> grid_;
  bool streaming_enabled_ = true;  // MRHASH_STREAM=0 turns the per-frame test off
  bool sync_compute_ = false;      // MRH_SYNC_COMPUTE=1: compute() blocks and reports its own frame's flags (the reference's contract)
  uint32_t flags_announced_ = 0, last_compute_flags_ = 0;
  float max_depth_ = 0.f;
  float reach_ = 0.f;  // farthest distance from the camera centre at which a frame can touch a block (set by setCamera)

  int hash_num_buckets_ = 0, num_sdf_blocks_ = 0, hash_bucket_size_ = 10;
  float sdf_truncation_, sdf_truncation_scale_;
  int integration_weight_sample_, integration_weight_max_ = 255;
  float virtual_voxel_size_;
  int linked_list_size_ = 7;
  int n_frames_invalidate_voxels_;
  int max_num_sdf_block_integrate_from_global_hash_ = 0;
  int voxel_extents_scale_;
  uint8_t min_weight_threshold_;
  float sdf_var_threshold_, vertices_merging_threshold_;
  std::array<float, 16> pose_;
  std::array<float, 16> camera_in_lidar_;
  bool have_depth_ = false, have_rgb_ = false;  // the images themselves live in the library (pinned staging + device slots)
  std::vector<uint8_t> rgb_;
  size_t depth_rows_ = 0, depth_cols_ = 0, rgb_rows_ = 0, rgb_cols_ = 0;
  std::vector<float> point_cloud_, normals_;
  void cacheMesh() const;
  void writeMesh(const std::string& filename, double t0);
  mutable std::vector<double> V_, C_;
  mutable std::vector<int32_t> F_;
  mutable bool mesh_cached_ = true;
  const double *mesh_v_ = nullptr, *mesh_c_ = nullptr;  // the library's buffers (mrh_extract_mesh)
  const int32_t* mesh_f_ = nullptr;
  uint64_t mesh_nv_ = 0, mesh_nf_ = 0;
  // 3DGS initialisation: on when a gs_optimization_param_path was given
  bool gs_enabled_ = false;
  float qtree_thresh_ = 0.1f;
  int qtree_min_pixel_size_ = 1;
  std::vector<mrh_splat_seed> seeds_;
  mrh_ctx* ctx_ = nullptr;
  mrh_comm* comm_ = nullptr;
  int comm_rank_ = 0, comm_world_ = 1, comm_chunk_log2_ = 3;
  bool tile_sharded_ = false;
  int device_id_ = 0;
};

}  // namespace pygeowrapper
