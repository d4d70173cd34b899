// oracle/_ref/libref_vhu_host.so — the part of the REFERENCE that compiles for the host as it lies, exposed over a C ABI
// so that tests can pin the oracle's restatement to the reference's own compiled code.  TEST INFRASTRUCTURE ONLY.
//
// What is compiled: mrhash/src/sdf/voxel_hash_utils.cuh (with cuda_utils.cuh and params.h, which it includes) from
// /root/reference, untouched, by g++.  Outside `#ifdef __CUDACC__` that header defines the voxel / hash-entry / vertex /
// triangle structs and the index helpers below as inline functions; everything else on the hot path is device code and
// cannot run here.  The only help the compile gets are two forced includes of headers that exist in this
// image — NVIDIA's own cuda_runtime.h (shipped inside the Triton wheel: vector types, make_int3, the empty host-side
// __device__ / __host__ macros) and <tuple> (voxel_hash_utils.cuh uses std::tie) — see oracle/Makefile.  No stub, no
// stand-in: nothing here defines a symbol the reference expects from elsewhere.
//
// This file is the driver (own code); it contains no reference source.
#include "voxel_hash_utils.cuh"

#include <cstddef>
#include <cstdint>

using namespace cupanutils::cugeoutils;

extern "C" {

// sizes and member offsets of the boundary structs: Voxel, HashEntry, Vertex, Triangle
void ref_struct_layout(int32_t out[16]) {
  int i = 0;
  out[i++] = (int32_t) sizeof(Voxel);
  out[i++] = (int32_t) offsetof(Voxel, sdf);
  out[i++] = (int32_t) offsetof(Voxel, sum_squared);
  out[i++] = (int32_t) offsetof(Voxel, rgb);
  out[i++] = (int32_t) offsetof(Voxel, weight);
  out[i++] = (int32_t) sizeof(HashEntry);
  out[i++] = (int32_t) offsetof(HashEntry, pos);
  out[i++] = (int32_t) offsetof(HashEntry, offset);
  out[i++] = (int32_t) offsetof(HashEntry, ptr);
  out[i++] = (int32_t) offsetof(HashEntry, resolution);
  out[i++] = (int32_t) sizeof(Vertex);
  out[i++] = (int32_t) offsetof(Vertex, c);
  out[i++] = (int32_t) sizeof(Triangle);
  out[i++] = (int32_t) offsetof(Triangle, v1);
  out[i++] = (int32_t) offsetof(Triangle, v2);
  out[i++] = 0;
}

// default-constructed Voxel and HashEntry, as raw bytes (12 and 24)
void ref_default_voxel(uint8_t out[12]) { const Voxel v; __builtin_memcpy(out, &v, sizeof v); }
void ref_default_hash_entry(uint8_t out[24]) { const HashEntry e; __builtin_memcpy(out, &e, sizeof e); }

// params.h constants the path depends on
void ref_constants(int64_t out[16]) {
  int i = 0;
  out[i++] = p0; out[i++] = p1; out[i++] = p2;
  out[i++] = sdf_block_size; out[i++] = total_sdf_block_size; out[i++] = finest_block_log2_dim;
  out[i++] = hash_bucket_size; out[i++] = linked_list_size; out[i++] = integration_weight_max;
  out[i++] = max_dda_iteration_count; out[i++] = n_threads; out[i++] = LOCK_ENTRY; out[i++] = FREE_ENTRY; out[i++] = NO_OFFSET;
  out[i++] = octree_branching_factor; out[i++] = 0;
}
double ref_float_constant(int which) {
  switch (which) {
    case 0: return (double) FLOAT_EPSILON;
    case 1: return (double) stream_threshold;
    case 2: return (double) radius_scale_chunk;
    case 3: return (double) SDFBlocks_ratio;
    default: return 0.0;
  }
}

// vhu.cuh:106-136: the index helpers, for whole arrays of inputs
void ref_linearize(const int32_t* xyz, int64_t n, int block_size, uint32_t* out) {
  for (int64_t i = 0; i < n; i++) out[i] = linearizeVoxelPos(make_int3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), block_size);
}
void ref_voxel_to_block_index(const int32_t* xyz, int64_t n, int block_size, uint32_t* out) {
  for (int64_t i = 0; i < n; i++) out[i] = virtualVoxelPosToSDFBlockIndex(make_int3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), block_size);
}
void ref_delinearize(const uint32_t* idx, int64_t n, int block_size, uint32_t* out_xyz) {
  for (int64_t i = 0; i < n; i++) {
    const uint3 p = delinearizeVoxelPos(idx[i], block_size);
    out_xyz[3 * i] = p.x; out_xyz[3 * i + 1] = p.y; out_xyz[3 * i + 2] = p.z;
  }
}
void ref_block_to_voxel(const int32_t* xyz, int64_t n, int32_t* out_xyz) {
  for (int64_t i = 0; i < n; i++) {
    const int3 v = SDFBlockToVirtualVoxelPos(make_int3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    out_xyz[3 * i] = v.x; out_xyz[3 * i + 1] = v.y; out_xyz[3 * i + 2] = v.z;
  }
}
// vhu.cuh:66-68 virtualVoxelPosToWorld(int3)
void ref_voxel_to_world(float vs, const int32_t* xyz, int64_t n, float* out_xyz) {
  for (int64_t i = 0; i < n; i++) {
    const float3 p = virtualVoxelPosToWorld(vs, make_int3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    out_xyz[3 * i] = p.x; out_xyz[3 * i + 1] = p.y; out_xyz[3 * i + 2] = p.z;
  }
}

}  // extern "C"
