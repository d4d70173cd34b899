// pygeowrapper.cpp — pybind11 binding with the surface of the reference's nanobind module
// (pybind/pygeowrapper.cpp:12-84): module `pygeowrapper`, class `GeoWrapper`, same method names, argument
// names/order/defaults and error behaviour (RuntimeError on shape violations).  nanobind is not available in
// this image; pybind11 returns numpy arrays where the reference returns Eigen matrices.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <vector>
#include <pybind11/stl.h>

#include "geowrapper.h"

namespace py = pybind11;
using pygeowrapper::GeoWrapper;

namespace {

template <typename T>
py::array_t<T> to_array(const std::vector<T>& v, size_t cols) {
  const size_t rows = cols ? v.size() / cols : 0;
  py::array_t<T> a({rows, cols});
  if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(T));
  return a;
}

}  // namespace

PYBIND11_MODULE(pygeowrapper, m) {
  m.doc() = "MI355X-native drop-in for mrhash.src.pygeowrapper (HIP kernels behind include/mrhash_hip.h)";
  py::class_<GeoWrapper>(m, "GeoWrapper")
    .def(py::init<float, float, int, float, int, int, bool, float, uint8_t, float, float, std::string, float, float, bool>(),
         py::arg("sdf_truncation"), py::arg("sdf_truncation_scale"), py::arg("integration_weight_sample"), py::arg("virtual_voxel_size"),
         py::arg("n_frames_invalidate_voxels"), py::arg("voxel_extents_scale"), py::arg("viewer_active"), py::arg("marching_cubes_threshold"),
         py::arg("min_weight_threshold"), py::arg("min_depth"), py::arg("max_depth"), py::arg("gs_optimization_param_path") = "",
         py::arg("sdf_var_threshold") = 0.f, py::arg("vertices_merging_threshold") = 0.f, py::arg("projective_sdf") = true)
    // getters
    .def("getHashNumBuckets", &GeoWrapper::getHashNumBuckets)
    .def("getNumSdfBlocks", &GeoWrapper::getNumSdfBlocks)
    .def("getHashBucketSize", &GeoWrapper::getHashBucketSize)
    .def("getSdfTruncation", &GeoWrapper::getSdfTruncation)
    .def("getSdfTruncationScale", &GeoWrapper::getSdfTruncationScale)
    .def("getIntegrationWeightSample", &GeoWrapper::getIntegrationWeightSample)
    .def("getIntegrationWeightMax", &GeoWrapper::getIntegrationWeightMax)
    .def("getVirtualVoxelSize", &GeoWrapper::getVirtualVoxelSize)
    .def("getLinkedListSize", &GeoWrapper::getLinkedListSize)
    .def("getNFramesInvalidateVoxels", &GeoWrapper::getNFramesInvalidateVoxels)
    .def("getMaxNumSdfBlockIntegrateFromGlobalHash", &GeoWrapper::getMaxNumSdfBlockIntegrateFromGlobalHash)
    .def("getVoxelExtentsScale", &GeoWrapper::getVoxelExtentsScale)
    .def("getCurrPose", [](const GeoWrapper& g) {
      py::array_t<float> a({4, 4});
      std::memcpy(a.mutable_data(), g.getCurrPose().data(), 16 * sizeof(float));
      return a;
    })
    .def("getPointCloud", [](const GeoWrapper& g) { return to_array<float>(g.pointCloud(), 3); })
    .def("getNormals", [](const GeoWrapper& g) { return to_array<float>(g.normals(), 3); })
    .def("getVertices", [](const GeoWrapper& g) { return to_array<double>(g.vertices(), 3); })
    .def("getFaces", [](const GeoWrapper& g) { return to_array<int32_t>(g.faces(), 3); })
    .def("getColors", [](const GeoWrapper& g) { return to_array<double>(g.colors(), 3); })
    // setters
    .def("setHashNumBuckets", &GeoWrapper::setHashNumBuckets)
    .def("setNumSdfBlocks", &GeoWrapper::setNumSdfBlocks)
    .def("setHashBucketSize", &GeoWrapper::setHashBucketSize)
    .def("setSdfTruncation", &GeoWrapper::setSdfTruncation)
    .def("setSdfTruncationScale", &GeoWrapper::setSdfTruncationScale)
    .def("setIntegrationWeightSample", &GeoWrapper::setIntegrationWeightSample)
    .def("setIntegrationWeightMax", &GeoWrapper::setIntegrationWeightMax)
    .def("setVirtualVoxelSize", &GeoWrapper::setVirtualVoxelSize)
    .def("setLinkedListSize", &GeoWrapper::setLinkedListSize)
    .def("setNFramesInvalidateVoxels", &GeoWrapper::setNFramesInvalidateVoxels)
    .def("setMaxNumSdfBlockIntegrateFromGlobalHash", &GeoWrapper::setMaxNumSdfBlockIntegrateFromGlobalHash)
    .def("setVoxelExtentsScale", &GeoWrapper::setVoxelExtentsScale)
    .def("setRGBImage", [](GeoWrapper& g, py::array_t<uint8_t, py::array::c_style | py::array::forcecast> a) {
      // geowrapper.cpp:246-274 (the reference runner passes float32 RGB and relies on the implicit cast)
      if (a.ndim() != 3) throw std::runtime_error("GeoWrapper::setRGBImage|input should be a 3D numpy array");
      if (a.shape(2) != 3) throw std::runtime_error("GeoWrapper::setRGBImage|input should have 3 channels");
      g.setRGBImage(a.data(), (size_t) a.shape(0), (size_t) a.shape(1));
    })
    .def("setDepthImage", [](GeoWrapper& g, py::array_t<float, py::array::c_style | py::array::forcecast> a) {
      // geowrapper.cpp:300-321
      if (a.ndim() != 2) throw std::runtime_error("GeoWrapper::setDepthImage|input should be a 2D numpy array");
      g.setDepthImage(a.data(), (size_t) a.shape(0), (size_t) a.shape(1));
    })
    .def("setPointCloud", [](GeoWrapper& g, py::array_t<float, py::array::c_style | py::array::forcecast> pts, bool compute_normals) {
      // geowrapper.cpp:345-405: [N, 3] points in the sensor frame (the runners pass points[:, :3]); copied
      if (pts.ndim() != 2) throw std::runtime_error("GeoWrapper::setPointCloud|input should be a 2D numpy array");
      if (pts.shape(1) < 3) throw std::runtime_error("GeoWrapper::setPointCloud|input should have at least 3 columns (x, y, z)");
      if (compute_normals) throw std::runtime_error("GeoWrapper::setPointCloud|normal estimation (MAD tree) is outside this library's scope");
      if (pts.shape(1) == 3) {
        g.setPointCloud(pts.data(), (size_t) pts.shape(0), nullptr);
      } else {
        std::vector<float> xyz((size_t) pts.shape(0) * 3);
        auto r = pts.unchecked<2>();
        for (py::ssize_t i = 0; i < pts.shape(0); i++) { xyz[3 * i] = r(i, 0); xyz[3 * i + 1] = r(i, 1); xyz[3 * i + 2] = r(i, 2); }
        g.setPointCloud(xyz.data(), (size_t) pts.shape(0), nullptr);
      }
    }, py::arg("input_point_cloud"), py::arg("compute_normals") = false)
    .def("setPointCloud", [](GeoWrapper& g, py::array_t<float, py::array::c_style | py::array::forcecast> pts,
                             py::array_t<float, py::array::c_style | py::array::forcecast> normals) {
      // geowrapper.cpp:460-490
      if (pts.ndim() != 2) throw std::runtime_error("GeoWrapper::setPointCloud|point cloud input should be a 2D numpy array");
      if (normals.ndim() != 2) throw std::runtime_error("GeoWrapper::setPointCloud|normals input should be a 2D numpy array");
      if (pts.shape(0) != normals.shape(0))
        throw std::runtime_error("GeoWrapper::setPointCloud|point_cloud input and normals input should have the same number of points");
      g.setPointCloud(pts.data(), (size_t) pts.shape(0), normals.data());
    })
    .def("setCamera", &GeoWrapper::setCamera)
    .def("setCurrPose", [](GeoWrapper& g, py::array_t<float, py::array::c_style | py::array::forcecast> t,
                           py::array_t<float, py::array::c_style | py::array::forcecast> q) {
      if (t.size() != 3 || q.size() != 4) throw std::runtime_error("GeoWrapper::setCurrPose|expected a 3-vector and a 4-vector (qx,qy,qz,qw)");
      g.setCurrPose({t.data()[0], t.data()[1], t.data()[2]}, {q.data()[0], q.data()[1], q.data()[2], q.data()[3]});
    })
    .def("setCameraInLidar", [](GeoWrapper& g, py::array_t<float, py::array::c_style | py::array::forcecast> m4) {
      if (m4.size() != 16) throw std::runtime_error("GeoWrapper::setCameraInLidar|expected a 4x4 matrix");
      std::array<float, 16> a;
      std::memcpy(a.data(), m4.data(), 16 * sizeof(float));
      g.setCameraInLidar(a);
    })
    .def("compute", &GeoWrapper::compute)
    .def("extractMesh", &GeoWrapper::extractMesh)
    .def("GSSavePointCloud", &GeoWrapper::GSSavePointCloud)
    .def("GSFinalOpt", &GeoWrapper::GSFinalOpt)
    .def("streamAllOut", &GeoWrapper::streamAllOut)
    // not part of the reference binding: lets tests drive and observe the streamer directly
    .def("_stream", [](GeoWrapper& g, py::array_t<float, py::array::c_style | py::array::forcecast> pos, float radius) {
      if (pos.size() != 3) throw std::runtime_error("GeoWrapper::_stream|expected a 3-vector");
      g.stream({pos.data()[0], pos.data()[1], pos.data()[2]}, radius);
    })
    .def("_hostGridBlocks", &GeoWrapper::hostGridBlocks)
    .def("_setSyncCompute", &GeoWrapper::setSyncCompute)
    .def("_lastComputeFlags", &GeoWrapper::lastComputeFlags)
    // multi-GPU (include/mrhash_comm.h, no reference counterpart): RCCL behind the C ABI, one GeoWrapper per rank
    .def_static("_commUniqueId", []() {
      const auto id = GeoWrapper::commUniqueId();
      return py::bytes((const char*) id.data(), id.size());
    })
    .def("_commInit", [](GeoWrapper& g, py::bytes id, int rank, int world, int chunk_log2, bool tile_sharded) {
      const std::string b = id;
      if (b.size() != MRH_COMM_ID_BYTES) throw std::runtime_error("GeoWrapper::_commInit|the id has 128 bytes");
      std::array<uint8_t, MRH_COMM_ID_BYTES> a;
      std::memcpy(a.data(), b.data(), a.size());
      g.commInit(a, rank, world, chunk_log2, tile_sharded);
    }, py::arg("id"), py::arg("rank"), py::arg("world"), py::arg("chunk_log2") = 3, py::arg("tile_sharded") = true)
    .def("_mergeSubmaps", &GeoWrapper::mergeSubmaps)
    // splat seeds accumulated so far (what the reference hands to GaussianModel::Add_gaussians): (xyz, scale, rgb)
    .def("_splatSeeds", [](const GeoWrapper& g) {
      const auto& seeds = g.splatSeeds();
      std::vector<float> xyz(seeds.size() * 3), scale(seeds.size());
      std::vector<uint8_t> rgb(seeds.size() * 3);
      for (size_t i = 0; i < seeds.size(); i++) {
        for (int k = 0; k < 3; k++) { xyz[3 * i + k] = seeds[i].p[k]; rgb[3 * i + k] = seeds[i].rgb[k]; }
        scale[i] = seeds[i].scale;
      }
      return py::make_tuple(to_array<float>(xyz, 3), to_array<float>(scale, 1), to_array<uint8_t>(rgb, 3));
    })
    .def("clearBuffers", &GeoWrapper::clearBuffers)
    .def("serializeData", &GeoWrapper::serializeData, py::arg("filename_hash") = "./data/hash_points.ply",
         py::arg("filename_voxel") = "./data/voxel_points.ply")
    .def("serializeGrid", &GeoWrapper::serializeGrid, py::arg("filename") = "./data/grid.bin")
    .def("deserializeGrid", &GeoWrapper::deserializeGrid, py::arg("filename") = "./data/grid.bin");
}
