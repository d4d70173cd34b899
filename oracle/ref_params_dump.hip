// oracle/_ref/ref_params_dump — the reference's params.h (mrhash/src/sdf/params.h, compiled where it lies, untouched) for
// gfx950: its marching-cubes tables are `__device__ const` arrays, so the only way to read the values THE REFERENCE'S OWN
// SOURCE defines — rather than a transcription of them — is a kernel.  hipcc provides `__device__` and `int3` natively;
// nothing else is needed.  Prints one JSON object; run by tests/test_parity_gpu.py on the GPU box (the binary is built here,
// where /root/reference exists, and travels with the snapshot).  TEST INFRASTRUCTURE ONLY; this driver is own code.
#include <hip/hip_runtime.h>

#include "params.h"

#include <cstdio>
#include <vector>

__global__ void k_dump(unsigned char* cls, unsigned char* data, unsigned short* vert, int* offs) {
  const int i = threadIdx.x;
  cls[i] = regularCellClass[i];
  if (i < 16) {
    data[i * 16] = regularCellData[i].geometryCounts;
    for (int k = 0; k < 15; k++) data[i * 16 + 1 + k] = regularCellData[i].vertexIndex[k];
  }
  for (int k = 0; k < 12; k++) vert[i * 12 + k] = regularVertexData[i][k];
  if (i < (int) vertex_offset_camera) { offs[i * 3] = vert_offset[i].x; offs[i * 3 + 1] = vert_offset[i].y; offs[i * 3 + 2] = vert_offset[i].z; }
}

int main() {
  unsigned char *cls, *data;
  unsigned short* vert;
  int* offs;
  if (hipMalloc((void**) &cls, 256) != hipSuccess || hipMalloc((void**) &data, 256) != hipSuccess || hipMalloc((void**) &vert, 256 * 12 * 2) != hipSuccess ||
      hipMalloc((void**) &offs, 8 * 3 * 4) != hipSuccess) {
    fprintf(stderr, "ref_params_dump: no HIP device\n");
    return 2;
  }
  hipLaunchKernelGGL(k_dump, dim3(1), dim3(256), 0, 0, cls, data, vert, offs);
  std::vector<unsigned char> h_cls(256), h_data(256);
  std::vector<unsigned short> h_vert(256 * 12);
  std::vector<int> h_offs(24);
  if (hipMemcpy(h_cls.data(), cls, 256, hipMemcpyDeviceToHost) != hipSuccess) return 3;
  (void) hipMemcpy(h_data.data(), data, 256, hipMemcpyDeviceToHost);
  (void) hipMemcpy(h_vert.data(), vert, 256 * 12 * 2, hipMemcpyDeviceToHost);
  (void) hipMemcpy(h_offs.data(), offs, 24 * 4, hipMemcpyDeviceToHost);
  printf("{\"p\": [%d, %d, %d], \"sdf_block_size\": %u, \"hash_bucket_size\": %u, \"linked_list_size\": %u, \"integration_weight_max\": %u, "
         "\"max_dda_iteration_count\": %u, \"n_threads\": %u, \"float_epsilon\": %.9g, \"stream_threshold\": %.9g,\n",
         p0, p1, p2, sdf_block_size, hash_bucket_size, linked_list_size, integration_weight_max, max_dda_iteration_count, n_threads, (double) FLOAT_EPSILON,
         (double) stream_threshold);
  auto arr = [](const char* name, const auto& v, const char* tail) {
    printf(" \"%s\": [", name);
    for (size_t i = 0; i < v.size(); i++) printf("%s%d", i ? ", " : "", (int) v[i]);
    printf("]%s\n", tail);
  };
  arr("regularCellClass", h_cls, ",");
  arr("regularCellData", h_data, ",");
  arr("regularVertexData", h_vert, ",");
  arr("vert_offset", h_offs, "}");
  return 0;
}
