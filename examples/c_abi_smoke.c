/* Plain-C user of the boundary (include/mrhash_hip.h): no C++, no Python, no torch.
 *   gcc -std=c11 -Iinclude examples/c_abi_smoke.c -o c_abi_smoke -Lmrhash_amd/csrc -lmrhash_hip -Wl,-rpath,$PWD/mrhash_amd/csrc -lm
 * Fuses the plane frame of tests/golden/cfg1_golden.json ("plane_1frame": 128x128, K = (128, 128, 64, 64), identity pose,
 * depth 1.0 m; the colour does not enter the counts) and prints the counts the fixture pins (100 blocks, 31081 weighted
 * voxels, 4608 triangles). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mrhash_hip.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_ != MRH_OK) {                                                                 \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mrh_last_error(ctx));          \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

int main(void) {
  mrh_ctx* ctx = NULL;
  mrh_params p;
  memset(&p, 0, sizeof p);
  p.abi_version = MRH_ABI_VERSION;
  p.sdf_truncation = 0.06f;
  p.integration_weight_sample = 1;
  p.integration_weight_max = 255;
  p.virtual_voxel_size = 0.02f;
  p.n_frames_invalidate_voxels = 0;
  p.voxel_extents_scale = 1;
  p.marching_cubes_threshold = 1.5f;
  p.min_weight_threshold = 5;
  p.projective_sdf = 1;
  p.min_depth = 0.01f;
  p.max_depth = 30.0f;
  p.num_sdf_blocks = 16384;
  p.shard_count = 1;
  if (mrh_create(&p, &ctx) != MRH_OK) {
    fprintf(stderr, "mrh_create: %s\n", mrh_last_error(NULL));
    return 1;
  }
  printf("%s\n", mrh_version());
  const int rows = 128, cols = 128;
  CHECK(mrh_set_camera(ctx, 128.f, 128.f, 64.f, 64.f, rows, cols, 0.01f, 30.0f, MRH_CAMERA_PINHOLE));
  const float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  CHECK(mrh_set_pose(ctx, R, t));
  float* depth = (float*) malloc(sizeof(float) * rows * cols);
  uint8_t* rgb = (uint8_t*) malloc(3 * rows * cols);
  for (int i = 0; i < rows * cols; i++) depth[i] = 1.0f;
  memset(rgb, 128, 3 * rows * cols);
  CHECK(mrh_upload_depth(ctx, depth, rows, cols));
  CHECK(mrh_upload_rgb(ctx, rgb, rows, cols));
  CHECK(mrh_integrate(ctx, -1));
  CHECK(mrh_sync(ctx));
  mrh_stats st;
  CHECK(mrh_get_stats(ctx, &st));
  uint64_t n = 0;
  CHECK(mrh_dump_blocks(ctx, NULL, NULL, 0, &n));
  mrh_block_desc* descs = (mrh_block_desc*) malloc(sizeof(mrh_block_desc) * n);
  mrh_voxel* vox = (mrh_voxel*) malloc(sizeof(mrh_voxel) * 512 * n);
  CHECK(mrh_dump_blocks(ctx, descs, vox, n, &n));
  uint64_t weighted = 0;
  for (uint64_t i = 0; i < n * 512; i++) weighted += vox[i].weight > 0;
  const mrh_triangle* tris = NULL;
  uint64_t nt = 0;
  CHECK(mrh_extract_triangles(ctx, &tris, &nt));
  const double *V, *C;
  const int32_t* F;
  uint64_t nv = 0, nf = 0;
  CHECK(mrh_extract_mesh(ctx, &V, &nv, &F, &nf, &C));
  printf("blocks %llu weighted_voxels %llu triangles %llu vertices %llu faces %llu free_fine %lld\n", (unsigned long long) n,
         (unsigned long long) weighted, (unsigned long long) nt, (unsigned long long) nv, (unsigned long long) nf, (long long) st.free_fine);
  free(depth); free(rgb); free(descs); free(vox);
  mrh_destroy(ctx);
  return 0;
}
