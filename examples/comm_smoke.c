/* First contact with more than one GPU, without Python: N processes (one per GPU) drive include/mrhash_comm.h in plain C.
 *
 *   gcc -std=c11 -O1 -Iinclude examples/comm_smoke.c -o comm_smoke -Lmrhash_amd/csrc -lmrhash_hip -Wl,-rpath,$PWD/mrhash_amd/csrc -lm
 *   ./comm_smoke 8            starts 8 copies of itself (RANK / WORLD_SIZE in their environment), rank r on device r
 *   RANK=r WORLD_SIZE=n MRH_SMOKE_ID_FILE=/shared/path ./comm_smoke     one rank, as a launcher (mpirun, srun) would start it
 *   MRH_COMM_SELF_LOOP=1 ./comm_smoke 1      a one-GPU box: the rank's own parts travel through ncclSend / ncclRecv to itself
 *
 * Every rank fuses its own four frames of a slanted wall into a sub-map (frame sharding), then
 *   mrh_comm_merge_submaps   all-to-all of blocks to their tile owner + weighted merge        (per-pair bytes printed)
 *   mrh_comm_exchange_halo   boundary blocks of every rank to every rank                      (per-rank bytes printed)
 *   mrh_comm_gather_mesh     per-rank marching cubes -> root                                  (triangle total printed)
 * and checks the merged map against ONE context that fused all the frames, on every rank for the blocks it owns: the same
 * positions (a position-hash summed over the ranks must equal the single map's), the same weights voxel for voxel, TSDF
 * values within 1e-5 (the fold's association order differs from the running mean's).  The scene keeps every block in
 * every sub-map from the first frame on and the frames carry one colour, so the merged weights and colours are exact
 * (DESIGN.md 5 says where that does not hold).  Exit code 0 = every rank passed.  A step that does not return is a hang
 * in that step: every line is flushed before the next collective starts. */
#define _DEFAULT_SOURCE
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "mrhash_comm.h"

static int g_rank = 0;
#define SAY(...) do { printf("[rank %d] ", g_rank); printf(__VA_ARGS__); printf("\n"); fflush(stdout); } while (0)
#define CHECK(ctx, call)                                                                                     \
  do {                                                                                                       \
    int rc_ = (call);                                                                                        \
    if (rc_ != MRH_OK) { SAY("FAILED %s (%d): %s", #call, rc_, mrh_last_error(ctx)); return 1; }              \
  } while (0)
#define CHECKM(comm, call)                                                                                   \
  do {                                                                                                       \
    int rc_ = (call);                                                                                        \
    if (rc_ != MRH_OK) { SAY("FAILED %s (%d): %s", #call, rc_, mrh_comm_last_error(comm)); return 1; }        \
  } while (0)

enum { ROWS = 128, COLS = 128, FRAMES_PER_RANK = 4 };

static void params(mrh_params* p, int device) {
  memset(p, 0, sizeof *p);
  p->abi_version = MRH_ABI_VERSION;
  p->sdf_truncation = 0.06f;
  p->integration_weight_sample = 1;
  p->integration_weight_max = 255;
  p->virtual_voxel_size = 0.02f;
  p->n_frames_invalidate_voxels = 0; /* no garbage collection: every sub-map keeps every block */
  p->voxel_extents_scale = 1;
  p->marching_cubes_threshold = 1.5f;
  p->min_weight_threshold = 1;
  p->projective_sdf = 1;
  p->min_depth = 0.01f;
  p->max_depth = 30.0f;
  p->num_sdf_blocks = 16384;
  p->shard_count = 1;
  p->device_id = device;
}

/* frame k of the stream: a wall slanted in x, the same for every k (so that every sub-map holds every block), one colour */
static void frame(float* depth, uint8_t* rgb) {
  for (int r = 0; r < ROWS; r++)
    for (int c = 0; c < COLS; c++) depth[r * COLS + c] = 1.0f + 0.002f * (float) c;
  memset(rgb, 150, 3 * ROWS * COLS);
}

static int fuse(mrh_ctx* ctx, int n, const float* depth, const uint8_t* rgb) {
  const float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  for (int k = 0; k < n; k++) {
    CHECK(ctx, mrh_set_pose(ctx, R, t));
    CHECK(ctx, mrh_upload_depth(ctx, depth, ROWS, COLS));
    CHECK(ctx, mrh_upload_rgb(ctx, rgb, ROWS, COLS));
    CHECK(ctx, mrh_integrate(ctx, -1));
  }
  CHECK(ctx, mrh_sync(ctx));
  return 0;
}

static uint64_t pos_hash(const mrh_block_desc* d) {
  uint64_t h = (uint64_t) (uint32_t) d->x * 0x9E3779B97F4A7C15ull;
  h ^= ((uint64_t) (uint32_t) d->y + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
  h ^= ((uint64_t) (uint32_t) d->z + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
  return h * 0xD6E8FEB86659FD93ull;
}
typedef struct { mrh_block_desc d; uint64_t i; } keyed;  /* a block and where its voxels sit in the dump */
static int cmp_desc(const void* a, const void* b) {
  const mrh_block_desc *p = &((const keyed*) a)->d, *q = &((const keyed*) b)->d;
  if (p->x != q->x) return p->x < q->x ? -1 : 1;
  if (p->y != q->y) return p->y < q->y ? -1 : 1;
  if (p->z != q->z) return p->z < q->z ? -1 : 1;
  return 0;
}

static int dump(mrh_ctx* ctx, mrh_block_desc** descs, mrh_voxel** vox, uint64_t* n) {
  CHECK(ctx, mrh_dump_blocks(ctx, NULL, NULL, 0, n));
  *descs = (mrh_block_desc*) malloc(sizeof(mrh_block_desc) * (*n + 1));
  *vox = (mrh_voxel*) malloc(sizeof(mrh_voxel) * 512 * (*n + 1));
  CHECK(ctx, mrh_dump_blocks(ctx, *descs, *vox, *n, n)); /* in no particular order */
  return 0;
}

static int run_rank(int rank, int world, const char* id_file) {
  g_rank = rank;
  /* ---- the 128-byte id: rank 0 creates and publishes it, the others poll */
  uint8_t id[MRH_COMM_ID_BYTES];
  if (rank == 0) {
    CHECKM(NULL, mrh_comm_unique_id(id));
    char tmp[4096];
    snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
    FILE* f = fopen(tmp, "wb");
    if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { SAY("FAILED writing %s", tmp); return 1; }
    fclose(f);
    chmod(tmp, 0600);
    if (rename(tmp, id_file)) { SAY("FAILED rename to %s", id_file); return 1; }
  } else {
    const time_t t0 = time(NULL);
    for (;;) {
      FILE* f = fopen(id_file, "rb");
      const size_t got = f ? fread(id, 1, sizeof id, f) : 0;
      if (f) fclose(f);
      if (got == sizeof id) break;
      if (time(NULL) - t0 > 120) { SAY("FAILED: rank 0 never published %s", id_file); return 1; }
      usleep(10000);
    }
  }
  /* rank r on device r; MRH_SMOKE_DEVICE overrides (e.g. 0 for a one-GPU box) */
  const int device = getenv("MRH_SMOKE_DEVICE") ? atoi(getenv("MRH_SMOKE_DEVICE")) : rank;
  SAY("mrh_comm_create(rank %d of %d, device %d) ...", rank, world, device);
  mrh_comm* comm = NULL;
  CHECKM(NULL, mrh_comm_create(id, rank, world, device, &comm));
  SAY("communicator up (%s)", mrh_version());
  {
    mrh_comm_status_info st;
    CHECKM(comm, mrh_comm_status(comm, &st));
    SAY("RCCL says: %d ranks, this is rank %d on device %d, version %d, async error %d (%s), library %s", st.rccl_ranks, st.rccl_rank, st.rccl_device,
        st.rccl_version, st.async_error, st.async_error_string, st.library_path);
    if (st.rccl_ranks != world || st.rccl_rank != rank) { SAY("FAILED: RCCL's view of the communicator differs from the launch (%d ranks, rank %d)", world, rank); return 1; }
  }
  CHECKM(comm, mrh_comm_barrier(comm));
  if (rank == 0) unlink(id_file);

  float* depth = (float*) malloc(sizeof(float) * ROWS * COLS);
  uint8_t* rgb = (uint8_t*) malloc(3 * ROWS * COLS);
  frame(depth, rgb);
  mrh_params p;
  params(&p, device);

  /* ---- the single map every rank compares with: all world * 4 frames through one context */
  mrh_ctx* single = NULL;
  if (mrh_create(&p, &single) != MRH_OK) { SAY("FAILED mrh_create: %s", mrh_last_error(NULL)); return 1; }
  CHECK(single, mrh_set_camera(single, 128.f, 128.f, 64.f, 64.f, ROWS, COLS, 0.01f, 30.0f, MRH_CAMERA_PINHOLE));
  if (fuse(single, world * FRAMES_PER_RANK, depth, rgb)) return 1;
  mrh_block_desc* sd; mrh_voxel* sv; uint64_t sn = 0;
  if (dump(single, &sd, &sv, &sn)) return 1;
  uint64_t single_hash = 0;
  keyed* sk = (keyed*) malloc(sizeof(keyed) * (sn + 1));
  for (uint64_t i = 0; i < sn; i++) { single_hash += pos_hash(&sd[i]); sk[i].d = sd[i]; sk[i].i = i; }
  qsort(sk, sn, sizeof *sk, cmp_desc);

  /* ---- this rank's sub-map */
  mrh_ctx* ctx = NULL;
  if (mrh_create(&p, &ctx) != MRH_OK) { SAY("FAILED mrh_create: %s", mrh_last_error(NULL)); return 1; }
  CHECK(ctx, mrh_set_camera(ctx, 128.f, 128.f, 64.f, 64.f, ROWS, COLS, 0.01f, 30.0f, MRH_CAMERA_PINHOLE));
  CHECK(ctx, mrh_comm_attach(ctx, comm));
  if (fuse(ctx, FRAMES_PER_RANK, depth, rgb)) return 1;
  SAY("sub-map fused; mrh_comm_merge_submaps ...");
  mrh_comm_merge_info info;
  CHECK(ctx, mrh_comm_merge_submaps(ctx, 1, &info));
  mrh_comm_phases ph;
  CHECK(ctx, mrh_comm_phase_times(ctx, &ph));
  SAY("merge: sent %llu blocks (%llu B), received %llu, kept %llu | pack %.3f counts %.3f collective %.3f unpack %.3f ms",
      (unsigned long long) info.blocks_sent, (unsigned long long) info.bytes_sent, (unsigned long long) info.blocks_received,
      (unsigned long long) info.blocks_kept, ph.pack_ms, ph.counts_ms, ph.collective_ms, ph.unpack_ms);

  /* ---- merged map vs single map, on the blocks this rank owns */
  mrh_block_desc* md; mrh_voxel* mv; uint64_t mn = 0;
  if (dump(ctx, &md, &mv, &mn)) return 1;
  uint64_t my_hash = 0, bad_w = 0, bad_c = 0, missing = 0;
  double max_d = 0.0;
  for (uint64_t i = 0; i < mn; i++) {
    my_hash += pos_hash(&md[i]);
    keyed probe;
    probe.d = md[i]; probe.i = 0;
    const keyed* hit = (const keyed*) bsearch(&probe, sk, sn, sizeof *sk, cmp_desc);
    if (!hit) { missing++; continue; }
    const mrh_voxel *a = mv + 512 * i, *b = sv + 512 * hit->i;
    for (int v = 0; v < 512; v++) {
      bad_w += a[v].weight != b[v].weight;
      bad_c += a[v].weight && memcmp(a[v].rgb, b[v].rgb, 3) != 0;
      if (a[v].weight && b[v].weight) { const double d = fabs((double) a[v].sdf - (double) b[v].sdf); if (d > max_d) max_d = d; }
    }
  }
  uint64_t mine[2] = {my_hash, mn};
  uint64_t* all = (uint64_t*) malloc(sizeof(uint64_t) * 2 * (size_t) world);
  CHECKM(comm, mrh_comm_allgather_bytes(comm, mine, sizeof mine, all));
  uint64_t sum_hash = 0, sum_n = 0;
  for (int r = 0; r < world; r++) { sum_hash += all[2 * r]; sum_n += all[2 * r + 1]; }
  const int map_ok = sum_hash == single_hash && sum_n == sn && !missing && !bad_w && !bad_c && max_d <= 1e-5;
  SAY("merged map: %llu owned blocks (all ranks %llu, single map %llu), position checksum %s, %llu missing, %llu weights / %llu colours differ, max |sdf diff| %.3g -> %s",
      (unsigned long long) mn, (unsigned long long) sum_n, (unsigned long long) sn, sum_hash == single_hash ? "equal" : "DIFFERENT",
      (unsigned long long) missing, (unsigned long long) bad_w, (unsigned long long) bad_c, max_d, map_ok ? "ok" : "MISMATCH");

  /* ---- halo exchange + sharded extraction */
  SAY("mrh_comm_exchange_halo ...");
  uint64_t taken = 0;
  CHECK(ctx, mrh_comm_exchange_halo(ctx, &taken));
  CHECK(ctx, mrh_comm_phase_times(ctx, &ph));
  SAY("halo: took %llu blocks, %llu B out / %llu B in | pack %.3f counts %.3f collective %.3f unpack %.3f ms", (unsigned long long) taken,
      (unsigned long long) ph.bytes_out, (unsigned long long) ph.bytes_in, ph.pack_ms, ph.counts_ms, ph.collective_ms, ph.unpack_ms);
  SAY("mrh_comm_gather_mesh ...");
  uint64_t nt = 0;
  CHECK(ctx, mrh_comm_gather_mesh(ctx, 0, &nt));
  int mesh_ok = 1;
  if (rank == 0) {
    const mrh_triangle* tris = NULL;
    uint64_t nt1 = 0;
    CHECK(single, mrh_extract_triangles(single, &tris, &nt1));
    /* the merged map's values differ from the single map's in their last bits, so the triangle COUNT is compared with a
     * tolerance of the voxels whose sign can flip, not the bytes: parity of the sharded extraction itself (bit for bit
     * against a single context on the same map) is tests/test_sharding*.py's business */
    mesh_ok = nt > 0 && llabs((long long) nt - (long long) nt1) <= (long long) (nt1 / 50 + 8);
    SAY("mesh on root: %llu triangles from %d ranks (single context: %llu) -> %s", (unsigned long long) nt, world, (unsigned long long) nt1, mesh_ok ? "ok" : "MISMATCH");
  }
  CHECK(ctx, mrh_drop_blocks(ctx, MRH_DROP_HALO, NULL));
  CHECK(ctx, mrh_comm_attach(ctx, NULL));
  mrh_destroy(ctx);
  mrh_destroy(single);
  CHECKM(comm, mrh_comm_barrier(comm));
  CHECKM(comm, mrh_comm_destroy(comm));
  free(depth); free(rgb); free(sd); free(sv); free(sk); free(md); free(mv); free(all);
  SAY("%s", map_ok && mesh_ok ? "PASS" : "FAIL");
  return map_ok && mesh_ok ? 0 : 1;
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IOLBF, 0);
  if (getenv("RANK") && getenv("WORLD_SIZE")) {
    const char* idf = getenv("MRH_SMOKE_ID_FILE");
    if (!idf) { fprintf(stderr, "comm_smoke: MRH_SMOKE_ID_FILE is not set\n"); return 2; }
    return run_rank(atoi(getenv("RANK")), atoi(getenv("WORLD_SIZE")), idf);
  }
  /* launcher: N copies of this program, started BEFORE anything touches the HIP runtime (no fork after initialisation) */
  const int world = argc > 1 ? atoi(argv[1]) : 1;
  if (world < 1 || world > 64) { fprintf(stderr, "usage: comm_smoke [ranks]\n"); return 2; }
  char idf[256], w[16];
  const char* dir = getenv("XDG_RUNTIME_DIR") ? getenv("XDG_RUNTIME_DIR") : "/tmp";
  snprintf(idf, sizeof idf, "%s/mrh_comm_smoke_%d_%ld.id", dir, (int) getpid(), (long) time(NULL));
  unlink(idf);
  snprintf(w, sizeof w, "%d", world);
  setenv("WORLD_SIZE", w, 1);
  setenv("MRH_SMOKE_ID_FILE", idf, 1);
  pid_t pids[64];
  for (int r = 0; r < world; r++) {
    pids[r] = fork();
    if (pids[r] == 0) {
      char rk[16];
      snprintf(rk, sizeof rk, "%d", r);
      setenv("RANK", rk, 1);
      execv("/proc/self/exe", argv);
      perror("execv");
      _exit(127);
    }
  }
  int bad = 0;
  for (int r = 0; r < world; r++) {
    int st = 0;
    waitpid(pids[r], &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "comm_smoke: rank %d ended with status 0x%x\n", r, st); bad++; }
  }
  unlink(idf);
  printf("comm_smoke: %d of %d ranks passed\n", world - bad, world);
  return bad ? 1 : 0;
}
