/*
 * mrhash_comm.h — multi-GPU entry points of libmrhash_hip.so: RCCL over xGMI behind the C ABI.
 *
 * The reference is single-GPU (no NCCL / MPI anywhere under mrhash/src), so nothing here replaces a reference
 * interface; this is BASELINE.json's "frames (or disjoint hash-space tiles) shard across the 8 GPUs of one node with
 * an RCCL all-gather of boundary voxels over xGMI".  One process per GPU.  The library opens the RCCL that belongs to
 * the HIP runtime it is itself linked against (librccl.so.1 next to libamdhip64.so.7, i.e. /opt/rocm/lib) the first
 * time one of these functions is called, so that a process holds ONE HIP runtime and every buffer RCCL is handed was
 * allocated by it; collectives run on the context's own stream, ordered with its kernels, on the library's own device
 * buffers — nothing is staged through the host and no other framework is involved.  A C++ GeoWrapper or a plain C
 * program shards through these calls exactly as mrhash_amd/parallel.py does.
 *
 * Two ways to use N GPUs:
 *   tile sharding  (mrh_params.shard_* / mrh_set_sharding): every rank sees every frame and fuses the blocks of the
 *                  chunks it owns; the union of the N maps is bit-identical to the single-GPU map.  With a communicator
 *                  attached, mrh_integrate runs the two MIN all-reduces of a starve frame itself (it never returns
 *                  MRH_PENDING_EXCHANGE); mrh_comm_exchange_halo brings the boundary blocks of the other ranks in before
 *                  marching cubes; mrh_comm_gather_mesh merges the per-rank extractions on one rank.
 *   frame sharding every rank fuses its own frames into its own sub-map; mrh_comm_merge_submaps folds the N sub-maps into
 *                  one tile-sharded map (all-to-all of blocks to their owner + weighted merge on the device).
 *
 * Conventions as in mrhash_hip.h: int status (0 = MRH_OK, negative = mrh_status), message through mrh_last_error(ctx) for
 * the calls that take a context and mrh_comm_last_error(comm) for the others.  All calls are collective: every rank of the
 * communicator makes the same call in the same order.
 */
#ifndef MRHASH_COMM_H
#define MRHASH_COMM_H

#include "mrhash_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MRH_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

typedef struct mrh_comm mrh_comm;

/* ncclGetUniqueId: one rank creates the id, the host distributes the 128 bytes to the others by whatever means it has
 * (a file on a shared path, MPI, a socket; mrhash_amd/parallel.py uses a file under /tmp keyed by the launcher). */
int mrh_comm_unique_id(uint8_t out_id[MRH_COMM_ID_BYTES]);

/* ncclCommInitRank on HIP device `device_id`.  Collective; blocks until all `world` ranks have joined — or for
 * MRH_COMM_INIT_TIMEOUT_S seconds (environment, default 180): past that the call fails with MRH_ERR_DEVICE (a peer never joined)
 * instead of keeping the process for ever; the caller falls back to a run without RCCL.  After such a timeout the init thread is
 * still inside ncclCommInitRank and stays there for the life of the process: do NOT create another communicator with the same id
 * in this process, and leave through _exit() (or quick_exit) rather than through static destructors — RCCL's and HIP's teardown
 * must not run under a thread that is still inside RCCL (bench.py does exactly that; tests let the timeout fire in a process of
 * its own). */
int mrh_comm_create(const uint8_t id[MRH_COMM_ID_BYTES], int rank, int world, int device_id, mrh_comm** out_comm);
int mrh_comm_destroy(mrh_comm* comm); /* NULL is a no-op; contexts must be detached (or destroyed) first */
const char* mrh_comm_last_error(const mrh_comm* comm); /* comm == NULL: last failing mrh_comm_create / _unique_id on this thread */
int mrh_comm_size(const mrh_comm* comm, int* out_rank, int* out_world);

/* What RCCL itself reports about the communicator: ncclCommCount / ncclCommUserRank / ncclCommCuDevice, the pending asynchronous
 * error (ncclCommGetAsyncError: 0 = ncclSuccess), ncclGetVersion and the path of the librccl the library opened.  Fields RCCL
 * cannot answer are -1.  Local (not collective). */
typedef struct mrh_comm_status_info {
  int  rccl_ranks;             /* ncclCommCount                                   */
  int  rccl_rank;              /* ncclCommUserRank                                */
  int  rccl_device;            /* ncclCommCuDevice (HIP device index)             */
  int  rccl_version;           /* ncclGetVersion, e.g. 22204                      */
  int  async_error;            /* ncclResult_t of ncclCommGetAsyncError (0 = none) */
  char async_error_string[64]; /* ncclGetErrorString of the above                 */
  char library_path[256];      /* the librccl.so the library opened               */
} mrh_comm_status_info;
int mrh_comm_status(mrh_comm* comm, mrh_comm_status_info* out);

/* Host-side helpers for the driver of a multi-rank run (timing brackets, counts): tiny collectives staged through a device
 * buffer on the communicator's own stream.  Blocking. */
int mrh_comm_barrier(mrh_comm* comm);
typedef enum mrh_comm_op { MRH_COMM_SUM = 0, MRH_COMM_MAX = 1, MRH_COMM_MIN = 2 } mrh_comm_op;
int mrh_comm_allreduce_f64(mrh_comm* comm, double* inout, uint64_t n, int op);
int mrh_comm_allgather_bytes(mrh_comm* comm, const void* send, uint64_t bytes_per_rank, void* recv /* world * bytes_per_rank */);

/* Binds a context to a communicator (same device).  comm == NULL detaches.  While attached, a tile-sharded context
 * (shard_count > 1) min-reduces the starve z-buffer over the ranks inside mrh_integrate / mrh_integrate_points:
 * ncclAllReduce(int64, MIN) enqueued on the context's stream between the starve passes — no host synchronisation, no
 * MRH_PENDING_EXCHANGE.  shard_count must equal the communicator's size for that. */
int mrh_comm_attach(mrh_ctx* ctx, mrh_comm* comm);

/* Boundary blocks of every rank to every rank, device to device: the owned blocks on the surface of their chunk are
 * packed (MRH_PACK_HALO), the counts exchanged, and each rank's records travel straight to each peer — grouped
 * ncclSend / ncclRecv, one direct xGMI link per pair, no padding to the largest rank — and are consumed where they land
 * (MRH_UNPACK_HALO).  Terminal for fusion until mrh_drop_blocks(MRH_DROP_HALO).  *out_taken = halo blocks this rank kept. */
int mrh_comm_exchange_halo(mrh_ctx* ctx, uint64_t* out_taken);

typedef struct mrh_comm_merge_info {
  uint64_t blocks_sent;      /* blocks this rank sent to other ranks                 */
  uint64_t blocks_received;  /* blocks it received from other ranks                  */
  uint64_t bytes_sent;       /* = blocks_sent * sizeof(mrh_block_record)             */
  uint64_t blocks_kept;      /* blocks of its own sub-map that it owns (no transfer) */
} mrh_comm_merge_info;
/* Frame-sharded sub-maps -> one tile-sharded map.  Sets the context's sharding to (rank, world, chunk_log2); packs, per
 * destination, the blocks that rank owns; all-to-all with true split sizes (grouped ncclSend / ncclRecv); empties the
 * local map; folds the incoming sub-maps in rank order with combineVoxel's arithmetic (MRH_UNPACK_MERGE; variance-adaptive
 * maps: a position at two resolutions ends up coarse, see mrh_unpack_mode). */
int mrh_comm_merge_submaps(mrh_ctx* ctx, int chunk_log2, mrh_comm_merge_info* out_info);

/* Sharded extraction: every rank runs marching cubes over the blocks it owns (call mrh_comm_exchange_halo first), the
 * per-block triangle runs of all ranks travel to `root` (descriptors + counts + the 72-byte triangles, device to device),
 * which brings them into the canonical single-GPU order and post-processes them; on root the result is read with
 * mrh_extract_mesh / mrh_get_triangle_blocks / mrh_get_triangles_device, *out_triangles = total; other ranks get 0. */
int mrh_comm_gather_mesh(mrh_ctx* ctx, int root, uint64_t* out_triangles);

/* HIP-event times of the phases of the last exchange call on this context and running sums of the starve all-reduces. */
typedef struct mrh_comm_phases {
  float    pack_ms;          /* select + pack on the device                                   */
  float    counts_ms;        /* count exchange (tiny all-gather + host read-back)             */
  float    collective_ms;    /* the data-carrying collective                                  */
  float    unpack_ms;        /* consume / merge / permute on the device                       */
  uint64_t bytes_out;        /* payload bytes this rank put on the wire in that collective    */
  uint64_t bytes_in;
  float    allreduce_ms_sum; /* starve z-buffer all-reduces since attach (two per starve frame) */
  uint32_t allreduce_count;
} mrh_comm_phases;
int mrh_comm_phase_times(mrh_ctx* ctx, mrh_comm_phases* out);

#ifdef __cplusplus
}
#endif

#endif /* MRHASH_COMM_H */
