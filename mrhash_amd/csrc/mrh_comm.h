// mrh_comm.h — RCCL behind the C ABI (include/mrhash_comm.h).  Included at the end of mrh_capi.hip: it drives the
// library's own pack / unpack / drop / extraction steps and needs the context's internals.
//
// RCCL is opened at run time, from the directory of the HIP runtime this library is itself bound to (dladdr of
// hipGetDeviceCount): /opt/rocm/lib/librccl.so.1 next to /opt/rocm/lib/libamdhip64.so.7.  That keeps ONE HIP runtime per
// process on the product path — every buffer handed to RCCL was allocated by the runtime RCCL itself uses — and keeps
// the 570 MB library out of single-GPU processes.  The reference has no counterpart (single GPU).
//
// Transport choice: xGMI on an MI355X node is a full mesh of point-to-point links (7 x ~153 GB/s per GPU), so the
// variable-size exchanges (halo blocks, sub-map blocks, triangle runs) go as grouped ncclSend / ncclRecv between every
// pair — each pair's payload crosses its own direct link once, with its true size — instead of a ring all-gather padded
// to the largest rank.  Fixed-size reductions (the starve z-buffer) use ncclAllReduce.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

struct mrh_comm {
  ncclComm_t nccl = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;  // host-side helpers (barrier, scalar reductions); data collectives run on the context's stream
  char* d_small = nullptr;       // staging for those helpers
  size_t small_cap = 0;
  int attached = 0;
  std::string err;
};

namespace {

thread_local std::string g_comm_err;

struct Rccl {
  void* handle = nullptr;
  std::string path, error;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  // optional (mrh_comm_status): absent symbols leave their fields at -1
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
};

Rccl* rccl() {
  static Rccl* r = [] {
    Rccl* x = new Rccl();
    std::vector<std::string> candidates;
    if (const char* e = getenv("MRH_RCCL_PATH")) candidates.push_back(e);
    Dl_info info;
    if (dladdr((void*) &hipGetDeviceCount, &info) && info.dli_fname) {  // the HIP runtime this library is bound to
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
      candidates.push_back(dir + "/librccl.so.1");
      candidates.push_back(dir + "/librccl.so");
    }
    candidates.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& p : candidates) {
      x->handle = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (x->handle) { x->path = p; break; }
      const char* why = dlerror();
      x->error += p + ": " + (why ? why : "?") + "; ";
    }
    if (!x->handle) return x;
    bool ok = true;
    auto sym = [&](const char* name) { void* s = dlsym(x->handle, name); if (!s) { ok = false; x->error += std::string("missing ") + name + "; "; } return s; };
    x->GetUniqueId = (decltype(x->GetUniqueId)) sym("ncclGetUniqueId");
    x->CommInitRank = (decltype(x->CommInitRank)) sym("ncclCommInitRank");
    x->CommDestroy = (decltype(x->CommDestroy)) sym("ncclCommDestroy");
    x->AllReduce = (decltype(x->AllReduce)) sym("ncclAllReduce");
    x->AllGather = (decltype(x->AllGather)) sym("ncclAllGather");
    x->Send = (decltype(x->Send)) sym("ncclSend");
    x->Recv = (decltype(x->Recv)) sym("ncclRecv");
    x->GroupStart = (decltype(x->GroupStart)) sym("ncclGroupStart");
    x->GroupEnd = (decltype(x->GroupEnd)) sym("ncclGroupEnd");
    x->GetErrorString = (decltype(x->GetErrorString)) sym("ncclGetErrorString");
    if (ok) {
      x->CommCount = (decltype(x->CommCount)) dlsym(x->handle, "ncclCommCount");
      x->CommUserRank = (decltype(x->CommUserRank)) dlsym(x->handle, "ncclCommUserRank");
      x->CommCuDevice = (decltype(x->CommCuDevice)) dlsym(x->handle, "ncclCommCuDevice");
      x->CommGetAsyncError = (decltype(x->CommGetAsyncError)) dlsym(x->handle, "ncclCommGetAsyncError");
      x->GetVersion = (decltype(x->GetVersion)) dlsym(x->handle, "ncclGetVersion");
      x->CommAbort = (decltype(x->CommAbort)) dlsym(x->handle, "ncclCommAbort");
    }
    if (!ok) { dlclose(x->handle); x->handle = nullptr; }
    else if (getenv("MRH_DEBUG")) fprintf(stderr, "[mrhash_hip] RCCL: %s\n", x->path.c_str());
    return x;
  }();
  return r;
}

int comm_fail(mrh_comm* m, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (m) m->err = buf;
  else g_comm_err = buf;
  return code;
}

#define COMM_HIP(m, expr)                                                                                              \
  do {                                                                                                                 \
    hipError_t e__ = (expr);                                                                                           \
    if (e__ != hipSuccess) return comm_fail(m, MRH_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e__));       \
  } while (0)
#define COMM_NCCL(m, expr)                                                                                             \
  do {                                                                                                                 \
    ncclResult_t r__ = (expr);                                                                                         \
    if (r__ != ncclSuccess) return comm_fail(m, MRH_ERR_DEVICE, "%s failed: %s", #expr, rccl()->GetErrorString(r__)); \
  } while (0)
// the same inside functions that report through the context
#define CTX_NCCL(c, expr)                                                                                              \
  do {                                                                                                                 \
    ncclResult_t r__ = (expr);                                                                                         \
    if (r__ != ncclSuccess) return fail(c, MRH_ERR_DEVICE, "%s failed: %s", #expr, rccl()->GetErrorString(r__));      \
  } while (0)

int comm_small(mrh_comm* m, size_t bytes) {
  if (bytes <= m->small_cap) return MRH_OK;
  COMM_HIP(m, hipStreamSynchronize(m->stream));
  if (m->d_small) COMM_HIP(m, hipFree(m->d_small));
  m->d_small = nullptr; m->small_cap = 0;
  const size_t cap = std::max<size_t>(bytes, 1u << 16);
  COMM_HIP(m, hipMalloc((void**) &m->d_small, cap));
  m->small_cap = cap;
  return MRH_OK;
}

// grow-only device buffer of the context (contents up to `keep` bytes survive)
int ctx_grow(mrh_ctx* c, char*& p, size_t& cap, const size_t need, const size_t keep) {
  if (need <= cap) return MRH_OK;
  const size_t ncap = need + need / 4;
  char* grown = nullptr;
  HIP_TRY(c, hipMalloc((void**) &grown, ncap));
  if (p && keep) HIP_TRY(c, hipMemcpyAsync(grown, p, keep, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (p) HIP_TRY(c, hipFree(p));
  p = grown;
  cap = ncap;
  return MRH_OK;
}

// every rank's `n_words` 64-bit words -> all[world * n_words] on the host (device staging in the communicator, the
// collective on the CONTEXT's stream so that it is ordered with the pack that produced the numbers)
int ctx_allgather_u64(mrh_ctx* c, const uint64_t* mine, const size_t n_words, uint64_t* all) {
  mrh_comm* m = c->comm;
  const size_t bytes = n_words * sizeof(uint64_t);
  if (comm_small(m, bytes * ((size_t) m->world + 1))) return fail(c, MRH_ERR_DEVICE, "%s", m->err.c_str());
  char* d_send = m->d_small;
  char* d_recv = m->d_small + bytes;
  HIP_TRY(c, hipMemcpyAsync(d_send, mine, bytes, hipMemcpyHostToDevice, c->stream));
  CTX_NCCL(c, rccl()->AllGather(d_send, d_recv, n_words, ncclUint64, m->nccl, c->stream));
  HIP_TRY(c, hipMemcpyAsync(all, d_recv, bytes * (size_t) m->world, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return MRH_OK;
}

// Every rank contributes the status of its local steps as word 0 of a small all-gather (plus `n_words` payload words); the
// call fails on EVERY rank if it failed on any, so that no rank walks into a data collective its peers will never join (a
// pack error, an out-of-memory in ctx_grow, a broken owner partition on one rank used to leave the others hanging inside
// ncclSend / ncclRecv, which have no time limit).  all: world * (1 + n_words) words, rank r's at r * (1 + n_words).
int ctx_agree(mrh_ctx* c, const int local_rc, const char* who, const uint64_t* mine, const size_t n_words, std::vector<uint64_t>& all) {
  mrh_comm* m = c->comm;
  std::vector<uint64_t> send(1 + n_words, 0);
  send[0] = (uint64_t) (uint32_t) local_rc;
  for (size_t i = 0; i < n_words; i++) send[1 + i] = mine ? mine[i] : 0;
  all.assign((size_t) m->world * (1 + n_words), 0);
  const std::string local_err = local_rc ? c->err : std::string();
  const int rc = ctx_allgather_u64(c, send.data(), 1 + n_words, all.data());
  if (rc) return rc;  // the collective itself failed: nothing more can be agreed on
  if (local_rc) { c->err = local_err; return local_rc; }
  for (int r = 0; r < m->world; r++) {
    const int prc = (int) (uint32_t) all[(size_t) r * (1 + n_words)];
    if (prc) return fail(c, MRH_ERR_STATE, "%s: rank %d reported error %d before the exchange; nothing was moved", who, r, prc);
  }
  return MRH_OK;
}

int need_comm(mrh_ctx* c, const char* who) {
  int rc = ensure_ready(c, who);
  if (rc) return rc;
  if (!c->comm) return fail(c, MRH_ERR_STATE, "%s: no communicator attached (mrh_comm_attach)", who);
  if (c->pending) return fail(c, MRH_ERR_STATE, "%s: an exchange is pending (call mrh_integrate_resume)", who);
  return MRH_OK;
}

// MRH_COMM_SELF_LOOP=1 (tests on a one-GPU box): a rank's own part of an exchange travels through ncclSend / ncclRecv to itself
// instead of staying where it is, so that the grouped point-to-point calls, their offsets and their sizes run through RCCL with
// real payloads even in a one-rank group.  Results are identical.
bool comm_self_loop() {
  static const bool on = getenv("MRH_COMM_SELF_LOOP") != nullptr;
  return on;
}

struct PhaseClock {
  mrh_ctx* c;
  int at = 0;
  explicit PhaseClock(mrh_ctx* ctx) : c(ctx) {}
  int mark() {  // events 0 .. 4: start, packed, counted, moved, consumed
    if (!c->comm_ev[at]) HIP_TRY(c, hipEventCreate(&c->comm_ev[at]));
    HIP_TRY(c, hipEventRecord(c->comm_ev[at], c->stream));
    at++;
    return MRH_OK;
  }
  int finish(const uint64_t bytes_out, const uint64_t bytes_in) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float ms[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i + 1 < at && i < 4; i++) HIP_TRY(c, hipEventElapsedTime(&ms[i], c->comm_ev[i], c->comm_ev[i + 1]));
    c->comm_phases.pack_ms = ms[0]; c->comm_phases.counts_ms = ms[1]; c->comm_phases.collective_ms = ms[2]; c->comm_phases.unpack_ms = ms[3];
    c->comm_phases.bytes_out = bytes_out; c->comm_phases.bytes_in = bytes_in;
    return MRH_OK;
  }
};

}  // namespace

static bool comm_matches_sharding(const mrh_ctx* c, int* comm_rank, int* comm_world) {
  *comm_rank = c->comm->rank; *comm_world = c->comm->world;
  return c->p.shard_count == c->comm->world && c->p.shard_rank == c->comm->rank;
}

static void comm_release(mrh_ctx* c) {  // free_all: the context goes away
  if (c->comm) { c->comm->attached--; c->comm = nullptr; }
}

// the two MIN all-reduces of a starve frame on a tile-sharded context with a communicator (called from starve_and_tail)
static int comm_allreduce_zbuf(mrh_ctx* c, u64* buf, const size_t n) {
  EvPair ev;
  if (!c->comm_ev_pool.empty()) { ev = c->comm_ev_pool.back(); c->comm_ev_pool.pop_back(); }
  else { HIP_TRY(c, hipEventCreate(&ev.a)); HIP_TRY(c, hipEventCreate(&ev.b)); }
  HIP_TRY(c, hipEventRecord(ev.a, c->stream));
  // every key is < 2^63 ("empty" = INT64_MAX): the unsigned order is the signed one
  CTX_NCCL(c, rccl()->AllReduce(buf, buf, n, ncclInt64, ncclMin, c->comm->nccl, c->stream));
  HIP_TRY(c, hipEventRecord(ev.b, c->stream));
  c->comm_ev_pending.push_back(ev);
  if (c->comm_ev_pending.size() >= 1024) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& e : c->comm_ev_pending) {
      float ms = 0.f;
      HIP_TRY(c, hipEventElapsedTime(&ms, e.a, e.b));
      c->comm_phases.allreduce_ms_sum += ms; c->comm_phases.allreduce_count++;
      c->comm_ev_pool.push_back(e);
    }
    c->comm_ev_pending.clear();
  }
  return MRH_OK;
}

extern "C" {

const char* mrh_comm_last_error(const mrh_comm* m) { return m ? m->err.c_str() : g_comm_err.c_str(); }

int mrh_comm_unique_id(uint8_t out_id[MRH_COMM_ID_BYTES]) {
  if (!out_id) return comm_fail(nullptr, MRH_ERR_INVALID_ARG, "mrh_comm_unique_id: null argument");
  Rccl* r = rccl();
  if (!r->handle) return comm_fail(nullptr, MRH_ERR_UNSUPPORTED, "RCCL not available: %s", r->error.c_str());
  static_assert(sizeof(ncclUniqueId) == MRH_COMM_ID_BYTES, "id size");
  ncclUniqueId id;
  COMM_NCCL(nullptr, r->GetUniqueId(&id));
  memcpy(out_id, &id, sizeof id);
  return MRH_OK;
}

int mrh_comm_create(const uint8_t id_bytes[MRH_COMM_ID_BYTES], int rank, int world, int device_id, mrh_comm** out) {
  if (!id_bytes || !out || world < 1 || rank < 0 || rank >= world) return comm_fail(nullptr, MRH_ERR_INVALID_ARG, "mrh_comm_create: bad argument (rank %d of %d)", rank, world);
  Rccl* r = rccl();
  if (!r->handle) return comm_fail(nullptr, MRH_ERR_UNSUPPORTED, "RCCL not available: %s", r->error.c_str());
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return comm_fail(nullptr, MRH_ERR_NO_DEVICE, "mrh_comm_create: no HIP device visible");
  if (device_id < 0 || device_id >= ndev) return comm_fail(nullptr, MRH_ERR_INVALID_ARG, "mrh_comm_create: device_id %d out of range (%d devices)", device_id, ndev);
  COMM_HIP(nullptr, hipSetDevice(device_id));
  mrh_comm* m = new mrh_comm();
  m->rank = rank; m->world = world; m->device = device_id;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof id);
  // ncclCommInitRank has no time limit of its own: a peer that never joins (a rank that died, a stale id, a fabric problem)
  // would keep this call — and the run — for ever.  It runs on a thread of its own; past MRH_COMM_INIT_TIMEOUT_S (default 180)
  // the call reports the failure and leaves that thread behind (there is no communicator to abort before the call returns; the
  // thread ends with the process), and the caller falls back to whatever it has without RCCL (bench.py: the host group).
  struct Init {
    std::mutex m; std::condition_variable cv;
    bool done = false, abandoned = false;
    ncclResult_t rc = ncclSuccess; ncclComm_t comm = nullptr;
  };
  auto st = std::make_shared<Init>();
  double limit_s = 180.0;
  if (const char* e = getenv("MRH_COMM_INIT_TIMEOUT_S")) { const double v = atof(e); if (v > 0) limit_s = v; }
  std::thread([st, r, world, id, rank, device_id] {
    ncclComm_t comm = nullptr;
    ncclResult_t rc = hipSetDevice(device_id) == hipSuccess ? r->CommInitRank(&comm, world, id, rank) : ncclUnhandledCudaError;
    std::unique_lock<std::mutex> lk(st->m);
    st->rc = rc; st->comm = comm; st->done = true;
    if (st->abandoned && rc == ncclSuccess && comm) {  // nobody is waiting any more: the communicator goes away again
      lk.unlock();
      if (r->CommAbort) (void) r->CommAbort(comm); else (void) r->CommDestroy(comm);
      return;
    }
    st->cv.notify_all();
  }).detach();
  ncclResult_t rc;
  {
    std::unique_lock<std::mutex> lk(st->m);
    if (!st->cv.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return st->done; })) {
      st->abandoned = true;
      comm_fail(nullptr, MRH_ERR_DEVICE, "ncclCommInitRank(rank %d of %d, device %d) did not return within %.0f s (MRH_COMM_INIT_TIMEOUT_S): a peer never joined",
                rank, world, device_id, limit_s);
      delete m;
      return MRH_ERR_DEVICE;
    }
    rc = st->rc; m->nccl = st->comm;
  }
  if (rc != ncclSuccess) {
    comm_fail(nullptr, MRH_ERR_DEVICE, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device_id, r->GetErrorString(rc));
    delete m;
    return MRH_ERR_DEVICE;
  }
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
    comm_fail(nullptr, MRH_ERR_DEVICE, "mrh_comm_create: hipStreamCreate failed");
    r->CommDestroy(m->nccl);
    delete m;
    return MRH_ERR_DEVICE;
  }
  *out = m;
  return MRH_OK;
}

int mrh_comm_destroy(mrh_comm* m) {
  if (!m) return MRH_OK;
  if (m->attached) return comm_fail(m, MRH_ERR_STATE, "mrh_comm_destroy: %d context(s) still attached", m->attached);
  (void) hipSetDevice(m->device);
  if (m->stream) { (void) hipStreamSynchronize(m->stream); }
  if (m->nccl) (void) rccl()->CommDestroy(m->nccl);
  if (m->d_small) (void) hipFree(m->d_small);
  if (m->stream) (void) hipStreamDestroy(m->stream);
  delete m;
  return MRH_OK;
}

// What RCCL itself says about the communicator (bench.py puts it into the N > 1 line: the driver can see that RCCL saw N ranks).
int mrh_comm_status(mrh_comm* m, mrh_comm_status_info* out) {
  if (!m || !out) return MRH_ERR_INVALID_ARG;
  Rccl* r = rccl();
  memset(out, 0, sizeof *out);
  out->rccl_ranks = out->rccl_rank = out->rccl_device = out->rccl_version = -1;
  out->async_error = -1;
  snprintf(out->async_error_string, sizeof out->async_error_string, "%s", "ncclCommGetAsyncError not available");
  if (r->CommCount && r->CommCount(m->nccl, &out->rccl_ranks) != ncclSuccess) out->rccl_ranks = -1;
  if (r->CommUserRank && r->CommUserRank(m->nccl, &out->rccl_rank) != ncclSuccess) out->rccl_rank = -1;
  if (r->CommCuDevice && r->CommCuDevice(m->nccl, &out->rccl_device) != ncclSuccess) out->rccl_device = -1;
  if (r->GetVersion && r->GetVersion(&out->rccl_version) != ncclSuccess) out->rccl_version = -1;
  if (r->CommGetAsyncError) {
    ncclResult_t ae = ncclSuccess;
    const ncclResult_t q = r->CommGetAsyncError(m->nccl, &ae);
    out->async_error = q == ncclSuccess ? (int) ae : (int) q;
    snprintf(out->async_error_string, sizeof out->async_error_string, "%s", r->GetErrorString(q == ncclSuccess ? ae : q));
  }
  snprintf(out->library_path, sizeof out->library_path, "%s", r->path.c_str());
  return MRH_OK;
}

int mrh_comm_size(const mrh_comm* m, int* out_rank, int* out_world) {
  if (!m) return MRH_ERR_INVALID_ARG;
  if (out_rank) *out_rank = m->rank;
  if (out_world) *out_world = m->world;
  return MRH_OK;
}

int mrh_comm_allreduce_f64(mrh_comm* m, double* inout, uint64_t n, int op) {
  if (!m || (n && !inout)) return MRH_ERR_INVALID_ARG;
  if (op != MRH_COMM_SUM && op != MRH_COMM_MAX && op != MRH_COMM_MIN) return comm_fail(m, MRH_ERR_INVALID_ARG, "mrh_comm_allreduce_f64: bad op %d", op);
  if (n == 0) return MRH_OK;
  COMM_HIP(m, hipSetDevice(m->device));
  int rc = comm_small(m, n * sizeof(double));
  if (rc) return rc;
  COMM_HIP(m, hipMemcpyAsync(m->d_small, inout, n * sizeof(double), hipMemcpyHostToDevice, m->stream));
  const ncclRedOp_t o = op == MRH_COMM_SUM ? ncclSum : op == MRH_COMM_MAX ? ncclMax : ncclMin;
  COMM_NCCL(m, rccl()->AllReduce(m->d_small, m->d_small, n, ncclFloat64, o, m->nccl, m->stream));
  COMM_HIP(m, hipMemcpyAsync(inout, m->d_small, n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
  COMM_HIP(m, hipStreamSynchronize(m->stream));
  return MRH_OK;
}

int mrh_comm_barrier(mrh_comm* m) {
  double one = 1.0;
  return mrh_comm_allreduce_f64(m, &one, 1, MRH_COMM_SUM);
}

int mrh_comm_allgather_bytes(mrh_comm* m, const void* send, uint64_t bytes, void* recv) {
  if (!m || (bytes && (!send || !recv))) return MRH_ERR_INVALID_ARG;
  if (bytes == 0) return MRH_OK;
  COMM_HIP(m, hipSetDevice(m->device));
  int rc = comm_small(m, bytes * ((size_t) m->world + 1));
  if (rc) return rc;
  COMM_HIP(m, hipMemcpyAsync(m->d_small, send, bytes, hipMemcpyHostToDevice, m->stream));
  COMM_NCCL(m, rccl()->AllGather(m->d_small, m->d_small + bytes, bytes, ncclUint8, m->nccl, m->stream));
  COMM_HIP(m, hipMemcpyAsync(recv, m->d_small + bytes, bytes * (size_t) m->world, hipMemcpyDeviceToHost, m->stream));
  COMM_HIP(m, hipStreamSynchronize(m->stream));
  return MRH_OK;
}

int mrh_comm_attach(mrh_ctx* c, mrh_comm* m) {
  if (!c) return MRH_ERR_INVALID_ARG;
  if (c->deferred.on) { (void) hipSetDevice(c->device); const int drc = flush_deferred(c); if (drc < 0) return drc; }
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_comm_attach: an exchange is pending (call mrh_integrate_resume)");
  if (m && m->device != c->device) return fail(c, MRH_ERR_INVALID_ARG, "mrh_comm_attach: communicator on device %d, context on device %d", m->device, c->device);
  if (c->comm == m) return MRH_OK;
  if (c->comm) {
    (void) hipSetDevice(c->device);
    (void) hipStreamSynchronize(c->stream);  // nothing of this context may still be inside a collective
    c->comm->attached--;
  }
  c->comm = m;
  if (m) m->attached++;
  return MRH_OK;
}

int mrh_comm_phase_times(mrh_ctx* c, mrh_comm_phases* out) {
  int rc = ensure_ready(c, "mrh_comm_phase_times");
  if (rc) return rc;
  if (!out) return MRH_ERR_INVALID_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (auto& e : c->comm_ev_pending) {
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, e.a, e.b));
    c->comm_phases.allreduce_ms_sum += ms; c->comm_phases.allreduce_count++;
    c->comm_ev_pool.push_back(e);
  }
  c->comm_ev_pending.clear();
  *out = c->comm_phases;
  return MRH_OK;
}

int mrh_comm_exchange_halo(mrh_ctx* c, uint64_t* out_taken) {
  int rc = need_comm(c, "mrh_comm_exchange_halo");
  if (rc) return rc;
  if (out_taken) *out_taken = 0;
  mrh_comm* m = c->comm;
  const int world = m->world, rank = m->rank;
  if (c->p.shard_count != world || c->p.shard_rank != rank)
    return fail(c, MRH_ERR_STATE, "mrh_comm_exchange_halo: the context is shard %d of %d, the communicator rank %d of %d", c->p.shard_rank, c->p.shard_count, rank, world);
  PhaseClock clk(c);
  const mrh_block_record* mine = nullptr;
  uint64_t n = 0;
  const size_t rec = sizeof(mrh_block_record);
  // local steps first, their status travels with the counts: either every rank enters the data exchange or none does
  rc = clk.mark();
  if (!rc) rc = mrh_pack_blocks(c, MRH_PACK_HALO, 0, &mine, &n, nullptr);
  if (!rc) rc = clk.mark();
  std::vector<uint64_t> agreed;
  rc = ctx_agree(c, rc, "mrh_comm_exchange_halo", &n, 1, agreed);
  if (rc) return rc;
  std::vector<uint64_t> counts((size_t) world);
  for (int r = 0; r < world; r++) counts[r] = agreed[2 * (size_t) r + 1];
  if ((rc = clk.mark())) return rc;
  uint64_t total_in = 0;
  std::vector<uint64_t> off((size_t) world, 0);
  const bool self = comm_self_loop();
  for (int r = 0; r < world; r++) { off[r] = total_in; if (r != rank || self) total_in += counts[r]; }
  rc = ctx_grow(c, c->d_xrecv, c->xrecv_cap, std::max<size_t>(total_in * rec, 256), 0);
  rc = ctx_agree(c, rc, "mrh_comm_exchange_halo", nullptr, 0, agreed);  // the receive buffer exists on every rank
  if (rc) return rc;
  if (world > 1 || self) {
    CTX_NCCL(c, rccl()->GroupStart());
    for (int r = 0; r < world; r++) {
      if (r == rank && !self) continue;
      if (n) CTX_NCCL(c, rccl()->Send(mine, (size_t) n * rec, ncclUint8, r, m->nccl, c->stream));
      if (counts[r]) CTX_NCCL(c, rccl()->Recv(c->d_xrecv + off[r] * rec, (size_t) counts[r] * rec, ncclUint8, r, m->nccl, c->stream));
    }
    CTX_NCCL(c, rccl()->GroupEnd());
  }
  if ((rc = clk.mark())) return rc;
  uint64_t taken = 0;
  if (total_in) {  // one call over every peer's records: a position has exactly one owner, so the keys are unique
    rc = mrh_unpack_blocks(c, MRH_UNPACK_HALO, (const mrh_block_record*) c->d_xrecv, total_in, 1, &taken);
    if (rc) return rc;
  }
  if ((rc = clk.mark())) return rc;
  if (out_taken) *out_taken = taken;
  return clk.finish((uint64_t) (world - 1) * n * rec, total_in * rec);
}

int mrh_comm_merge_submaps(mrh_ctx* c, int chunk_log2, mrh_comm_merge_info* out) {
  int rc = need_comm(c, "mrh_comm_merge_submaps");
  if (rc) return rc;
  if (out) memset(out, 0, sizeof *out);
  if (c->halo_upper) return fail(c, MRH_ERR_STATE, "mrh_comm_merge_submaps: halo blocks are present (mrh_drop_blocks(MRH_DROP_HALO) first)");
  mrh_comm* m = c->comm;
  const int world = m->world, rank = m->rank;
  const int old_rank = c->p.shard_rank, old_count = c->p.shard_count, old_log2 = c->p.shard_chunk_log2;
  const size_t rec = sizeof(mrh_block_record);
  PhaseClock clk(c);
  // one send buffer, the parts in destination order (every live block has exactly one owner: n_all records in total).
  // Local steps first; their status travels with the count matrix, so a rank that failed keeps the others out of the exchange.
  std::vector<uint64_t> out_counts((size_t) world, 0), out_off((size_t) world, 0);
  auto pack_all = [&]() -> int {
    int r2 = mrh_set_sharding(c, rank, world, chunk_log2);
    if (r2) return r2;
    if ((r2 = clk.mark())) return r2;
    int n_all = 0;
    if ((r2 = select_blocks(c, kSelAll, 0, &n_all))) return r2;
    if ((r2 = ctx_grow(c, c->d_xsend, c->xsend_cap, std::max<size_t>((size_t) n_all * rec, 256), 0))) return r2;
    uint64_t packed = 0;
    for (int dest = 0; dest < world; dest++) {
      int n = 0;
      if ((r2 = select_blocks(c, kSelOwner, dest, &n))) return r2;
      out_off[dest] = packed;
      out_counts[dest] = (uint64_t) n;
      if (packed + (uint64_t) n > (uint64_t) n_all) return fail(c, MRH_ERR_STATE, "mrh_comm_merge_submaps: the owner partition does not add up");
      if (n) k_pack_records<<<n < 4096 ? n : 4096, 512, 0, c->stream>>>(c->tab, 0, n, c->d_xsend + packed * rec);
      packed += (uint64_t) n;
    }
    HIP_TRY(c, hipGetLastError());
    return clk.mark();
  };
  auto restore_sharding = [&](const int code) {  // nothing has been moved or dropped yet: the context goes back to what it was
    const std::string keep = c->err;
    (void) mrh_set_sharding(c, old_rank, old_count, old_log2);
    c->err = keep;
    return code;
  };
  rc = pack_all();
  std::vector<uint64_t> agreed;
  rc = ctx_agree(c, rc, "mrh_comm_merge_submaps", out_counts.data(), (size_t) world, agreed);
  if (rc) return restore_sharding(rc);
  std::vector<uint64_t> in_counts((size_t) world), in_off((size_t) world, 0);
  uint64_t total_in = 0;
  const bool self = comm_self_loop();
  for (int src = 0; src < world; src++) {
    in_counts[src] = agreed[(size_t) src * (world + 1) + 1 + rank];  // matrix[src][dest = this rank]
    in_off[src] = total_in;
    if (src != rank || self) total_in += in_counts[src];
  }
  rc = ctx_grow(c, c->d_xrecv, c->xrecv_cap, std::max<size_t>(total_in * rec, 256), 0);
  rc = ctx_agree(c, rc, "mrh_comm_merge_submaps", nullptr, 0, agreed);
  if (rc) return restore_sharding(rc);
  if ((rc = clk.mark())) return rc;
  uint64_t sent = 0;
  if (world > 1 || self) {
    CTX_NCCL(c, rccl()->GroupStart());
    for (int r = 0; r < world; r++) {
      if (r == rank && !self) continue;
      if (out_counts[r]) CTX_NCCL(c, rccl()->Send(c->d_xsend + out_off[r] * rec, (size_t) out_counts[r] * rec, ncclUint8, r, m->nccl, c->stream));
      if (in_counts[r]) CTX_NCCL(c, rccl()->Recv(c->d_xrecv + in_off[r] * rec, (size_t) in_counts[r] * rec, ncclUint8, r, m->nccl, c->stream));
      if (r != rank) sent += out_counts[r];
    }
    CTX_NCCL(c, rccl()->GroupEnd());
  }
  if ((rc = clk.mark())) return rc;
  // the owned blocks come back through the fold, at this rank's position in the order: deterministic for a given world size
  rc = mrh_drop_blocks(c, MRH_DROP_ALL, nullptr);
  if (rc) return rc;
  for (int src = 0; src < world; src++) {
    const char* seg = (src == rank && !self) ? c->d_xsend + out_off[rank] * rec : c->d_xrecv + in_off[src] * rec;
    if (in_counts[src] == 0) continue;
    rc = mrh_unpack_blocks(c, MRH_UNPACK_MERGE, (const mrh_block_record*) seg, in_counts[src], 1, nullptr);
    if (rc) return rc;
  }
  if ((rc = clk.mark())) return rc;
  if (out) {
    out->blocks_sent = sent; out->blocks_received = total_in - (self ? in_counts[rank] : 0); out->bytes_sent = sent * rec; out->blocks_kept = out_counts[rank];
  }
  return clk.finish(sent * rec, (total_in - (self ? in_counts[rank] : 0)) * rec);
}

int mrh_comm_gather_mesh(mrh_ctx* c, int root, uint64_t* out_triangles) {
  int rc = need_comm(c, "mrh_comm_gather_mesh");
  if (rc) return rc;
  if (out_triangles) *out_triangles = 0;
  mrh_comm* m = c->comm;
  const int world = m->world, rank = m->rank;
  if (root < 0 || root >= world) return fail(c, MRH_ERR_INVALID_ARG, "mrh_comm_gather_mesh: root %d of %d ranks", root, world);
  PhaseClock clk(c);
  uint64_t nt = 0;
  const mrh_block_desc* descs = nullptr;
  const uint32_t* cnts = nullptr;
  uint64_t nblk = 0;
  rc = clk.mark();
  if (!rc) rc = mrh_extract_triangles(c, nullptr, &nt);  // the soup stays in c->d_soup
  if (!rc) rc = mrh_get_triangle_blocks(c, &descs, &cnts, &nblk);
  // per-block metadata of the non-empty blocks: 16-byte descriptor + 4-byte count
  std::vector<mrh_block_desc> my_d;
  std::vector<uint32_t> my_c;
  if (!rc)
    for (uint64_t i = 0; i < nblk; i++)
      if (cnts[i]) { my_d.push_back(descs[i]); my_c.push_back(cnts[i]); }
  const uint64_t nb = my_d.size();
  if (!rc) rc = clk.mark();
  const uint64_t mine[2] = {nb, rc ? 0 : nt};
  std::vector<uint64_t> agreed;
  rc = ctx_agree(c, rc, "mrh_comm_gather_mesh", mine, 2, agreed);
  if (rc) return rc;
  std::vector<uint64_t> sizes((size_t) world * 2);
  for (int r = 0; r < world; r++) { sizes[2 * r] = agreed[3 * (size_t) r + 1]; sizes[2 * r + 1] = agreed[3 * (size_t) r + 2]; }
  if ((rc = clk.mark())) return rc;
  uint64_t tot_b = 0, tot_t = 0;
  std::vector<uint64_t> boff((size_t) world), toff((size_t) world);
  for (int r = 0; r < world; r++) { boff[r] = tot_b; toff[r] = tot_t; tot_b += sizes[2 * r]; tot_t += sizes[2 * r + 1]; }
  // metadata travels through device staging (20 bytes a block), the triangles from soup to soup.  The triangle area starts on a
  // 256-byte boundary: the run merge reads it with 16-byte loads (k_permute_runs), and 20 * tot_b is only 4-byte aligned.
  const size_t meta_mine = (size_t) nb * 20, meta_all = (size_t) tot_b * 20, meta_span = (meta_all + 255) & ~(size_t) 255;
  rc = ctx_grow(c, c->d_xsend, c->xsend_cap, std::max<size_t>(meta_mine, 256), 0);
  if (!rc && nb) {
    hipError_t e = hipMemcpyAsync(c->d_xsend, my_d.data(), nb * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_xsend + nb * 16, my_c.data(), nb * 4, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) rc = fail(c, MRH_ERR_DEVICE, "mrh_comm_gather_mesh: staging the block metadata failed: %s", hipGetErrorString(e));
  }
  const size_t tri = sizeof(mrh_triangle);
  if (!rc && rank == root) rc = ctx_grow(c, c->d_xrecv, c->xrecv_cap, std::max<size_t>(meta_span + tot_t * tri, 256), 0);
  rc = ctx_agree(c, rc, "mrh_comm_gather_mesh", nullptr, 0, agreed);  // buffers exist everywhere before anybody sends
  if (rc) return rc;
  char* d_meta = c->d_xrecv;             // root: [rank r's descs | counts] at boff[r] * 20
  char* d_tris = c->d_xrecv + meta_span;  // root: rank r's triangles at toff[r]
  const mrh_triangle* soup = c->soup_n ? c->d_soup : nullptr;
  const bool self = comm_self_loop();
  if (world > 1 || self) {
    CTX_NCCL(c, rccl()->GroupStart());
    if (rank == root && self) {
      if (nb) CTX_NCCL(c, rccl()->Send(c->d_xsend, meta_mine, ncclUint8, root, m->nccl, c->stream));
      if (nt) CTX_NCCL(c, rccl()->Send(soup, (size_t) nt * tri, ncclUint8, root, m->nccl, c->stream));
    }
    if (rank == root) {
      for (int r = 0; r < world; r++) {
        if (r == root && !self) continue;
        if (sizes[2 * r]) CTX_NCCL(c, rccl()->Recv(d_meta + boff[r] * 20, (size_t) sizes[2 * r] * 20, ncclUint8, r, m->nccl, c->stream));
        if (sizes[2 * r + 1]) CTX_NCCL(c, rccl()->Recv(d_tris + toff[r] * tri, (size_t) sizes[2 * r + 1] * tri, ncclUint8, r, m->nccl, c->stream));
      }
    } else {
      if (nb) CTX_NCCL(c, rccl()->Send(c->d_xsend, meta_mine, ncclUint8, root, m->nccl, c->stream));
      if (nt) CTX_NCCL(c, rccl()->Send(soup, (size_t) nt * tri, ncclUint8, root, m->nccl, c->stream));
    }
    CTX_NCCL(c, rccl()->GroupEnd());
  }
  uint64_t bytes_in = 0;
  if (rank == root) {
    if (nb && !self) HIP_TRY(c, hipMemcpyAsync(d_meta + boff[root] * 20, c->d_xsend, meta_mine, hipMemcpyDeviceToDevice, c->stream));
    if (nt && !self) HIP_TRY(c, hipMemcpyAsync(d_tris + toff[root] * tri, soup, (size_t) nt * tri, hipMemcpyDeviceToDevice, c->stream));
    bytes_in = (tot_b - nb) * 20 + (tot_t - nt) * tri;
  }
  if ((rc = clk.mark())) return rc;
  if (rank == root) {
    std::vector<char> h_meta(meta_all ? meta_all : 1);
    if (meta_all) HIP_TRY(c, hipMemcpyAsync(h_meta.data(), d_meta, meta_all, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<mrh_block_desc> all_d((size_t) tot_b);
    std::vector<uint32_t> all_c((size_t) tot_b);
    for (int r = 0; r < world; r++) {
      const uint64_t b = sizes[2 * r];
      if (!b) continue;
      memcpy(&all_d[boff[r]], h_meta.data() + boff[r] * 20, b * 16);
      memcpy(&all_c[boff[r]], h_meta.data() + boff[r] * 20 + b * 16, b * 4);
    }
    // d_tris is not c->d_soup: the run merge writes the soup buffer while it reads this one
    rc = mrh_process_triangle_runs(c, all_d.data(), all_c.data(), tot_b, (const mrh_triangle*) d_tris, tot_t, 1);
    if (rc) return rc;
    if (out_triangles) *out_triangles = tot_t;
  }
  if ((rc = clk.mark())) return rc;
  return clk.finish(rank == root ? 0 : meta_mine + nt * tri, bytes_in);
}

}  // extern "C"
