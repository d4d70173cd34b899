// mrh_capi.hip — implementation of the C ABI (include/mrhash_hip.h) on top of the gfx950 kernels.
//
// Host side of the thin HIP layer: owns the device buffers, enqueues the per-frame kernel chain on one
// stream with no host round trip, and only synchronises in the calls that hand data back.
// There is NO CPU fallback in this file: without a HIP device mrh_create fails with MRH_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <sys/mman.h>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define MRH_CPU_RELAX() _mm_pause()
#else
#define MRH_CPU_RELAX() ((void) 0)  // host code as the device pass sees it
#endif

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mrhash_hip.h"
#include "../../include/mrhash_comm.h"
#include "mrh_kernels.h"
#include "mrh_mc.h"
#include "mrh_fast.h"
#include "mrh_pipe.h"
#include "mrh_fast2.h"
#include "mrh_mesh.h"
#include "mrh_lidar.h"
#include "mrh_scan.h"
#include "mrh_sort.h"
#include "mrh_splat.h"

using namespace mrh;

constexpr int kPipeRing = 6;  // = mrh::kListSets: frames in flight + 2

namespace {

thread_local std::string g_create_err;

struct EvPair {
  hipEvent_t a, b;
};

struct UpSlot {
  void* h = nullptr;            // pinned staging
  void* d = nullptr;            // device image
  size_t cap = 0;
  hipEvent_t copied = nullptr;  // H2D out of `h` done (copy stream)
  bool copied_rec = false;
  uint64_t last_seq = 0;        // newest frame_done mark of a frame that read `d` (0: none)
};
struct UpRing {
  UpSlot s[3];
  int cur = -1;                 // slot holding the current image; -1: none, or a caller's device pointer
  hipStream_t stream = nullptr; // this ring's copy stream
  hipEvent_t last_copy = nullptr;  // newest copy event of this ring
  bool waited[2] = {false, false}; // ... has been waited for by {main, front} stream
};

// Host result buffer of the blocking calls (triangle soup, V / F / C): grow-only, never zero-filled, PINNED.
// A device-to-host copy into pageable memory is pinned and unpinned by the runtime around every call, page by page: with
// transparent huge pages behind the buffer that is cheap (tools/micro/d2h_paths.hip: 128 MB in 2.4 ms), with 4 KiB pages it
// doubles the copy (30 MB of V / F / C: 0.57 -> 1.5 ms) — and which of the two a malloc'ed buffer gets depends on what the
// process freed before (glibc raises its mmap threshold after the first large free; the next buffer then comes from the heap,
// where MADV_HUGEPAGE does nothing for pages that already exist).  Pinned once, the copy runs at link speed every time, and a
// kernel can write the buffer (k_copy_out).  The pinned memory is an anonymous 2 MiB-aligned mapping advised to huge pages,
// touched, and registered (hipHostRegister): 1.5 ms for 36 MB where hipHostMalloc takes 5-9 ms (tools/micro/pinned_alloc_cost.hip)
// — what a context's FIRST extraction pays.  If the registration is refused the mapping stays as a pageable buffer (dev == nullptr:
// copies go through hipMemcpyAsync).
template <typename T>
struct HostVec {
  T* p = nullptr;    // host pointer
  T* dev = nullptr;  // the same memory as the device sees it (nullptr: not registered)
  bool pin = true;   // false: a plain huge-page mapping the device never touches (V / C doubles, filled by the host's widening)
  size_t n = 0, cap = 0;
  size_t span = 0, head = 0;  // the mapping: its size and the bytes between its base and p
  HostVec() = default;
  HostVec(const HostVec&) = delete;
  HostVec& operator=(const HostVec&) = delete;
  ~HostVec() { release(); }
  void release() {
    if (!p) return;
    if (dev) (void) hipHostUnregister((void*) p);
    (void) munmap((void*) ((char*) p - head), span);
    p = nullptr; dev = nullptr; cap = 0; span = 0; head = 0;
  }
  T* data() { return p; }
  const T* data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  void clear() { n = 0; }
  const T& operator[](size_t i) const { return p[i]; }
  bool pin_pending = false;  // mapped and faulted in by reserve_unpinned, not registered yet: the next resize_discard registers it
  // the mapping alone: mmap + huge-page advice + first touch.  No HIP call — safe on a helper thread next to a frame loop (a
  // hipHostRegister on another thread holds the runtime's lock for its whole 1-2 ms: round 6 measured the frame loop at a quarter
  // of its rate with the registration on a helper thread)
  void map_(const size_t count, const bool touch) {
    release();
    const size_t want = count + count / 8;  // head room: a map that grows a little keeps its buffer
    const size_t bytes = ((want * sizeof(T) + (2u << 20) - 1) >> 21) << 21;
    const size_t sp = bytes + (2u << 20);
    void* m = mmap(nullptr, sp, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) throw std::bad_alloc();
    // the whole span is kept (the unaligned head stays untouched, i.e. unbacked): one munmap releases it
    char* aligned = (char*) (((uintptr_t) m + (2u << 20) - 1) & ~(uintptr_t) ((2u << 20) - 1));
    (void) madvise(aligned, bytes, MADV_HUGEPAGE);
    // fault the pages in (as huge pages) before they are pinned; a buffer the device never sees is faulted in by whoever
    // writes it first — the widening threads, side by side (47 MB of V / C at the driver's workload: zeroing them here, on one
    // thread, was most of a context's first extraction) — unless the prewarm asks for it
    if (touch) for (size_t o = 0; o < bytes; o += 4096) ((volatile char*) aligned)[o] = 0;
    span = sp;
    head = (size_t) (aligned - (char*) m);
    p = (T*) aligned;
    dev = nullptr;
    cap = bytes / sizeof(T);
  }
  void register_() {
    pin_pending = false;
    void* d = nullptr;
    const size_t bytes = cap * sizeof(T);
    if (hipHostRegister((void*) p, bytes, hipHostRegisterDefault) == hipSuccess && hipHostGetDevicePointer(&d, (void*) p, 0) == hipSuccess && d) {
      dev = (T*) d;
    } else {
      (void) hipGetLastError();
      (void) hipHostUnregister((void*) p);
      (void) hipGetLastError();
      dev = nullptr;
    }
  }
  // capacity for `count` elements, faulted in, registration left to the first resize_discard (helper thread: no HIP call)
  void reserve_unpinned(const size_t count) {
    if (count <= cap) return;
    map_(count, true);
    pin_pending = pin;
    n = 0;
  }
  // contents are NOT preserved when the buffer grows
  void resize_discard(size_t count) {
    if (count > cap) {
      map_(count, pin);
      pin_pending = false;
      if (pin) register_();
    } else if (pin_pending) {
      register_();
    }
    n = count;
  }
  void assign(const T* a, const T* b) {
    resize_discard((size_t) (b - a));
    if (n) memcpy(p, a, n * sizeof(T));
  }
};

}  // namespace

struct mrh_ctx {
  mrh_params p;
  int device = 0;
  hipStream_t stream = nullptr;
  Cam cam;
  Map map;
  Tab tab;
  bool has_camera = false;
  bool spherical = false;
  // images.  Host uploads (mrh_upload_depth / _rgb) go through a ring of three slots per image kind — pinned staging +
  // device buffer — on a second stream, so the copy of frame N+1 overlaps the kernels of frame N; the frame's kernels
  // wait for the newest copy event, a slot is rewritten only after the last frame that read it (frame_done event).
  UpRing up_depth, up_rgb;
  // A frame of host images is LAUNCHED one mrh_integrate late (round 5): by then its two transfers have completed and the frame's
  // kernels need no cross-stream wait — a wait that is enqueued while its event is still pending costs the waiting stream ~6 us
  // of idle time, and a host-fed frame had two of them in front of its 41 us of kernels.  mrh_integrate checks what it can,
  // keeps {pose, image pointers, ring state} and returns; the next mrh_integrate — or whichever other entry point needs the map
  // (ensure_ready) — runs the frame first, with those inputs swapped in.  MRH_DEFER_UPLOADS=0 launches at once.
  struct DeferredFrame {
    bool on = false;
    int n_inval = 0;
    Cam cam;
    const float* d_depth = nullptr; const uint8_t* d_rgb = nullptr;
    int depth_rows = 0, depth_cols = 0, rgb_rows = 0, rgb_cols = 0;
    struct Ring { int cur; hipEvent_t last_copy; bool waited[2]; } ring[2];
  } deferred;
  int defer_uploads = 1;
  bool copy_ready = false;             // copy stream and frame marks exist
  hipEvent_t frame_done[8] = {};       // recorded behind the kernels that READ a frame's ring slots (front stream for a pipelined frame): slot reuse
  hipEvent_t peek_done[8] = {};        // recorded on the main stream behind the k_report of a mark: what the non-blocking peeks query
  uint64_t frame_seq = 1;
  // pool level for the host without a read-back stall (mrh_peek_free_blocks): a 2-int D2H per frame into pinned memory
  int* h_peek = nullptr;               // [8][8] pinned: ctr[0 .. 4] = free-list levels ... error flags per report
  // grow-only device scratch of the extraction (0: block list / counts / per-voxel counts, 1: mesh post-process, 2: V / C / F):
  // a mesh of a million triangles needs ~400 MB of temporaries, and hipMalloc + hipFree of those cost more than the kernels
  void* arena[3] = {nullptr, nullptr, nullptr};
  size_t arena_cap[3] = {0, 0, 0};
  uint64_t peek_seq[8] = {};
  bool peek_enabled = false;
  const float* d_depth = nullptr;
  const uint8_t* d_rgb = nullptr;
  int depth_rows = 0, depth_cols = 0, rgb_rows = 0, rgb_cols = 0;
  // scratch
  u32* d_decision = nullptr;
  u64* d_zbuf = nullptr;  // 2 * npix
  size_t zbuf_n = 0;
  // starve frames on the two-launch path (mrh_fast2.h: k_starve_z / k_starve_tail): two PAIRS of z-buffers, the tail launch of
  // one starve frame puts the other pair back to "empty" for the next
  u64* d_zfused = nullptr;  // 2 pairs x 2 x npix
  size_t zfused_n = 0;
  bool zfused_clean[2] = {false, false};
  size_t zfused_clean_npix = 0;  // the image size the clean pairs were cleared for (a pair holds zbuf0 | zbuf1 at THAT size)
  int zfused_next = 0;
  bool starve_fused = true;  // MRH_STARVE_FUSED=0: the eight launches of rounds 1-5 (k_starve<0,1,2>, k_summarize_visible, k_free_lists)
  uint64_t n_starve_fused = 0;
  int4* d_realloc = nullptr;
  int4* d_reint = nullptr;
  int* d_flag = nullptr;
  u64* d_upd_partials = nullptr;
  u32* d_misc = nullptr;  // 4 words for k_get_voxel
  float* d_rcp_w = nullptr;  // Fast::rcp_w
  Fast fast;              // fast path buffers
  size_t fast_npix = 0;
  uint2* dcx_buf = nullptr;  // {cleaned depth, packed colour} of the current frame (written by k_front)
  // ---- pipelined frames (integrate_lazy; MRH_PIPE=0 keeps the two serial launches on one stream) ----
  // The front half of a frame (k_front<..., LAZY>) is launched on `stream_front`, its integration (k_back<..., LZ = 2>) on the
  // main stream behind one event; the front stream never waits for the main one, so the front half of frame g + 1 runs next to
  // the integration of frame g.  Up to kPipeRing - 1 frames are in flight, each with its own {depth, colour} image, lists,
  // list-counter set and want stamps.  Everything that is not a pipelined frame meets the map only after k_reclaim.
  int pipe = 1;
  int pipe_grid = 1024;                     // workgroups of a pipelined integration: ONE resident generation (4 per CU x 256 CUs).  With 2048 the
                                            // second generation competes with the front half's workgroups for the slots the first one frees: 37.3 against
                                            // 34.8 us per frame (MRH_PIPE_GRID; the serial launch keeps 2048)
  int pipe_uploads = 0;                     // MRH_PIPE_UPLOADS=1: pipeline frames whose images came through mrh_upload_* too
  int pipe_period = 64;                     // the reclaim (and one serial frame) every so many pipelined frames; MRH_PIPE_PERIOD.  (32 until
                                            // round 6: a period boundary costs the pipeline ~60 us, the zombies it bounds are also bounded by the
                                            // pool test below (zombies <= pool / 8); 64 — the census period — gave +4 % at 100 steps, 128 no more)
  hipStream_t stream_front = nullptr;
  hipEvent_t ev_front[kPipeRing] = {};
  uint2* pipe_dcx[kPipeRing] = {};
  size_t pipe_npix = 0;
  int4* ring_vis[kPipeRing] = {}; int4* ring_bbox[kPipeRing] = {}; int4* ring_cfree[kPipeRing] = {}; float* ring_zmin[kPipeRing] = {};  // [0] = the context's own
  u32* want_ring = nullptr;                 // kPipeRing x slots stamps
  int* h_levels = nullptr;                  // pinned {fine free-list level, zombies, sequence number of the last integration that started}
  uint64_t pipe_seq = 0;                    // frames issued by integrate_lazy (pipelined or not)
  uint64_t pipe_base = 0;                   // every frame below this sequence number is known complete (host synchronised)
  int lazy_run = 0;                         // pipelined frames since the last reclaim
  bool zombies_possible = false;
  bool last_frame_lazy = false;
  bool flushed_since_frame = false;          // an entry point other than the per-frame ones ran since the last frame
  int sync_streak = 0;
  bool front_needs_sync = false;            // the main stream changed the table / free list behind the front stream's back
  // the integration of the newest pipelined frame is enqueued by the NEXT mrh_integrate (or by whichever other entry point comes
  // first): by then its front half has usually finished, the host sees that (hipEventQuery) and the main stream needs no
  // cross-stream wait in front of the launch — such a wait costs ~6 us of idle main stream per frame on this runtime
  struct PendingBack {
    bool on = false;
    Cam cam;
    Fast f;
    Lists L;
    int set = 0, zero_set = 0, ring = 0, seq = 0;
    u32 stamp = 0;
    float thr = 0.f;
    bool free_ = false, profile = false, safe_div = false, count_zombies = false, sph = false;
    bool starve = false;  // a starve frame: behind the integration (which collects nothing) the three fused starve launches
    EvPair ev = {nullptr, nullptr};
    uint64_t report_seq = 0;  // frame mark whose pool report was written before this integration ran (refreshed behind it)
  };
  static constexpr int kPendMax = 3;
  PendingBack pendq[kPendMax];            // oldest first
  int npend = 0;
  int pipe_defer = 1;                     // integrations kept back (MRH_PIPE_DEFER, 1 .. kPendMax - 1): the older a front half, the surer it has finished
  uint64_t dbg_waits = 0;
  double dbg_spin_us = 0, dbg_api_us = 0; uint64_t dbg_lazy_frames = 0;  // MRH_DEBUG: where the host's time in integrate_lazy goes
  int4* d_cfree = nullptr;
  // LiDAR scan of the current frame (mrh_lidar.h)
  float* d_cloud = nullptr; size_t cloud_n = 0;  // spherical camera: getDepth(cloud) image of the current frame (k_cloud_depth)
  float* d_normals = nullptr; uint64_t normals_cap = 0, num_normals = 0;  // one normal per point (mrh_upload_normals)
  bool frame_general = false;       // this frame ran through the general kernels (mrh_kernels.h): GC by k_gc_identify / k_gc_free
  bool fast_summaries_stale = false;  // single-resolution map: a general frame left Fast::summary behind
  float* d_points = nullptr;        // owned copy (mrh_upload_points) ...
  const float* d_points_cur = nullptr;  // ... or the caller's device pointer (mrh_set_points_device)
  uint64_t points_cap = 0, num_points = 0;
  u32* d_pt_counts = nullptr; u32* d_pt_offsets = nullptr; uint64_t pt_cap = 0;
  u32* h_scan = nullptr;               // pinned {hwm, last offset, last count, sequence}: the one report of a scan
  u32 scan_seq = 0;
  void* d_rec_keys[2] = {nullptr, nullptr}; float* d_rec_vals[2] = {nullptr, nullptr}; uint64_t rec_cap = 0; size_t rec_key_bytes = 0;
  void* d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
  // voxel-bucket scans (mrh_scan.h): per-voxel counters + block stamps (allocated with the first scan), stash, placed records, chunks
  Scan scan = {};
  int lidar_buckets = 1;         // MRH_LIDAR_BUCKETS=0: scans through the sorted records of mrh_lidar.h (cross-check)
  int scan_state = 0;            // 0: scratch not tried yet, 1: allocated, -1: does not fit / not applicable (sorted path)
  bool scan_dirty = false;       // a scan failed half way: the counters are cleared before the next one
  uint64_t scan_rec_cap = 0, scan_wg_cap = 0;
  u32* d_scan_ctr = nullptr;
  u32 scan2_seq = 0;
  size_t scan_lds_set = 0;
  // 3DGS splat seeds (mrh_splat.h): sized for one (image shape, min pixel size)
  QTree qt = {0, 0, 0, 0, 0};
  QSum* d_qt_sums = nullptr; u32* d_qt_flags = nullptr; u32* d_qt_unc = nullptr; u64* d_qt_marks = nullptr; u64* d_qt_pos = nullptr;
  mrh_splat_seed* d_qt_parked = nullptr; mrh_qtree_leaf* d_qt_leaves = nullptr;
  // what the caller takes from a seeding call is written by its last launch straight into pinned host memory (a few hundred to a few
  // thousand 20-byte seeds and two counters): one synchronisation, no transfer calls (they were two pageable read-backs, each behind
  // a synchronisation of its own: ~35 of the call's 135 us)
  mrh_splat_seed* h_qt_seeds = nullptr; u32 qt_seed_cap = 0;
  u64* h_qt_out = nullptr;   // [0] totals (leaves | seeds << 32), [1] literal evaluations
  u64* d_qt_misc = nullptr;  // [0] totals (leaves | seeds << 32), [1] uncertain-node counter (low word)
  int qt_literal = 0;        // MRH_QTREE_LITERAL=1: every node error through the reference's summation order (cross-check)
  uint32_t qt_last_literal = 0;
  std::vector<mrh_qtree_leaf> qt_leaves;
  uint64_t qt_n_leaves = 0;            // leaves of the last mrh_splat_seeds, still on the device (d_qt_leaves) until someone asks
  bool qt_leaves_on_host = true;
  int scan_layout_hint = 0;    // mrh_set_scan_layout / MRH_SCAN_ROW_LEN: > 0 points per row of the caller's organised scans, 0 find out (host clouds), < 0 none
  int scan_row_len = 0;        // ... of the CURRENT cloud (0: not organised, or not known)
  uint64_t scan_detect_n = 0;  // the look at a host cloud is repeated when the cloud's size changes and every 64th upload (a sensor keeps its layout;
  int scan_detect_len = 0, scan_detect_age = 0;  // the look itself costs the calling thread ~20 us of cache misses, more than the order wins per scan)
  int scan_patch_log2 = 4;     // MRH_SCAN_PATCH_LOG2: columns (log2) of the beam patch a walk workgroup takes from an organised scan; 8 = 256 consecutive points
  int mr_fused = 1;          // MRH_MR_FUSED=0: multi-resolution maps always through the general kernels (mrh_kernels.h)
  bool mr_next_general = true;    // the next multi-resolution frame must take the general path (frame 0 / after a starve frame / after an import)
  bool mr_summaries_valid = false;  // fast.summary / summary_c describe every live block (the general kernels do not maintain them)
  bool frame_fused_mr = false;
  bool refill_flag_valid = false;  // d_flag holds the refill test for the next fused frame (taken by k_mr_tail)
  int mesh_on_host = 0;      // MRH_MESH_HOST=1: mesh post-process with the host restatement instead of mrh_mesh.h
  float* d_zmin = nullptr;   // per visible-list entry (Lists::zmin)
  uint64_t fast_frames = 0;  // fast-path frames issued: parity selects the list-counter set
  int frame_parity = 0;
  u64* d_cnt_partials = nullptr;
  int fused_grid = 2048;  // x 4 waves
  int sweep_wgs_mr = 1024; // the same for multi-resolution maps (9x the descriptors); MRH_SWEEP_WGS_MR
  int sweep_wgs = 128;    // descriptor-sweep workgroups appended to the allocation launch (k_front)
  bool frame_gc_inline = false;
  int integrate_grid = 1024;
  int low_blocks_to_allocate = 0;
  uint64_t num_blocks = 0, slots = 0, max_triangles = 0;
  uint64_t frames = 0;
  int pending = 0;           // sharded starve frames: 1 after pass 0, 2 after pass 1
  int pending_max_frames = 0;
  // mesh (host)
  HostVec<mrh_triangle> tris;   // host copy of the soup: only when the caller of mrh_extract_triangles asks for it
  std::vector<mrh_block_desc> tri_blocks;
  std::vector<uint32_t> tri_counts;
  // ... of the last extraction, still on the device (arena slot 0) until mrh_get_triangle_blocks asks
  int tri_dev_n = 0;
  const int4* d_tri_sorted = nullptr;
  const u32* d_tri_counts = nullptr;
  u64* h_mc = nullptr;  // pinned: triangle total of the extraction in flight
  hipEvent_t ev_mc_total = nullptr;  // ... has landed
  u32* d_mc_recs = nullptr; size_t mc_rec_cap = 0;  // corner records of the count pass (mrh_mc.h McRecords), grow-only
  uint64_t mc_rec_fallbacks = 0;                    // extractions whose records did not fit (emitted by k_mc<emit> instead)
  HostVec<double> V, C;
  HostVec<int32_t> F;
  // V and C cross the link in fp32 (k_stage_out) and are widened by the host while the rest is still on its way (widen_from_staging)
  HostVec<float> V32, C32;     // pinned staging
  HostVec<u32> stage_ctl;      // pinned: [0..5] {vertices, faces, epoch} as three u64, [16..] one flag word per 64 KiB chunk of V32, then of C32
  u32 stage_epoch = 0;
  void* mesh_clean_base = nullptr;  // arena slot 1 as the last extraction left it: the first mesh_clean_words words are 0xFFFFFFFF
  size_t mesh_clean_words = 0;
  // The host side of a context's FIRST extraction — four pinned mappings (mmap + first touch + hipHostRegister: ~0.9 ms for the
  // 20 MB of a 0.5 M-triangle mesh) and the 24 MB of doubles the caller sees — used to be paid inside that call, after a
  // synchronisation that told it the sizes: 2.8 ms where every later extraction takes 0.85, and a one-shot extractMesh (what
  // every runner of the reference does) only ever makes the first.  A context that fuses frames will be asked for its mesh: at
  // the end of its THIRD mrh_integrate — a context is still allocating and warming up there — those buffers are sized from the
  // live blocks (64 vertices a block: twice what the rooms of the benchmarks yield, so a map that keeps growing still fits).
  // The first extraction then finds its staging ready and runs like any other; if the estimate was short, it grows the
  // buffers as before.  Done in the calling thread, once: a helper thread was built first (round 6) and slowed the frame loop
  // by 8 % for as long as it was faulting pages in, at whichever frame it was started.  MRH_PREWARM=0 switches it off.
  bool prewarm_on = true, prewarm_done = false;
  uint64_t n_extractions = 0;
  bool f64_link = false;       // MRH_MESH_F64_LINK=1: V / C widened on the device and copied as doubles (the round-3 path; A/B, tests)
  // profiling
  int profile = 0;
  std::vector<EvPair> ev_pool;
  std::vector<EvPair> ev_pending;
  std::vector<EvPair> ev_pending_front;  // the allocation launch (k_front) of profiled fast-path frames
  float sum_ms = 0.f, last_ms = 0.f;
  uint64_t n_ms = 0;
  float sum_front_ms = 0.f;
  uint64_t n_front_ms = 0;
  uint64_t prev_total_updated = 0, prev_inserted = 0, prev_freed = 0, total_compact = 0;
  uint64_t last_triangles = 0;
  // hash-table upkeep (mrh_kernels.h: k_table_census / k_rehash_*)
  int census_period = 64;          // frames between two censuses; MRH_REHASH_PERIOD
  int census_force = 0;            // MRH_REHASH_FORCE=1: every census rebuilds (tests)
  uint64_t frames_since_census = 0;
  bool table_dirty = false;        // bulk erase / insert since the last census (stream-out, import, drop): census before the next frame
  // device error flags: `flags_seen` = union of everything taken off the device since create / reset (stats),
  // `flags_deferred` = taken but not yet returned to the caller by mrh_sync, `flags_peeked` = already returned by a peek
  u32 flags_seen = 0, flags_deferred = 0, flags_peeked = 0;
  // multi-GPU block exchange
  char* d_pack = nullptr; size_t pack_cap = 0;      // mrh_pack_blocks result (records)
  mrh_triangle* d_soup = nullptr; size_t soup_cap = 0, soup_n = 0;  // triangle soup of the last extraction / run merge (mrh_get_triangles_device)
  int4* d_halo = nullptr; size_t halo_cap = 0, halo_upper = 0;  // blocks brought in by MRH_UNPACK_HALO (upper bound of the device count)
  u32* d_taken = nullptr;
  // marching cubes timing (mrh_stats)
  hipEvent_t mc_ev[4] = {};
  float last_mc_count_ms = 0.f, last_mc_emit_ms = 0.f;
  uint64_t last_mc_blocks = 0;
  // MeshExtractor::merge_mesh_ (mrh_mesh_merge_begin / _end): the soups of the extractions in between, back to back
  bool merge_on = false;
  mrh_triangle* d_acc = nullptr; size_t acc_cap = 0, acc_n = 0;
  // RCCL (mrh_comm.h): the communicator this context is attached to, exchange buffers, phase clocks
  mrh_comm* comm = nullptr;
  char* d_xsend = nullptr; size_t xsend_cap = 0;
  char* d_xrecv = nullptr; size_t xrecv_cap = 0;
  hipEvent_t comm_ev[5] = {};
  mrh_comm_phases comm_phases = {};
  std::vector<EvPair> comm_ev_pool, comm_ev_pending;
  std::string err;
};

static int comm_allreduce_zbuf(mrh_ctx* c, mrh::u64* buf, size_t n);  // mrh_comm.h
static void comm_release(mrh_ctx* c);
static bool comm_matches_sharding(const mrh_ctx* c, int* comm_rank, int* comm_world);

namespace {

int fail(mrh_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  else g_create_err = buf;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                                   \
  do {                                                                                                       \
    hipError_t e__ = (expr);                                                                                 \
    if (e__ != hipSuccess) return fail(ctx, MRH_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

// device scratch that is released on every path out of a function (error returns included)
template <typename T>
struct DevBuf {
  T* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void) hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**) &p, n * sizeof(T)); }
  operator T*() const { return p; }
};

uint64_t next_pow2(uint64_t v) {
  uint64_t r = 1;
  while (r < v) r <<= 1;
  return r;
}

void free_all(mrh_ctx* c) {
  if (!c) return;
  (void) hipSetDevice(c->device);
  for (UpRing* r : {&c->up_depth, &c->up_rgb})
    if (r->stream) { (void) hipStreamSynchronize(r->stream); (void) hipStreamDestroy(r->stream); }
  if (c->stream_front) (void) hipStreamSynchronize(c->stream_front);
  if (c->stream) (void) hipStreamSynchronize(c->stream);
  auto F = [](void* p) { if (p) (void) hipFree(p); };
  F(c->dcx_buf); F(c->want_ring); F(c->fast.zlist);
  for (int i = 0; i < kPipeRing; i++) { F(c->pipe_dcx[i]); if (i) { F(c->ring_vis[i]); F(c->ring_bbox[i]); F(c->ring_cfree[i]); F(c->ring_zmin[i]); } if (c->ev_front[i]) (void) hipEventDestroy(c->ev_front[i]); }
  if (c->h_levels) (void) hipHostFree(c->h_levels);
  if (c->stream_front) { (void) hipStreamSynchronize(c->stream_front); (void) hipStreamDestroy(c->stream_front); }
  F(c->tab.keys); F(c->tab.vals); F(c->tab.heap_fine); F(c->tab.heap_coarse); F(c->tab.desc_fine); F(c->tab.desc_coarse);
  F(c->tab.pool); F(c->tab.compact); F(c->tab.ctr); F(c->tab.prof);
  for (UpRing* r : {&c->up_depth, &c->up_rgb})
    for (UpSlot& u : r->s) {
      if (u.h) (void) hipHostFree(u.h);
      F(u.d);
      if (u.copied) (void) hipEventDestroy(u.copied);
    }
  for (hipEvent_t e : c->frame_done) if (e) (void) hipEventDestroy(e);
  for (hipEvent_t e : c->peek_done) if (e) (void) hipEventDestroy(e);
  if (c->h_peek) (void) hipHostFree(c->h_peek);
  if (c->h_mc) (void) hipHostFree(c->h_mc);
  if (c->h_scan) (void) hipHostFree(c->h_scan);
  for (void* a : c->arena) if (a) (void) hipFree(a);
  F(c->d_decision); F(c->d_zbuf); F(c->d_zfused); F(c->d_realloc); F(c->d_reint); F(c->d_flag);
  F(c->d_upd_partials); F(c->d_misc); F(c->d_rcp_w); F(c->d_cfree); F(c->d_zmin); F(c->d_points); F(c->d_pt_counts); F(c->d_pt_offsets); F(c->d_rec_keys[0]); F(c->d_rec_keys[1]); F(c->d_rec_vals[0]); F(c->d_rec_vals[1]); F(c->d_sort_tmp); F(c->scan.vcnt); F(c->scan.bstamp); F(c->scan.st_meta); F(c->scan.st_sdf); F(c->scan.st_grp); F(c->scan.wgdesc); F(c->scan.rec); F(c->scan.chunks); F(c->d_scan_ctr); F(c->fast.summary); F(c->fast.summary_c); F(c->fast.bbox); F(c->d_cnt_partials);
  F(c->d_pack); F(c->d_halo); F(c->d_taken); F(c->d_cloud); F(c->d_normals); F(c->d_soup); F(c->d_mc_recs);
  for (hipEvent_t e : c->mc_ev) if (e) (void) hipEventDestroy(e);
  if (c->ev_mc_total) (void) hipEventDestroy(c->ev_mc_total);
  comm_release(c);
  F(c->d_xsend); F(c->d_xrecv); F(c->d_acc);
  for (hipEvent_t e : c->comm_ev) if (e) (void) hipEventDestroy(e);
  for (auto& e : c->comm_ev_pool) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
  for (auto& e : c->comm_ev_pending) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
  F(c->d_qt_sums); F(c->d_qt_flags); F(c->d_qt_unc); F(c->d_qt_marks); F(c->d_qt_pos); F(c->d_qt_parked); F(c->d_qt_leaves); F(c->d_qt_misc);
  if (c->h_qt_seeds) (void) hipHostFree(c->h_qt_seeds);
  if (c->h_qt_out) (void) hipHostFree(c->h_qt_out);
  for (int i = 0; i < c->npend; i++) if (c->pendq[i].profile) c->ev_pool.push_back(c->pendq[i].ev);
  c->npend = 0;
  for (auto& e : c->ev_pool) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
  for (auto& e : c->ev_pending) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
  for (auto& e : c->ev_pending_front) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
  if (c->stream) (void) hipStreamDestroy(c->stream);
}

// (re)initialises every device structure to the empty map (voxel_data_structures.cpp:58-87 + ctor counters)
int init_buffers(mrh_ctx* c) {
  hipStream_t s = c->stream;
  c->mr_next_general = true;
  c->refill_flag_valid = false;
  c->mr_summaries_valid = false;
  c->fast_frames = 0;
  if (c->stream_front) HIP_TRY(c, hipStreamSynchronize(c->stream_front));
  for (int i = 0; i < c->npend; i++) if (c->pendq[i].profile) c->ev_pool.push_back(c->pendq[i].ev);
  c->npend = 0;  // a reset map has nothing left to integrate
  c->pipe_seq = 0;
  c->pipe_base = 0;
  c->lazy_run = 0;
  c->zombies_possible = false;
  c->front_needs_sync = false;
  if (c->h_levels) { c->h_levels[0] = (int) c->num_blocks - 1; c->h_levels[1] = 0; c->h_levels[2] = -1; }
  if (c->want_ring) HIP_TRY(c, hipMemsetAsync(c->want_ring, 0, (size_t) kPipeRing * c->slots * sizeof(u32), s));
  const Tab& t = c->tab;
  k_init_table<<<1024, 256, 0, s>>>(t.keys, c->slots);
  k_init_heap<<<1024, 256, 0, s>>>(t.heap_fine, (u32) c->num_blocks, getenv("MRH_DEBUG_HEAP_DESCENDING") ? 1 : 0);
  HIP_TRY(c, hipMemsetAsync(t.vals, 0, c->slots * sizeof(u32), s));
  HIP_TRY(c, hipMemsetAsync(t.desc_fine, 0, c->num_blocks * sizeof(int4), s));
  if (t.multi_res) HIP_TRY(c, hipMemsetAsync(t.desc_coarse, 0, c->num_blocks * 8 * sizeof(int4), s));
  HIP_TRY(c, hipMemsetAsync(t.pool, 0, c->num_blocks * (size_t) kFineBytes, s));
  int h_ctr[CTR_COUNT];
  memset(h_ctr, 0, sizeof h_ctr);
  h_ctr[CTR_HEAP_FINE] = (int) c->num_blocks - 1;  // voxel_data_structures.cuh:91-92
  h_ctr[CTR_HEAP_COARSE] = -1;                     // :94-95
  HIP_TRY(c, hipMemcpyAsync(t.ctr, h_ctr, sizeof h_ctr, hipMemcpyHostToDevice, s));
  HIP_TRY(c, hipMemsetAsync(t.prof, 0, PROF_COUNT * sizeof(u64), s));
  HIP_TRY(c, hipMemsetAsync(c->d_upd_partials, 0, (size_t) c->integrate_grid * sizeof(u64), s));
  HIP_TRY(c, hipMemsetAsync(c->d_cnt_partials, 0, (size_t) 32768 * 4 * sizeof(u64), s));
  HIP_TRY(c, hipStreamSynchronize(s));
  c->frames = 0;
  c->frames_since_census = 0;
  c->table_dirty = false;
  c->flags_seen = c->flags_deferred = c->flags_peeked = 0;
  c->halo_upper = 0;
  c->prev_total_updated = c->prev_inserted = c->prev_freed = c->total_compact = 0;
  c->sum_ms = c->last_ms = 0.f;
  c->n_ms = 0;
  c->sum_front_ms = 0.f;
  c->n_front_ms = 0;
  c->tris.clear(); c->V.clear(); c->C.clear(); c->F.clear();
  c->last_triangles = 0;
  return MRH_OK;
}

int drain_events(mrh_ctx* c) {
  for (auto& e : c->ev_pending) {
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, e.a, e.b));
    c->sum_ms += ms;
    c->last_ms = ms;
    c->n_ms++;
    c->ev_pool.push_back(e);
  }
  c->ev_pending.clear();
  for (auto& e : c->ev_pending_front) {
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, e.a, e.b));
    c->sum_front_ms += ms;
    c->n_front_ms++;
    c->ev_pool.push_back(e);
  }
  c->ev_pending_front.clear();
  return MRH_OK;
}

int check_device_flags(mrh_ctx* c, u32 flags) {
  // the flags are already cleared on the device (take_device_flags): whatever else is reported first, a scan that left its
  // bounds has left counters behind, and the next scan must start from zero
  if (flags & ERR_SCAN) c->scan_dirty = true;
  if (flags & ERR_RANGE) return fail(c, MRH_ERR_OUT_OF_RANGE, "a block coordinate left the packed-key range of +-2^20 blocks");
  if (flags & ERR_POOL) return fail(c, MRH_ERR_CAPACITY, "SDF block pool exhausted (num_sdf_blocks = %llu)", (unsigned long long) c->num_blocks);
  if (flags & ERR_TABLE) return fail(c, MRH_ERR_CAPACITY, "hash table probe limit reached (hash_slots = %llu)", (unsigned long long) c->slots);
  if (flags & ERR_TRI) return fail(c, MRH_ERR_CAPACITY, "triangle buffer full (max_triangles = %llu)", (unsigned long long) c->max_triangles);
  if (flags & ERR_SCAN) return fail(c, MRH_ERR_DEVICE, "a LiDAR scan left its bounds (voxels per beam, touched blocks or chunks): the map is not usable");
  return MRH_OK;
}

// Device error flags are taken off the device and cleared there in one stream-ordered step (nothing else runs on the
// stream in between), so a flag is reported for the call that raised it and not for every later one.
int take_device_flags(mrh_ctx* c, u32* out) {
  u32 flags = 0;
  HIP_TRY(c, hipMemcpyAsync(&flags, &c->tab.ctr[CTR_ERROR], sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (flags) HIP_TRY(c, hipMemsetAsync(&c->tab.ctr[CTR_ERROR], 0, sizeof(u32), c->stream));
  if (flags & ERR_POOL) c->table_dirty = true;  // keys without storage exist (publish_without_storage): census + rebuild before the next frame,
                                                // so that the positions can be allocated again as soon as the pool has room (vds.cu:566-569 retries every frame)
  c->flags_seen |= flags;
  *out = flags;
  return MRH_OK;
}

// Table upkeep between two frames (mrh_kernels.h): census of the tombstones every `census_period` frames or after a bulk
// change, rebuild decided on the device.  Four short launches, no host round trip.
int maintain_table(mrh_ctx* c, bool force_census) {
  if (c->pending || c->census_period < 0) return MRH_OK;
  if (!force_census && !c->table_dirty && c->frames_since_census < (uint64_t) c->census_period) return MRH_OK;
  hipStream_t s = c->stream;
  const Tab& t = c->tab;
  const int grid = (int) std::min<uint64_t>(2048, (c->slots + 255) / 256);
  k_table_census<<<grid, 256, 0, s>>>(t, (size_t) c->slots);
  k_rehash_decide<<<1, 1, 0, s>>>(t, (u32) (c->slots / 4), c->census_force);
  k_rehash_clear<<<grid, 256, 0, s>>>(t, (size_t) c->slots);
  k_rehash_insert<<<1024, 256, 0, s>>>(t);
  c->frames_since_census = 0;
  c->table_dirty = false;
  HIP_TRY(c, hipGetLastError());
  return MRH_OK;
}

int ensure_device(mrh_ctx* c, const char* who) {
  if (!c) return MRH_ERR_INVALID_ARG;
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(c, MRH_ERR_DEVICE, "%s: hipSetDevice failed: %s", who, hipGetErrorString(e));
  return MRH_OK;
}
int strict_point(mrh_ctx* c);
// every entry point except the per-frame ones (setters, mrh_integrate, the non-blocking peeks): behind the pipelined frames issued
// so far, the zombies nobody wanted leave the table, so that whatever the call reads, changes or waits for is exactly the map two
// serial launches per frame would have left
int flush_deferred(mrh_ctx* c);
int ensure_ready(mrh_ctx* c, const char* who) {
  int rc = ensure_device(c, who);
  if (rc) return rc;
  rc = flush_deferred(c);  // a host-fed frame that mrh_integrate kept back runs before anything else looks at the map
  if (rc < 0) return rc;
  if (c->stream_front) c->front_needs_sync = true;  // whatever this call does to the map, the front stream must see it before its next launch
  c->flushed_since_frame = true;
  return strict_point(c);
}

// compacts every live block (no frustum filter) and returns the count; blocking
int compact_all(mrh_ctx* c, int* out_n) {
  hipStream_t s = c->stream;
  HIP_TRY(c, hipMemsetAsync(&c->tab.ctr[CTR_COMPACT], 0, sizeof(int), s));
  k_compact<<<512, 256, 0, s>>>(c->cam, c->map, c->tab, 0);
  if (!c->h_mc) {  // pinned: a copy into a stack variable is staged by the runtime
    HIP_TRY(c, hipHostMalloc((void**) &c->h_mc, 8 * sizeof(u64), hipHostMallocDefault));
    memset(c->h_mc, 0, 8 * sizeof(u64));
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_mc + 4, &c->tab.ctr[CTR_COMPACT], sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  *out_n = *(const int*) (c->h_mc + 4);
  return MRH_OK;
}

struct KeyHash3 {
  size_t operator()(const std::array<uint64_t, 3>& k) const {
    uint64_t h = k[0] * 0x9E3779B97F4A7C15ull;
    h ^= (k[1] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
    h ^= (k[2] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
    return (size_t) h;
  }
};
struct FaceHash {
  size_t operator()(const std::array<int32_t, 3>& f) const {
    uint64_t h = (uint64_t) (uint32_t) f[0] * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t) (uint32_t) f[1] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
    h ^= ((uint64_t) (uint32_t) f[2] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
    return (size_t) h;
  }
};

// MeshExtractor::processTriangles for a single extraction (mesh_extractor.cpp:9-76):
// soup -> vertex merge (exact bit pattern, or floor(v/eps) cells; first occurrence keeps index and colour)
// -> drop degenerate faces -> drop repeated faces keeping the first.
void widen_quiesce();
void process_triangles(mrh_ctx* c) {
  const size_t nt = c->tris.size();
  widen_quiesce();  // a helper of the last extraction's widening may still be writing the arrays that are about to be replaced
  c->V.clear(); c->C.clear(); c->F.clear();
  if (nt == 0) return;
  std::vector<double> V, C;
  std::vector<int32_t> F;
  const double eps = (double) c->p.vertices_merging_threshold;
  const double inv_eps = eps != 0.0 ? 1.0 / eps : 0.0;
  std::unordered_map<std::array<uint64_t, 3>, int32_t, KeyHash3> vmap;
  vmap.reserve(nt * 3);
  std::vector<int32_t> faces(nt * 3);
  for (size_t i = 0; i < nt; i++)
    for (int k = 0; k < 3; k++) {
      const mrh_vertex& v = c->tris[i].v[k];
      const double p[3] = {(double) v.p[0], (double) v.p[1], (double) v.p[2]};
      std::array<uint64_t, 3> key;
      for (int a = 0; a < 3; a++) {
        if (eps == 0.0) memcpy(&key[a], &p[a], 8);
        else key[a] = (uint64_t) (uint32_t) (int32_t) std::floor(p[a] * inv_eps);
      }
      const bool has_nan = p[0] != p[0] || p[1] != p[1] || p[2] != p[2];  // never equal to anything (Vector3dEqual)
      auto it = has_nan ? vmap.end() : vmap.find(key);
      int32_t idx;
      if (it != vmap.end()) idx = it->second;
      else {
        idx = (int32_t) (V.size() / 3);
        if (!has_nan) vmap.emplace(key, idx);
        V.insert(V.end(), {p[0], p[1], p[2]});
        C.insert(C.end(), {(double) v.c[0], (double) v.c[1], (double) v.c[2]});
      }
      faces[i * 3 + k] = idx;
    }
  std::unordered_map<std::array<int32_t, 3>, char, FaceHash> seen;
  seen.reserve(nt);
  for (size_t i = 0; i < nt; i++) {
    const std::array<int32_t, 3> f = {faces[i * 3], faces[i * 3 + 1], faces[i * 3 + 2]};
    if (f[0] == f[1] || f[0] == f[2] || f[1] == f[2]) continue;
    if (!seen.emplace(f, 1).second) continue;
    F.insert(F.end(), {f[0], f[1], f[2]});
  }
  c->V.assign(V.data(), V.data() + V.size());
  c->C.assign(C.data(), C.data() + C.size());
  c->F.assign(F.data(), F.data() + F.size());
}

// grow-only scratch `slot` of at least `bytes` (contents undefined); the previous buffer is released only after the stream drained
int arena_get(mrh_ctx* c, const int slot, const size_t bytes, void** out) {
  if (bytes > c->arena_cap[slot]) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->arena[slot]) HIP_TRY(c, hipFree(c->arena[slot]));
    c->arena[slot] = nullptr; c->arena_cap[slot] = 0;
    if (slot == 1) c->mesh_clean_words = 0;  // new memory: the post-process's tables are not the empty ones it left behind
    const size_t cap = bytes + bytes / 4;
    HIP_TRY(c, hipMalloc(&c->arena[slot], cap));
    c->arena_cap[slot] = cap;
  }
  *out = c->arena[slot];
  return MRH_OK;
}

// Results leave the device through a copy KERNEL writing pinned host memory, not through hipMemcpyAsync: the runtime's choice
// of SDMA engine for a stream is not stable within a process — the second context of a process (and every later one) moved its
// V / C / F at 22 GB/s instead of 54 (tools/dbg_extract2.py: 30 MB in 1.24 vs 0.56 ms; tools/micro/d2h_streams.hip and
// d2h_second_alloc.hip rule out the host buffer and the stream order in isolation) while 16-byte stores of a kernel reach 53-54 GB/s
// every time.  It also lets the copy read its sizes on the device: no host round trip between the post-process and the copy.
//   part p copies ceil(min(count[p], cap[p]) * unit[p] / 16) 16-byte words (both sides are padded to a multiple of 16 bytes)
struct CopyOut {
  const uint4* src[3];
  uint4* dst[3];
  const u64* count[3];  // device: elements of part p (nullptr: use fixed[p])
  u64 fixed[3], cap[3];
  u32 unit[3];          // bytes per element
};
__global__ __launch_bounds__(256) void k_copy_out(const CopyOut a) {
#pragma unroll 1
  for (int p = 0; p < 3; p++) {
    if (!a.dst[p]) continue;
    u64 n = a.count[p] ? *a.count[p] : a.fixed[p];
    if (n > a.cap[p]) n = a.cap[p];
    const size_t n16 = (size_t) ((n * a.unit[p] + 15) / 16);
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) a.dst[p][i] = a.src[p][i];
  }
}
// V / C / F of an extraction are 48 B per vertex + 12 B per face as the reference hands them out (Eigen::MatrixXd / MatrixXi,
// geowrapper.h:91-93) — 30 MB at the driver's workload, 0.56 ms of a 1.3 ms extraction at the link's 54 GB/s.  The vertex
// arithmetic is fp32 (mesh_extractor.cu:6-36), so the doubles carry no more than the floats they are widened from: V and C cross the
// link as fp32 (24 B per vertex) into pinned staging, in 64 KiB chunks that each raise a flag word when they have landed, and
// host threads widen chunk after chunk into the caller-visible double arrays while the following chunks and the faces are still
// on the link (widen_from_staging below; (double) (float) is exact, the arrays are the same bytes as before).  F goes straight
// to its final buffer.  Few workgroups, each walking its chunks in order: the link is the bottleneck, and chunks must COMPLETE in
// order for the host to overlap, not all at the end.
constexpr u32 kStageChunk = 64u << 10;            // bytes
constexpr u32 kStageWords = kStageChunk / 16;     // uint4 per chunk
constexpr u32 kStageHdrWords = 16;                // u32 words of stage_ctl before the first flag
struct StageOut {
  const uint4* src[3];   // device: V32, C32, F
  uint4* dst[3];         // pinned: V32 staging, C32 staging, F
  const u64* totals;     // device: [0] vertices, [1] faces
  u64 cap_v, cap_f;      // elements the destinations hold
  u64* hdr;              // pinned: [0] vertices, [1] faces, [2] epoch (written last)
  u32* flags;            // pinned: chunk c of V32 -> flags[c], of C32 -> flags[flag_stride + c]
  u32 flag_stride;
  u32 epoch;
  // workgroups copy_wgs .. gridDim.x - 1 do not copy: they refill the post-process's index tables with "empty" for the next
  // extraction (the link keeps the copying workgroups busy for 0.35 ms; the fills used to cost 2 x 9 us up front)
  u32 copy_wgs;
  u32* clear;
  size_t clear_words;
};
__global__ __launch_bounds__(256) void k_stage_out(const StageOut a) {
  if (blockIdx.x >= a.copy_wgs) {
    const size_t n4 = a.clear_words / 4, stride = (size_t) (gridDim.x - a.copy_wgs) * 256;
    uint4* c4 = (uint4*) a.clear;
    const uint4 ff = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    for (size_t i = (size_t) (blockIdx.x - a.copy_wgs) * 256 + threadIdx.x; i < n4; i += stride) c4[i] = ff;
    if (blockIdx.x == a.copy_wgs && threadIdx.x < (a.clear_words & 3)) a.clear[n4 * 4 + threadIdx.x] = 0xFFFFFFFFu;
    return;
  }
  const u64 nv = a.totals[0], nf = a.totals[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.hdr[0] = nv; a.hdr[1] = nf;
    __hip_atomic_store(&a.hdr[2], (u64) a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (nv > a.cap_v || nf > a.cap_f) return;  // uniform: the host grows the buffers and launches again
  const size_t w16[3] = {(size_t) ((nv * 12 + 15) / 16), (size_t) ((nv * 12 + 15) / 16), (size_t) ((nf * 12 + 15) / 16)};
  const u32 nch[3] = {(u32) ((w16[0] + kStageWords - 1) / kStageWords), (u32) ((w16[1] + kStageWords - 1) / kStageWords),
                      (u32) ((w16[2] + kStageWords - 1) / kStageWords)};
  const u32 total = nch[0] + nch[1] + nch[2];
  for (u32 ch = blockIdx.x; ch < total; ch += a.copy_wgs) {  // uniform per workgroup
    const int p = ch < nch[0] ? 0 : (ch < nch[0] + nch[1] ? 1 : 2);
    const u32 lc = ch - (p > 0 ? nch[0] : 0u) - (p > 1 ? nch[1] : 0u);
    const size_t lo = (size_t) lc * kStageWords, hi = lo + kStageWords < w16[p] ? lo + kStageWords : w16[p];
    const uint4* __restrict__ src = a.src[p];
    uint4* __restrict__ dst = a.dst[p];
    uint4 r[kStageWords / 256];
#pragma unroll
    for (u32 k = 0; k < kStageWords / 256; k++) {
      const size_t i = lo + k * 256 + threadIdx.x;
      if (i < hi) r[k] = src[i];
    }
#pragma unroll
    for (u32 k = 0; k < kStageWords / 256; k++) {
      const size_t i = lo + k * 256 + threadIdx.x;
      if (i < hi) dst[i] = r[k];
    }
    if (p < 2) {
      // (the fence is needed — plain stores to the pinned buffer sit in the XCD's L2: with "stores acknowledged, then the flag" alone
      // tools/stress_extract.py read a torn mesh within 50 extractions — and it is not what the kernel waits for: write-through stores
      // (sc0 sc1) + acknowledgement + flag, no fence, gave the same 0.83-0.87 ms per extraction; profiles/r06/ab_stage_out_fences.txt)
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(&a.flags[(p ? a.flag_stride : 0u) + lc], a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// host side of the above (defined with the copy pool further down): widens nfloat floats of each of the two staging parts into
// dst[0] / dst[1], chunk by chunk as flags[part][chunk] reaches `epoch` (flags == nullptr: everything has landed already).
// `drained(arg)` tells whether the stream has run dry (then a missing flag means a failed launch).  Returns false if it gave up.
bool widen_from_staging(double* const dst[2], const float* const src[2], const volatile u32* const flags[2], u32 epoch, size_t nfloat,
                        bool (*drained)(void*), void* arg);
void widen_prewake();
void widen_quiesce();
uint64_t widen_redone();

// MeshExtractor::processTriangles on the device (mrh_mesh.h): fills V / C / F from a triangle soup in device memory.
// MRH_MESH_HOST=1 keeps the host restatement above (same arrays; tests compare the two).
int process_triangles_device(mrh_ctx* c, const mrh_triangle* d_tris, const size_t nt) {
  // a helper that lost its core during the last extraction's widening may still be reading the staging this one is about to
  // rewrite, or writing the arrays it may regrow (CopyPool: the call no longer waits for its helpers)
  widen_quiesce();
  c->V.clear(); c->C.clear(); c->F.clear();
  if (nt == 0) return MRH_OK;
  if (nt * 3 >= (1ull << 30)) return fail(c, MRH_ERR_CAPACITY, "mesh post-process: %zu triangles exceed the 2^30 soup vertices one index table holds", nt);
  hipStream_t s = c->stream;
  const u32 n = (u32) (nt * 3), ntr = (u32) nt;
  const double eps = (double) c->p.vertices_merging_threshold;
  const double inv_eps = eps != 0.0 ? 1.0 / eps : 0.0;
  const u32 cap = (u32) next_pow2((uint64_t) n * 2);  // load factor <= 1/2
  const u32 fcap = (u32) next_pow2((uint64_t) ntr * 2);
  MeshScratch m;
  m.bytes = (size_t) n * 4 * 5 + ((size_t) cap + fcap) * 4 + 32 * 256;
  {
    const int arc = arena_get(c, 1, m.bytes, &m.base);
    if (arc) return arc;
  }
  // The two index tables come first, so that they lie where the last extraction's lay: that extraction's read-back kernel left
  // them empty again (k_stage_out's extra workgroups clear them while the link is busy), and the two fills — 24 MB at the
  // driver's workload, ahead of the vertex and of the face kernels — are only needed when the scratch moved or grew.
  u32* table = m.take<u32>(cap);
  u32* ftable = m.take<u32>(fcap);
  const size_t clear_words = (size_t) ((ftable + fcap) - (u32*) m.base);
  const bool tables_clean = !c->f64_link && c->mesh_clean_base == m.base && c->mesh_clean_words >= clear_words && !getenv("MRH_MESH_FILL");
  c->mesh_clean_words = 0;  // dirty from here on, until a clear is enqueued
  const u32 vtiles = (n + kMeshTile - 1) / kMeshTile, ftiles = (ntr + 255) / 256;
  u32* rep = m.take<u32>(n);   u32* vloc = m.take<u32>(n);   u32* corner = m.take<u32>(n);
  u32* floc = m.take<u32>(n);  // faces (nt <= n)
  u32* tcount = m.take<u32>(vtiles);  u32* toff = m.take<u32>(vtiles);  // first occurrences per vertex tile, and their scan
  u32* fcount = m.take<u32>(ftiles);  u32* foff = m.take<u32>(ftiles);  // kept faces per face tile
  u64* d_totals = m.take<u64>(2);
  const u32 gv = (n + 255) / 256, gf = (ntr + 255) / 256;
  const float* soup = (const float*) d_tris;
  int rc = MRH_OK;
  void *dV = nullptr, *dC = nullptr;  // doubles with f64_link, floats otherwise
  int* dF = nullptr;
  if (!c->h_mc) {
    HIP_TRY(c, hipHostMalloc((void**) &c->h_mc, 8 * sizeof(u64), hipHostMallocDefault));
    memset(c->h_mc, 0, 8 * sizeof(u64));
  }
  const bool f64 = c->f64_link;
  if (!f64) widen_prewake();  // the helper threads are awake and spinning by the time the first chunk lands
#define MESH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(c, MRH_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); goto done; } } while (0)
  {
    // V, C (at most n vertices each) and the faces (at most nt) share slot 2, sized by those bounds: nothing of the
    // post-process waits for a count from the device
    {
      void* vcf = nullptr;
      const size_t vbytes = ((size_t) n * 3 * (f64 ? sizeof(double) : sizeof(float)) + 16 + 255) & ~(size_t) 255;  // + 16: the copy kernels read whole 16-byte words
      rc = arena_get(c, 2, 2 * vbytes + (size_t) ntr * 3 * sizeof(int) + 16, &vcf);
      if (rc) goto done;
      dV = vcf;
      dC = (char*) vcf + vbytes;
      dF = (int*) ((char*) vcf + 2 * vbytes);
    }
    // ---- vertices
    if (!tables_clean) MESH_TRY(hipMemsetAsync(m.base, 0xFF, clear_words * 4, s));
    k_mesh_vertex_insert<<<(n + kMeshTile - 1) / kMeshTile, kMeshTile, 0, s>>>(soup, n, eps, inv_eps, table, cap - 1, rep);
    k_mesh_vertex_rep<<<vtiles, kMeshTile, 0, s>>>(soup, n, eps, inv_eps, table, cap - 1, rep, vloc, tcount);
    k_tile_scan<<<1, 1024, 0, s>>>(tcount, vtiles, toff, d_totals);
    if (f64) k_mesh_emit_vertices<double><<<gv, 256, 0, s>>>(soup, rep, vloc, toff, n, (double*) dV, (double*) dC, corner);
    else k_mesh_emit_vertices<float><<<gv, 256, 0, s>>>(soup, rep, vloc, toff, n, (float*) dV, (float*) dC, corner);
    // ---- faces
    k_mesh_face_insert<<<gf, 256, 0, s>>>(corner, ntr, ftable, fcap - 1);
    k_mesh_face_keep<<<gf, 256, 0, s>>>(corner, ntr, ftable, fcap - 1, floc, fcount);
    k_tile_scan<<<1, 1024, 0, s>>>(fcount, ftiles, foff, d_totals + 1);
    k_mesh_emit_faces<<<gf, 256, 0, s>>>(corner, floc, foff, ntr, dF);
    const bool dbg = getenv("MRH_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // (Tried in round 4: V and C on a second stream as soon as the vertices are final, next to the face kernels.  The copy kernel and
    // k_mesh_face_insert do not share the memory system gracefully — the insert's atomics ran 37 -> 240-450 us whether the copy had
    // 1 024 or 128 workgroups — and the extraction took as long as before.)
    if (f64) {
      // ---- the round-3 way out: doubles over the link.  V, C, F go out behind the post-process without the host in between
      // (k_copy_out reads the two totals on the device), into the buffers of the previous extraction; if they turn out too small
      // (or not pinned) they grow and the copy runs again.
      const bool pinned = c->V.dev && c->C.dev && c->F.dev && !getenv("MRH_D2H_MEMCPY");  // MRH_D2H_MEMCPY=1: hipMemcpyAsync instead (A/B)
      auto copy_out = [&](const bool by_kernel, const size_t nv_known, const size_t nf_known) {
        if (by_kernel) {
          CopyOut a;
          a.src[0] = (const uint4*) dV; a.dst[0] = (uint4*) c->V.dev; a.count[0] = d_totals; a.cap[0] = c->V.cap / 3; a.unit[0] = 24;
          a.src[1] = (const uint4*) dC; a.dst[1] = (uint4*) c->C.dev; a.count[1] = d_totals; a.cap[1] = c->C.cap / 3; a.unit[1] = 24;
          a.src[2] = (const uint4*) dF; a.dst[2] = (uint4*) c->F.dev; a.count[2] = d_totals + 1; a.cap[2] = c->F.cap / 3; a.unit[2] = 12;
          a.fixed[0] = a.fixed[1] = a.fixed[2] = 0;
          k_copy_out<<<1024, 256, 0, s>>>(a);
          return hipGetLastError();
        }
        hipError_t e = hipMemcpyAsync(c->V.data(), dV, nv_known * 3 * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(c->C.data(), dC, nv_known * 3 * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && nf_known) e = hipMemcpyAsync(c->F.data(), dF, nf_known * 3 * sizeof(int), hipMemcpyDeviceToHost, s);
        return e;
      };
      const size_t cap_v = std::min(c->V.cap, c->C.cap) / 3, cap_f = c->F.cap / 3;
      const bool speculative = pinned && cap_v > 0 && c->V.data() && c->C.data() && c->F.data();
      if (speculative) MESH_TRY(copy_out(true, 0, 0));
      MESH_TRY(hipMemcpyAsync(c->h_mc + 2, d_totals, 2 * sizeof(u64), hipMemcpyDeviceToHost, s));
      MESH_TRY(hipStreamSynchronize(s));
      const double t1 = now();
      const size_t nv = (size_t) c->h_mc[2], nf = (size_t) c->h_mc[3];
      const bool fits = speculative && nv <= cap_v && nf <= cap_f;
      c->V.resize_discard(nv * 3); c->C.resize_discard(nv * 3); c->F.resize_discard(std::max<size_t>(nf, 1) * 3);
      c->F.n = nf * 3;
      if (!fits) {
        MESH_TRY(copy_out(c->V.dev && c->C.dev && c->F.dev && !getenv("MRH_D2H_MEMCPY"), nv, nf));
        MESH_TRY(hipStreamSynchronize(s));
      }
      MESH_TRY(hipGetLastError());
      if (dbg) fprintf(stderr, "[mrhash_hip] mesh post-process: %u soup vertices -> %zu vertices, %zu faces | kernels%s %.2f ms, second copy (buffers grown) %.2f (%.1f MB)\n",
                       n, nv, nf, speculative ? " + copy to the host" : "", t1 - t0, now() - t1, (nv * 48 + nf * 12) / 1e6);
    } else {
      // ---- fp32 over the link, widened by the host as the chunks land (k_stage_out).  Speculative like the above: into the
      // staging of the previous extraction, and again if that turns out too small.
      struct Drain { hipStream_t s; };
      Drain drain{s};
      auto drained = [](void* a) { return hipStreamQuery(((Drain*) a)->s) != hipErrorNotReady; };
      auto stage_ready = [&] { return c->V32.dev && c->C32.dev && c->F.dev && c->stage_ctl.dev; };
      bool clear_pending = true;  // the first launch of this extraction also clears the index tables
      auto launch = [&](u32& epoch_out) {
        StageOut a;
        a.src[0] = (const uint4*) dV; a.src[1] = (const uint4*) dC; a.src[2] = (const uint4*) dF;
        a.dst[0] = (uint4*) c->V32.dev; a.dst[1] = (uint4*) c->C32.dev; a.dst[2] = (uint4*) c->F.dev;
        a.totals = d_totals;
        a.cap_v = std::min(c->V32.cap, c->C32.cap) / 3; a.cap_f = c->F.cap / 3;
        // one flag per chunk the staging can hold
        const u32 max_chunks = (u32) ((a.cap_v * 12 + 15) / 16 / kStageWords + 1);
        const size_t room = (c->stage_ctl.cap - kStageHdrWords) / 2;
        if (max_chunks > room) a.cap_v = (u64) (room > 1 ? (room - 1) : 0) * kStageChunk / 12;
        a.hdr = (u64*) c->stage_ctl.dev;
        a.flags = c->stage_ctl.dev + kStageHdrWords;
        a.flag_stride = (u32) room;
        if (++c->stage_epoch == 0) c->stage_epoch = 1;
        a.epoch = epoch_out = c->stage_epoch;
        static const int grid = getenv("MRH_STAGE_WGS") ? std::max(1, atoi(getenv("MRH_STAGE_WGS"))) : 64;
        a.copy_wgs = (u32) grid;
        a.clear = (u32*) m.base;
        a.clear_words = clear_pending ? clear_words : 0;
        k_stage_out<<<grid + (clear_pending ? 256 : 0), 256, 0, s>>>(a);
        if (clear_pending) { c->mesh_clean_base = m.base; c->mesh_clean_words = clear_words; }
        clear_pending = false;
        return a;
      };
      // waits for the header of launch `epoch`; false: the stream ran dry without it (a failed launch)
      auto wait_hdr = [&](const u32 epoch) {
        const volatile u64* hdr = (const volatile u64*) c->stage_ctl.data();
        for (u32 spins = 1;; spins++) {
          if (hdr[2] == (u64) epoch) break;
          MRH_CPU_RELAX();
          if ((spins & 1023u) == 0 && drained(&drain)) {  // dry: the header is there, or about to be — or the launch failed
            const auto t = std::chrono::steady_clock::now();
            while (hdr[2] != (u64) epoch && std::chrono::steady_clock::now() - t < std::chrono::milliseconds(200)) MRH_CPU_RELAX();
            if (hdr[2] == (u64) epoch) break;
            return false;
          }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return true;
      };
      auto widen = [&](const u32 epoch, const size_t nv, const bool flagged) {
        double* const dst[2] = {c->V.data(), c->C.data()};
        const float* const src[2] = {c->V32.data(), c->C32.data()};
        const volatile u32* f0 = (const volatile u32*) c->stage_ctl.data() + kStageHdrWords;
        const volatile u32* const flags[2] = {flagged ? f0 : nullptr, flagged ? f0 + (c->stage_ctl.cap - kStageHdrWords) / 2 : nullptr};
        return widen_from_staging(dst, src, flags, epoch, nv * 3, drained, &drain);
      };
      bool done_ok = false;
      size_t nv = 0, nf = 0;
      double t1 = t0;
      if (stage_ready() && std::min(c->V32.cap, c->C32.cap) >= 3 && c->F.cap >= 3) {
        u32 epoch = 0;
        const StageOut a = launch(epoch);
        MESH_TRY(hipGetLastError());
        MESH_TRY(hipMemcpyAsync(c->h_mc + 2, d_totals, 2 * sizeof(u64), hipMemcpyDeviceToHost, s));
        if (!wait_hdr(epoch)) { MESH_TRY(hipStreamSynchronize(s)); MESH_TRY(hipGetLastError()); rc = fail(c, MRH_ERR_DEVICE, "mesh read-back: the staging kernel did not report"); goto done; }
        const volatile u64* hdr = (const volatile u64*) c->stage_ctl.data();
        nv = (size_t) hdr[0]; nf = (size_t) hdr[1];
        if (nv <= a.cap_v && nf <= a.cap_f) {
          c->V.resize_discard(nv * 3); c->C.resize_discard(nv * 3);
          c->F.n = nf * 3;
          const bool ok = widen(epoch, nv, true);
          MESH_TRY(hipStreamSynchronize(s));  // the faces, and the end of the launch
          MESH_TRY(hipGetLastError());
          if (!ok) { rc = fail(c, MRH_ERR_DEVICE, "mesh read-back: a staged chunk never arrived"); goto done; }
          done_ok = true;
        } else {
          MESH_TRY(hipStreamSynchronize(s));
        }
        t1 = now();
      } else {
        MESH_TRY(hipMemcpyAsync(c->h_mc + 2, d_totals, 2 * sizeof(u64), hipMemcpyDeviceToHost, s));
        MESH_TRY(hipStreamSynchronize(s));
        nv = (size_t) c->h_mc[2]; nf = (size_t) c->h_mc[3];
        t1 = now();
      }
      if (!done_ok) {  // first extraction, or the mesh outgrew the buffers: size them and go again
        c->V32.resize_discard(nv * 3 + 4); c->C32.resize_discard(nv * 3 + 4);
        c->F.resize_discard(std::max<size_t>(nf, 1) * 3 + 4);
        c->F.n = nf * 3;
        c->stage_ctl.resize_discard(kStageHdrWords + 2 * ((std::min(c->V32.cap, c->C32.cap) * 4 + kStageChunk - 1) / kStageChunk + 2));
        c->V.resize_discard(nv * 3); c->C.resize_discard(nv * 3);
        if (stage_ready()) {
          u32 epoch = 0;
          const StageOut a = launch(epoch);
          MESH_TRY(hipGetLastError());
          bool ok = wait_hdr(epoch) && nv <= a.cap_v && nf <= a.cap_f;
          if (ok) ok = widen(epoch, nv, true);
          MESH_TRY(hipStreamSynchronize(s));
          MESH_TRY(hipGetLastError());
          if (!ok) { rc = fail(c, MRH_ERR_DEVICE, "mesh read-back: the staged copy did not complete"); goto done; }
        } else {  // registration refused: plain copies, then the widening
          MESH_TRY(hipMemcpyAsync(c->V32.data(), dV, nv * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
          MESH_TRY(hipMemcpyAsync(c->C32.data(), dC, nv * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
          if (nf) MESH_TRY(hipMemcpyAsync(c->F.data(), dF, nf * 3 * sizeof(int), hipMemcpyDeviceToHost, s));
          MESH_TRY(hipStreamSynchronize(s));
          (void) widen(0, nv, false);
        }
      }
      MESH_TRY(hipGetLastError());
      if (dbg) fprintf(stderr, "[mrhash_hip] mesh post-process: %u soup vertices -> %zu vertices, %zu faces | kernels + fp32 staging + widening %.2f ms, second pass (buffers grown) %.2f (%.1f MB over the link)\n",
                       n, nv, nf, t1 - t0, now() - t1, (nv * 24 + nf * 12) / 1e6);
    }
  }
done:
#undef MESH_TRY
  return rc;
}



// the soup buffer: grow-only, owned by the context, valid until the next extraction
int ensure_soup(mrh_ctx* c, size_t n) {
  if (n > c->soup_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_soup) HIP_TRY(c, hipFree(c->d_soup));
    c->d_soup = nullptr; c->soup_cap = 0;
    const size_t cap = n + n / 8;
    HIP_TRY(c, hipMalloc((void**) &c->d_soup, cap * sizeof(mrh_triangle) + 16));  // + 16: k_copy_out reads whole 16-byte words
    c->soup_cap = cap;
  }
  return MRH_OK;
}

int ensure_zbuf(mrh_ctx* c, size_t npix) {
  if (c->zbuf_n < npix) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_zbuf) HIP_TRY(c, hipFree(c->d_zbuf));
    c->d_zbuf = nullptr;
    HIP_TRY(c, hipMalloc((void**) &c->d_zbuf, 2 * npix * sizeof(u64)));
    c->zbuf_n = npix;
  }
  return MRH_OK;
}

// A starve frame of a single-resolution, unsharded map on the two-launch path: behind the frame's k_back<FREE = false>, the two
// min-passes and the tail (pass 2 + summaries + garbage collection + the other z-buffer pair cleared) — three launches on the main
// stream, nothing of the pipeline flushed.  lz: 2 = a pipelined frame (collected blocks become zombies), 0 = a serial frame.
int launch_starve_fused(mrh_ctx* c, const Cam& k, const Fast& f, const Lists& L, const int set, const float thr, const u32 stamp, const int lz) {
  const size_t npix = (size_t) k.rows * k.cols;
  hipStream_t s = c->stream;
  if (c->zfused_n < npix) {
    HIP_TRY(c, hipStreamSynchronize(s));
    if (c->d_zfused) HIP_TRY(c, hipFree(c->d_zfused));
    c->d_zfused = nullptr; c->zfused_n = 0;
    HIP_TRY(c, hipMalloc((void**) &c->d_zfused, 4 * npix * sizeof(u64)));
    c->zfused_n = npix;
    c->zfused_clean[0] = c->zfused_clean[1] = false;
  }
  if (c->zfused_clean_npix != npix) c->zfused_clean[0] = c->zfused_clean[1] = false;  // the camera changed size since the pairs were cleared
  c->zfused_clean_npix = npix;
  const int p = c->zfused_next, q = p ^ 1;
  u64* z0 = c->d_zfused + (size_t) p * 2 * c->zfused_n;
  u64* z1 = z0 + npix;
  u64* other = c->d_zfused + (size_t) q * 2 * c->zfused_n;
  // "empty" = INT64_MAX: above every key (depth bits of a finite positive float < 0x7F800000)
  if (!c->zfused_clean[p]) k_fill_u64<<<256, 256, 0, s>>>(z0, 2 * npix, 0x7FFFFFFFFFFFFFFFull);
  c->zfused_clean[p] = false;
  const int grid = 2048;  // x 4 waves, one block each per round
  if (k.model) {
    k_starve_z<0, true><<<grid, 256, 0, s>>>(k, c->map, c->tab, f, L.vis, set, z0, z1);
    k_starve_z<1, true><<<grid, 256, 0, s>>>(k, c->map, c->tab, f, L.vis, set, z0, z1);
    if (lz == 2) k_starve_tail<2, true><<<grid, 256, 0, s>>>(k, c->map, c->tab, f, L, set, thr, stamp, z0, z1, other, 2 * npix);
    else k_starve_tail<0, true><<<grid, 256, 0, s>>>(k, c->map, c->tab, f, L, set, thr, stamp, z0, z1, other, 2 * npix);
  } else {
    k_starve_z<0, false><<<grid, 256, 0, s>>>(k, c->map, c->tab, f, L.vis, set, z0, z1);
    k_starve_z<1, false><<<grid, 256, 0, s>>>(k, c->map, c->tab, f, L.vis, set, z0, z1);
    if (lz == 2) k_starve_tail<2, false><<<grid, 256, 0, s>>>(k, c->map, c->tab, f, L, set, thr, stamp, z0, z1, other, 2 * npix);
    else k_starve_tail<0, false><<<grid, 256, 0, s>>>(k, c->map, c->tab, f, L, set, thr, stamp, z0, z1, other, 2 * npix);
  }
  HIP_TRY(c, hipGetLastError());
  c->zfused_clean[q] = true;
  c->zfused_next = q;
  c->n_starve_fused++;
  return MRH_OK;
}
// may this frame's starve step take the three fused launches?  (tile-sharded maps reduce the z-buffers over the ranks between the
// passes, multi-resolution and general frames walk lists of another kind: they keep k_starve<0,1,2>)
bool starve_fused_ok(const mrh_ctx* c) { return c->starve_fused && c->p.shard_count <= 1 && !c->tab.multi_res && !c->frame_general; }

// one of the three starve passes over the current compact (fast path: visible) list
int launch_starve(mrh_ctx* c, int pass) {
  const Cam& k = c->cam;
  const size_t npix = (size_t) k.rows * k.cols;
  hipStream_t s = c->stream;
  if (pass == 0) {
    int rc = ensure_zbuf(c, npix);
    if (rc) return rc;
    // "empty" = INT64_MAX: above every key (depth bits of a finite positive float < 0x7F800000) in both the
    // unsigned and the signed reading, so shards can be min-reduced as int64
    k_fill_u64<<<256, 256, 0, s>>>(c->d_zbuf, 2 * npix, 0x7FFFFFFFFFFFFFFFull);
    k_starve<0><<<c->integrate_grid, 512, 0, s>>>(k, c->map, c->tab, c->d_zbuf, c->d_zbuf + npix);
  } else if (pass == 1) {
    k_starve<1><<<c->integrate_grid, 512, 0, s>>>(k, c->map, c->tab, c->d_zbuf, c->d_zbuf + npix);
  } else {
    k_starve<2><<<c->integrate_grid, 512, 0, s>>>(k, c->map, c->tab, c->d_zbuf, c->d_zbuf + npix);
  }
  return MRH_OK;
}

// everything of a frame that follows the starve step
int frame_tail(mrh_ctx* c, bool starved, int max_num_frames) {
  hipStream_t s = c->stream;
  const Cam& k = c->cam;
  const Map& m = c->map;
  const Tab& t = c->tab;
  const float thr = m.trunc + m.trunc_scale * k.max_depth;  // getTruncation(camera.maxDepth(), ...), vds.cu:1720
  if (c->frame_general) {  // garbageCollectIdentify + garbageCollectFree over the compact list (vds.cu:1674-1713, :1827-1844)
    if (max_num_frames > 0) {
      k_gc_identify<<<c->integrate_grid, 512, 0, s>>>(t, thr, c->d_decision);
      if (c->profile) k_gc_free<true><<<256, 256, 0, s>>>(t, c->d_decision);
      else k_gc_free<false><<<256, 256, 0, s>>>(t, c->d_decision);
    }
    if (!t.multi_res) c->fast_summaries_stale = true;  // the general kernels do not maintain the fast path's GC summaries
  } else if (!t.multi_res) {
    if (starved) k_summarize_visible<<<1024, 256, 0, s>>>(t, c->fast);  // weights changed: the GC summaries follow the payload
    if (max_num_frames > 0 && !c->frame_gc_inline) {
      const Lists L = {t.compact, c->fast.bbox, c->d_cfree, c->d_zmin, (u32) c->num_blocks};
      k_free_lists<<<256, 256, 0, s>>>(t, c->fast, L, c->frame_parity, thr);
    }
  }
  c->frames++;
  HIP_TRY(c, hipGetLastError());
  return MRH_OK;
}

// starve (voxel_data_structures.cpp:139) + the rest of the frame; sharded contexts stop for the host's min-reduction
int starve_and_tail(mrh_ctx* c, int max_num_frames) {
  const bool starve = max_num_frames > 0 && c->frames > 0 && c->frames % (uint64_t) max_num_frames == 0;
  if (starve) {
    int rc = launch_starve(c, 0);
    if (rc) return rc;
    if (c->p.shard_count > 1 && c->comm) {
      // a communicator is attached: the two min-reductions over the shards run on this stream, between the passes —
      // ncclAllReduce(int64, MIN) over xGMI, no host synchronisation, the frame stays one enqueue
      const size_t npix = (size_t) c->cam.rows * c->cam.cols;
      rc = comm_allreduce_zbuf(c, c->d_zbuf, npix);
      if (rc) return rc;
      if ((rc = launch_starve(c, 1))) return rc;
      rc = comm_allreduce_zbuf(c, c->d_zbuf + npix, npix);
      if (rc) return rc;
      if ((rc = launch_starve(c, 2))) return rc;
      return frame_tail(c, starve, max_num_frames);
    }
    if (c->p.shard_count > 1) {
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      c->pending = 1;
      c->pending_max_frames = max_num_frames;
      return MRH_PENDING_EXCHANGE;
    }
    if ((rc = launch_starve(c, 1))) return rc;
    if ((rc = launch_starve(c, 2))) return rc;
  }
  return frame_tail(c, starve, max_num_frames);
}

}  // namespace

extern "C" {

#define MRH_STR2(x) #x
#define MRH_STR(x) MRH_STR2(x)
const char* mrh_version(void) { return "mrhash_hip abi" MRH_STR(MRH_ABI_VERSION) " gfx950 hand-written-hip"; }
#ifdef MRH_MC_TRACE
// tuning builds only (tools/trace_mc.sh): read (and optionally clear) the phase accumulators of k_mc
int mrh_debug_mc_trace(uint32_t* out, int clear) {  // out: 2 x 65536 x 8 words
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(d_mc_trace), sizeof(d_mc_trace)) != hipSuccess) return MRH_ERR_DEVICE;
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(d_mc_trace)) != hipSuccess || hipMemset(p, 0, sizeof(d_mc_trace)) != hipSuccess) return MRH_ERR_DEVICE;
  }
  return MRH_OK;
}
#endif

#ifdef MRH_SCAN_TRACE
// tuning builds only (tools/trace_scan.sh): read (and optionally clear) the phase stamps of the scan kernels
int mrh_debug_scan_trace(unsigned long long* out, int clear) {  // out: 4 x kScanTraceWgs x 8 words
  if (hipDeviceSynchronize() != hipSuccess) return MRH_ERR_DEVICE;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(d_scan_trace), sizeof(d_scan_trace)) != hipSuccess) return MRH_ERR_DEVICE;
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(d_scan_trace)) != hipSuccess || hipMemset(p, 0, sizeof(d_scan_trace)) != hipSuccess) return MRH_ERR_DEVICE;
  }
  return MRH_OK;
}
#endif

const char* mrh_last_error(const mrh_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int mrh_create(const mrh_params* p, mrh_ctx** out) {
  if (!p || !out) return fail(nullptr, MRH_ERR_INVALID_ARG, "mrh_create: null argument");
  if (p->abi_version != MRH_ABI_VERSION) return fail(nullptr, MRH_ERR_INVALID_ARG, "mrh_create: abi_version mismatch");
  if (!(p->virtual_voxel_size > 0.f)) return fail(nullptr, MRH_ERR_INVALID_ARG, "mrh_create: virtual_voxel_size must be > 0");
  if (!(p->sdf_truncation >= 0.f) || !(p->sdf_truncation_scale >= 0.f))
    return fail(nullptr, MRH_ERR_INVALID_ARG, "mrh_create: sdf_truncation and sdf_truncation_scale must be >= 0");
  if (p->voxel_extents_scale != 0 && p->voxel_extents_scale != 1)
    return fail(nullptr, MRH_ERR_UNSUPPORTED, "mrh_create: voxel_extents_scale != 1 is incoherent in the reference (vhu.cuh:90-92 vs 138-140)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, MRH_ERR_NO_DEVICE, "mrh_create: no HIP device visible");
  if (p->device_id < 0 || p->device_id >= ndev) return fail(nullptr, MRH_ERR_INVALID_ARG, "mrh_create: device_id %d out of range (%d devices)", p->device_id, ndev);
  if (p->shard_count > 1 && (p->shard_rank < 0 || p->shard_rank >= p->shard_count))
    return fail(nullptr, MRH_ERR_INVALID_ARG, "mrh_create: shard_rank out of range");

  mrh_ctx* c = new mrh_ctx();
  c->p = *p;
  if (c->p.integration_weight_max == 0) c->p.integration_weight_max = 255;
  if (c->p.voxel_extents_scale == 0) c->p.voxel_extents_scale = 1;
  if (c->p.shard_count < 1) c->p.shard_count = 1;
  c->device = p->device_id;
  memset(&c->tab, 0, sizeof c->tab);
  memset(&c->cam, 0, sizeof c->cam);
#define CREATE_TRY(expr)                                                                                         \
  do {                                                                                                           \
    hipError_t e__ = (expr);                                                                                     \
    if (e__ != hipSuccess) {                                                                                     \
      fail(nullptr, MRH_ERR_DEVICE, "mrh_create: %s failed: %s", #expr, hipGetErrorString(e__));                 \
      free_all(c);                                                                                               \
      delete c;                                                                                                  \
      return MRH_ERR_DEVICE;                                                                                     \
    }                                                                                                            \
  } while (0)
  CREATE_TRY(hipSetDevice(c->device));
  CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

  // capacities: geowrapper.cpp:37-54 when not given explicitly
  size_t free_b = 0, total_b = 0;
  CREATE_TRY(hipMemGetInfo(&free_b, &total_b));
  const double to_alloc = (double) free_b * 0.70;  // SDFBlocks_ratio
  c->num_blocks = p->num_sdf_blocks ? p->num_sdf_blocks : (uint64_t) ((to_alloc * 0.70) / (12.0 * 512.0));
  if (c->num_blocks >= (1ull << 28)) c->num_blocks = (1ull << 28) - 1;  // 31-bit coarse unit ids
  c->max_triangles = p->max_triangles ? p->max_triangles : (uint64_t) ((to_alloc * 0.25) / 72.0);
  c->slots = next_pow2(p->hash_slots ? p->hash_slots : 4 * c->num_blocks);
  if (c->slots < 1024) c->slots = 1024;
  c->low_blocks_to_allocate = (int) ((float) c->num_blocks * 0.1f);  // voxel_data_structures.cuh:57-61

  Tab& t = c->tab;
  t.slot_mask = (u32) (c->slots - 1);
  t.max_probe = 512;
  t.cap_blocks = (u32) c->num_blocks;
  t.multi_res = p->sdf_var_threshold > 0.f ? 1u : 0u;
  CREATE_TRY(hipMalloc((void**) &t.keys, c->slots * sizeof(u64)));
  CREATE_TRY(hipMalloc((void**) &t.vals, c->slots * sizeof(u32)));
  CREATE_TRY(hipMalloc((void**) &t.heap_fine, (c->num_blocks + 1) * sizeof(u32)));
  CREATE_TRY(hipMalloc((void**) &t.desc_fine, c->num_blocks * sizeof(int4)));
  if (t.multi_res) {
    CREATE_TRY(hipMalloc((void**) &t.heap_coarse, (c->num_blocks * 8 + 9) * sizeof(u32)));
    CREATE_TRY(hipMalloc((void**) &t.desc_coarse, c->num_blocks * 8 * sizeof(int4)));
    CREATE_TRY(hipMalloc((void**) &c->d_realloc, c->num_blocks * sizeof(int4)));
    CREATE_TRY(hipMalloc((void**) &c->d_reint, c->num_blocks * sizeof(int4)));
  }
  CREATE_TRY(hipMalloc((void**) &t.pool, c->num_blocks * (size_t) kFineBytes));
  CREATE_TRY(hipMalloc((void**) &t.compact, c->num_blocks * (t.multi_res ? 9 : 1) * sizeof(int4)));
  CREATE_TRY(hipMalloc((void**) &c->d_decision, c->num_blocks * (t.multi_res ? 9 : 1) * sizeof(u32)));
  CREATE_TRY(hipMalloc((void**) &t.ctr, CTR_COUNT * sizeof(int)));
  CREATE_TRY(hipMalloc((void**) &t.prof, PROF_COUNT * sizeof(u64)));
  CREATE_TRY(hipMalloc((void**) &c->d_flag, sizeof(int)));
  CREATE_TRY(hipMalloc((void**) &c->d_misc, 4 * sizeof(u32)));
  CREATE_TRY(hipMalloc((void**) &c->d_upd_partials, (size_t) c->integrate_grid * sizeof(u64)));
  CREATE_TRY(hipMalloc((void**) &c->d_cnt_partials, (size_t) 32768 * 4 * sizeof(u64)));  // max MRH_FUSED_GRID
  memset(&c->fast, 0, sizeof c->fast);
  CREATE_TRY(hipMalloc((void**) &c->fast.summary, c->num_blocks * sizeof(uint2)));
  c->fast.zlist_cap = (u32) c->num_blocks;
  if (const char* g = getenv("MRH_ZLIST_CAP")) { const int v = atoi(g); if (v > 0 && (uint64_t) v < c->num_blocks) c->fast.zlist_cap = (u32) v; }
  const size_t list_cap = c->num_blocks * (t.multi_res ? 9 : 1);  // visible / free lists may hold coarse units, too
  CREATE_TRY(hipMalloc((void**) &c->fast.bbox, list_cap * sizeof(int4)));
  if (t.multi_res) CREATE_TRY(hipMalloc((void**) &c->fast.summary_c, c->num_blocks * 8 * sizeof(uint2)));
#ifdef MRH_TRACE
  CREATE_TRY(hipMalloc((void**) &c->fast.trace, c->num_blocks * 8 * sizeof(u64)));
  CREATE_TRY(hipMemset(c->fast.trace, 0, c->num_blocks * 8 * sizeof(u64)));
#endif
  CREATE_TRY(hipMalloc((void**) &c->d_cfree, list_cap * sizeof(int4)));
  CREATE_TRY(hipMalloc((void**) &c->d_zmin, list_cap * sizeof(float)));
#undef CREATE_TRY

  Map& m = c->map;
  m.vs = p->virtual_voxel_size;
  m.trunc = p->sdf_truncation;
  m.trunc_scale = p->sdf_truncation_scale;
  m.var_threshold = p->sdf_var_threshold;
  m.mc_threshold = p->marching_cubes_threshold;
  m.weight_sample = p->integration_weight_sample & 0xFF;
  m.weight_max = c->p.integration_weight_max & 0xFF;
  m.min_weight_threshold = p->min_weight_threshold;
  m.shard_rank = c->p.shard_rank;
  m.shard_count = c->p.shard_count;
  m.shard_chunk_log2 = (p->shard_chunk_log2 > 0 && p->shard_chunk_log2 < 16) ? p->shard_chunk_log2 : 3;
  {  // where is voxel -> block an arithmetic shift?  (mrh_device.h: world_to_block_fast)
    const u32 init = 1u << 23;
    u32 first_bad = 0;
    if (hipMemcpy(c->d_misc, &init, sizeof init, hipMemcpyHostToDevice) != hipSuccess) first_bad = 1;
    k_block_shift_limit<<<(1 << 23) / 256, 256, 0, c->stream>>>(m.vs, c->d_misc);
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&first_bad, c->d_misc, sizeof first_bad, hipMemcpyDeviceToHost) != hipSuccess) first_bad = 1;
    int lim = 1;
    while ((u32) (lim << 1) <= first_bad && lim < (1 << 22)) lim <<= 1;  // largest power of two <= first mismatch
    m.block_shift_limit = first_bad <= 1 ? 0 : lim;
    if (getenv("MRH_DEBUG")) fprintf(stderr, "[mrhash_hip] voxel->block is a shift for |v| < %d (first mismatch at %u, voxel size %g)\n", m.block_shift_limit, first_bad, (double) m.vs);
  }

  {  // correctly rounded reciprocals for the short divisions of the running mean (mrh_device.h: div_cr)
    auto rn_reciprocal = [](float b) {  // fp64 quotient, then the nearest of the three neighbouring floats (b * c is exact in fp64)
      const float c0 = (float) (1.0 / (double) b);
      float best = c0;
      double err = std::fabs(1.0 - (double) c0 * (double) b);
      for (float t : {std::nextafter(c0, 0.f), std::nextafter(c0, INFINITY)}) {
        const double e = std::fabs(1.0 - (double) t * (double) b);
        if (e < err) { err = e; best = t; }
      }
      return best;
    };
    // weight sums: the kernels use v_rcp_f32 + one Newton step; for the integers 1 .. 510 that must be RN(1 / w)
    std::vector<float> dev(kRcpWeightEntries, 0.f);
    bool ok = hipMalloc((void**) &c->d_rcp_w, dev.size() * sizeof(float)) == hipSuccess;
    if (ok) {
      k_rcp_weights<<<(kRcpWeightEntries + 255) / 256, 256, 0, c->stream>>>(c->d_rcp_w);
      ok = hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(dev.data(), c->d_rcp_w, dev.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
    }
    int w_bad = ok ? 0 : 1;
    for (int w = 1; ok && w <= 510; w++) w_bad += dev[w] != rn_reciprocal((float) w);
    m.wsum_two_steps = w_bad ? 1 : 0;
    const float half_vs = m.vs / 2;
    m.r_half_vs = rn_reciprocal(half_vs);
    u32 bad = 1;
    if (hipMemset(c->d_misc, 0, sizeof(u32)) == hipSuccess) {
      k_check_div_cr<<<4096, 256, 0, c->stream>>>(half_vs, m.r_half_vs, c->d_misc);
      if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&bad, c->d_misc, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
    }
    m.half_vs_two_steps = bad ? 1 : 0;
    if (const char* g = getenv("MRH_SAFE_DIV")) { if (atoi(g)) m.half_vs_two_steps = 1; }  // force the fallback instantiation (tests)
    if (getenv("MRH_DEBUG")) fprintf(stderr, "[mrhash_hip] division by vs / 2 with one residual step: %u mismatches over the working range -> %s; refined reciprocals of the weight sums: %d not correctly rounded\n", bad, bad ? "two steps" : "one step", w_bad);
  }

  if (const char* g = getenv("MRH_FUSED_GRID")) {  // tuning knob: workgroups (x4 waves) of the fused integrate kernel
    const int v = atoi(g);
    if (v > 0 && v <= 32768) c->fused_grid = v;
  }
  if (const char* g = getenv("MRH_DEFER_UPLOADS")) c->defer_uploads = atoi(g) ? 1 : 0;
  if (const char* g = getenv("MRH_PIPE")) c->pipe = atoi(g) ? 1 : 0;
  if (const char* g = getenv("MRH_STARVE_FUSED")) c->starve_fused = atoi(g) != 0;
  if (const char* g = getenv("MRH_PREWARM")) c->prewarm_on = atoi(g) != 0;
  if (const char* g = getenv("MRH_PIPE_GRID")) { const int v = atoi(g); if (v > 0 && v <= 32768) c->pipe_grid = v; }
  if (const char* g = getenv("MRH_PIPE_DEFER")) { const int v = atoi(g); if (v >= 1 && v < mrh_ctx::kPendMax) c->pipe_defer = v; }
  if (const char* g = getenv("MRH_PIPE_UPLOADS")) c->pipe_uploads = atoi(g) ? 1 : 0;
  if (const char* g = getenv("MRH_PIPE_PERIOD")) { const int v = atoi(g); if (v > 0) c->pipe_period = v; }
  if (const char* g = getenv("MRH_SWEEP_WGS")) { const int v = atoi(g); if (v > 0 && v <= 4096) c->sweep_wgs = v; }
  if (const char* g = getenv("MRH_MESH_HOST")) c->mesh_on_host = atoi(g) ? 1 : 0;
  if (const char* g = getenv("MRH_MESH_F64_LINK")) c->f64_link = atoi(g) != 0;
  c->V.pin = c->C.pin = c->f64_link;  // fp32 link: the doubles are written by the host only
  if (const char* g = getenv("MRH_QTREE_LITERAL")) c->qt_literal = atoi(g) ? 1 : 0;
  if (const char* g = getenv("MRH_MR_FUSED")) c->mr_fused = atoi(g) ? 1 : 0;
  if (const char* g = getenv("MRH_SCAN_ROW_LEN")) c->scan_layout_hint = atoi(g);
  if (const char* g = getenv("MRH_SCAN_PATCH_LOG2")) { const int v = atoi(g); if (v >= 0 && v <= 8) c->scan_patch_log2 = v; }
  if (const char* g = getenv("MRH_LIDAR_BUCKETS")) c->lidar_buckets = atoi(g) ? 1 : 0;
  if (const char* g = getenv("MRH_SCAN_SEQ_START")) c->scan2_seq = (u32) strtoul(g, nullptr, 0);  // tests: scans next to the wrap of the block stamps
  if (const char* g = getenv("MRH_REHASH_PERIOD")) { const int v = atoi(g); if (v > 0) c->census_period = v; }
  if (const char* g = getenv("MRH_REHASH_FORCE")) c->census_force = atoi(g) ? 1 : 0;
  if (const char* g = getenv("MRH_REHASH_OFF")) { if (atoi(g)) c->census_period = -1; }  // no upkeep at all (tests: shows what it prevents)
  if (const char* g = getenv("MRH_SWEEP_WGS_MR")) { const int v = atoi(g); if (v > 0 && v <= 4096) c->sweep_wgs_mr = v; }
  int rc = init_buffers(c);
  if (rc != MRH_OK) {
    g_create_err = c->err;
    free_all(c);
    delete c;
    return rc;
  }
  // identity pose; camera must be set by the caller (geowrapper.cpp:80 installs a 1x1 placeholder)
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const float z[3] = {0, 0, 0};
  mrh_set_pose(c, I, z);
  c->cam.min_depth = p->min_depth;
  c->cam.max_depth = p->max_depth;
  *out = c;
  return MRH_OK;
}

int mrh_destroy(mrh_ctx* c) {
  if (!c) return MRH_OK;
  if (getenv("MRH_DEBUG") && c->dbg_lazy_frames)
    fprintf(stderr, "[mrhash_hip] pipelined frames %llu: host waited %.1f us per frame for the ring, spent %.1f us per frame in the launch calls, %llu cross-stream waits\n",
            (unsigned long long) c->dbg_lazy_frames, c->dbg_spin_us / c->dbg_lazy_frames, c->dbg_api_us / c->dbg_lazy_frames, (unsigned long long) c->dbg_waits);
  if (getenv("MRH_DEBUG") && c->d_scan_ctr && c->scan2_seq) {  // the last scan's counters (mrh_scan.h)
    u32 h[2 * SC_N] = {0};
    (void) hipStreamSynchronize(c->stream);
    (void) hipMemcpy(h, c->d_scan_ctr, sizeof(h), hipMemcpyDeviceToHost);
    const u32* k = h + (c->scan2_seq & 1u) * SC_N;
    fprintf(stderr, "[mrhash_hip] last scan: %u records, %u chunks + %u runs beyond a wave\n", k[SC_PLACED], k[SC_CHUNKS], k[SC_BIG]);
  }
#ifdef MRH_TRACE
  if (const char* path = getenv("MRH_TRACE_FILE")) {  // tuning builds: phase timestamps of the last k_back launch
    if (c->fast.trace) {
      hipDeviceSynchronize();
      const size_t n = std::min<size_t>(c->num_blocks, 40960) * 8;
      std::vector<u64> h(n);
      hipMemcpy(h.data(), c->fast.trace, n * sizeof(u64), hipMemcpyDeviceToHost);
      if (FILE* fp = fopen(path, "wb")) { fwrite(h.data(), sizeof(u64), n, fp); fclose(fp); }
    }
  }
#endif
  if (c->deferred.on) { (void) hipSetDevice(c->device); (void) flush_deferred(c); }  // the frame mrh_integrate accepted last
  widen_quiesce();  // the result arrays are about to be unmapped
  if (getenv("MRH_DEBUG") || getenv("MRH_WIDEN_REPORT"))
    fprintf(stderr, "[mrhash_hip] widening: %llu chunks redone by the calling thread (their helper had not finished 40 us after the chunk landed)\n", (unsigned long long) widen_redone());
  free_all(c);
  delete c;
  return MRH_OK;
}

int mrh_reset(mrh_ctx* c) {
  int rc = ensure_ready(c, "mrh_reset");
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  rc = drain_events(c);
  if (rc) return rc;
  return init_buffers(c);
}

int mrh_set_camera(mrh_ctx* c, float fx, float fy, float cx, float cy, int rows, int cols, float min_depth, float max_depth, int model) {
  if (!c) return MRH_ERR_INVALID_ARG;
  if (rows <= 0 || cols <= 0 || (model != MRH_CAMERA_PINHOLE && model != MRH_CAMERA_SPHERICAL))
    return fail(c, MRH_ERR_INVALID_ARG, "mrh_set_camera: bad rows/cols/model");
  if (c->deferred.on) {  // a frame kept back by mrh_integrate was issued under the old camera
    const int frc = ensure_device(c, "mrh_set_camera");
    if (frc) return frc;
    const int drc = flush_deferred(c);
    if (drc < 0) return drc;
  }
  Cam& k = c->cam;
  // camera.cuh:19-34
  k.fx = fx; k.fy = fy; k.ifx = 1.f / fx; k.ify = 1.f / fy; k.cx = cx; k.cy = cy;
  k.rows = rows; k.cols = cols;
  k.row_thr = (int) ((float) (unsigned) rows * 0.5f);
  k.col_thr = (int) ((float) (unsigned) cols * 0.5f);
  k.min_depth = min_depth; k.max_depth = max_depth;
  k.max_int_dist = max_depth;  // geowrapper.cpp:111 setIntegrationDistance(max_depth)
  c->spherical = model == MRH_CAMERA_SPHERICAL;
  k.model = c->spherical ? 1 : 0;
  c->has_camera = true;
  return MRH_OK;
}

int mrh_set_pose(mrh_ctx* c, const float R[9], const float t[3]) {
  if (!c || !R || !t) return MRH_ERR_INVALID_ARG;
  Cam& k = c->cam;
  memcpy(k.R, R, 36);
  memcpy(k.t, t, 12);
  // cuda_algebra.cuh:45-57, 137-143 (the reference recomputes this per thread on the device)
  k.Ri[0] = R[0]; k.Ri[1] = R[3]; k.Ri[2] = R[6];
  k.Ri[3] = R[1]; k.Ri[4] = R[4]; k.Ri[5] = R[7];
  k.Ri[6] = R[2]; k.Ri[7] = R[5]; k.Ri[8] = R[8];
  // evaluated with separate products and sums (no FMA): this TU is built with -ffp-contract=off
  const float x = k.Ri[0] * t[0] + k.Ri[1] * t[1] + k.Ri[2] * t[2];
  const float y = k.Ri[3] * t[0] + k.Ri[4] * t[1] + k.Ri[5] * t[2];
  const float z = k.Ri[6] * t[0] + k.Ri[7] * t[1] + k.Ri[8] * t[2];
  k.ti[0] = -x; k.ti[1] = -y; k.ti[2] = -z;
  return MRH_OK;
}

namespace {

// Host copy into pinned staging with non-temporal stores: the destination is read next by the DMA engine, not by this
// core, so write-allocating it through the cache only costs bandwidth (tools/micro/staging_copy.hip: 1.2 MB in 28.5 us
// vs 40.9 us with memcpy, cold pageable source).
#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) void copy_streaming_avx2(void* dst, const void* src, size_t n) {
  const __m256i* s = (const __m256i*) src;
  __m256i* d = (__m256i*) dst;  // pinned allocations are page-aligned
  const size_t v = n / 32;
  for (size_t i = 0; i < v; i++) _mm256_stream_si256(d + i, _mm256_loadu_si256(s + i));
  _mm_sfence();
  if (n & 31) memcpy((char*) dst + v * 32, (const char*) src + v * 32, n & 31);
}
// floats -> doubles with non-temporal stores (the doubles are read by the caller later, not by this core)
__attribute__((target("avx2"))) void widen_floats_avx2(double* dst, const float* src, size_t n) {
  const size_t v = n / 8;
  for (size_t i = 0; i < v; i++) {
    const __m256 f = _mm256_loadu_ps(src + i * 8);
    _mm256_stream_pd(dst + i * 8, _mm256_cvtps_pd(_mm256_castps256_ps128(f)));
    _mm256_stream_pd(dst + i * 8 + 4, _mm256_cvtps_pd(_mm256_extractf128_ps(f, 1)));
  }
  _mm_sfence();
  for (size_t i = v * 8; i < n; i++) dst[i] = (double) src[i];
}
void widen_floats(double* dst, const float* src, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2 && ((uintptr_t) dst & 31) == 0) widen_floats_avx2(dst, src, n);
  else for (size_t i = 0; i < n; i++) dst[i] = (double) src[i];
}
void copy_chunk(void* dst, const void* src, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2 && n >= (64u << 10) && ((uintptr_t) dst & 31) == 0) copy_streaming_avx2(dst, src, n);
  else memcpy(dst, src, n);
}

// The setter's copy of a 640x480 frame (1.2 MB depth + 0.9 MB colour) is what bounds the host-input path: one core moves
// it at ~28 GB/s with streaming stores, 75 us per frame against 45 us of GPU work.  A small pool of helper threads shares
// every copy (128 KiB chunks handed out by an atomic counter; the calling thread works too).  The helpers spin for a short
// while after a job, so that in a frame loop the next upload finds them awake, and sleep on a condition variable
// otherwise.  One pool per process, started by the first large upload, MRH_COPY_THREADS=0 turns it off.
struct CopyPool {
  static constexpr size_t kChunk = 128u << 10;
  struct Job {
    std::atomic<char*> dst{nullptr}; std::atomic<const char*> src{nullptr}; std::atomic<size_t> bytes{0}, nchunks{0};
    // widening jobs (widen_from_staging): two parts of `bytes` bytes of floats each, chunk i < nchunks / 2 belongs to part 0;
    // a chunk is taken up when its flag word equals `epoch` (flags == nullptr: at once)
    std::atomic<int> widen{0};
    std::atomic<char*> dst2{nullptr}; std::atomic<const char*> src2{nullptr};
    std::atomic<const volatile uint32_t*> flags{nullptr}, flags2{nullptr};
    std::atomic<uint32_t> epoch{0};
  };
  static constexpr size_t kWidenChunk = 64u << 10;  // = kStageChunk: bytes of floats per flag
  static constexpr size_t kMaxStates = 1u << 16;    // chunks of one widening job that carry a state (beyond: the job waits for every helper)
  std::atomic<int> abort_widen{0};
  std::atomic<int64_t> spin_until_ns{0};  // helpers do not go to sleep before this time (widen_prewake)
  static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::thread> threads;
  std::atomic<uint64_t> generation{0};  // bumped once per job, after the job's tickets are out
  // Chunk tickets carry the job they belong to: (generation << 32) | next chunk.  A helper that saw generation g and was
  // descheduled can only ever claim a chunk of job g, and only while job g is unfinished (an unclaimed chunk of g exists):
  // it can neither consume a ticket of a later job nor count a chunk into its `done`.  The descriptor of job g lives in
  // jobs[g & 1], which is rewritten only by job g + 2, i.e. after g and g + 1 have both completed.
  std::atomic<uint64_t> ticket{0};
  std::atomic<size_t> done{0};
  std::atomic<int> sleepers{0};
  // A WIDENING job does not wait for its helpers (round 5): every chunk has a state {0 not done, 2 done}, and when the tickets
  // have run out the submitting thread REDOES whatever is not done after a short grace — the bytes are the same whoever writes
  // them ((double) (float) of pinned staging that nothing rewrites meanwhile) —, so a helper that claimed a chunk and then lost
  // its core costs the call one chunk of work instead of the scheduler's time slice (tools/stress_extract.py: tail of hundreds of
  // ms with the host oversubscribed).  Such a straggler may still be reading the staging and writing the doubles after the call
  // has returned: `inflight` counts the helpers between "about to claim" and "finished", and whoever is about to rewrite the
  // staging, release or regrow the arrays, or publish another job waits for it to reach zero first (quiesce()).
  // Upload jobs (copy()) keep waiting for every chunk: their source is the CALLER's buffer, which is free on return.
  std::atomic<int> inflight{0};
  std::unique_ptr<std::atomic<uint8_t>[]> state{new std::atomic<uint8_t>[kMaxStates]};
  std::atomic<uint64_t> redone{0};  // chunks the submitting thread redid (MRH_DEBUG / tools/stress_extract.py)
  Job jobs[2];
  bool started = false;

  void quiesce() {
    while (inflight.load(std::memory_order_seq_cst) != 0) MRH_CPU_RELAX();
  }
  // claims the next chunk of job g; false: none left (or the tickets belong to another job)
  bool claim(const uint64_t g, const Job& j, size_t& i) {
    uint64_t cur = ticket.load(std::memory_order_acquire);
    for (;;) {
      if ((cur >> 32) != (g & 0xFFFFFFFFull)) return false;  // another job's tickets: not ours to take
      i = (size_t) (cur & 0xFFFFFFFFull);
      if (i >= j.nchunks.load(std::memory_order_relaxed)) return false;
      if (ticket.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel, std::memory_order_acquire)) return true;
    }
  }
  void work(const uint64_t g) {  // helpers
    Job& j = jobs[g & 1];
    for (;;) {
      inflight.fetch_add(1, std::memory_order_seq_cst);  // BEFORE the claim: a submitter that sees zero knows nobody holds a chunk
      size_t i;
      if (!claim(g, j, i)) { inflight.fetch_sub(1, std::memory_order_seq_cst); break; }
      // chunk i of job g is ours: nobody rewrites the descriptor before `inflight` is back at zero
      if (j.widen.load(std::memory_order_relaxed)) {
        if (widen_chunk(j, i, nullptr, nullptr) && i < kMaxStates) state[i].store(2, std::memory_order_release);
      } else {
        const size_t off = i * kChunk, len = std::min(kChunk, j.bytes.load(std::memory_order_relaxed) - off);
        copy_chunk(j.dst.load(std::memory_order_relaxed) + off, j.src.load(std::memory_order_relaxed) + off, len);
      }
      done.fetch_add(1, std::memory_order_acq_rel);
      inflight.fetch_sub(1, std::memory_order_seq_cst);
    }
  }
  // has the flag of chunk i of a widening job arrived?
  static bool chunk_landed(const Job& j, const size_t i) {
    const size_t half = j.nchunks.load(std::memory_order_relaxed) / 2;
    const volatile uint32_t* fl = i >= half ? j.flags2.load(std::memory_order_relaxed) : j.flags.load(std::memory_order_relaxed);
    return !fl || fl[i >= half ? i - half : i] == j.epoch.load(std::memory_order_relaxed);
  }
  // one chunk of a widening job; the submitting thread passes `drained` and gives up (abort_widen) when the stream has run dry
  // without the chunk's flag.  false: not widened (given up)
  bool widen_chunk(Job& j, const size_t i, bool (*drained)(void*), void* arg) {
    const size_t half = j.nchunks.load(std::memory_order_relaxed) / 2;
    const int part = i >= half ? 1 : 0;
    const size_t lc = i - (part ? half : 0);
    const volatile uint32_t* fl = part ? j.flags2.load(std::memory_order_relaxed) : j.flags.load(std::memory_order_relaxed);
    if (fl) {
      const uint32_t epoch = j.epoch.load(std::memory_order_relaxed);
      for (uint32_t spins = 1; fl[lc] != epoch; spins++) {
        if (abort_widen.load(std::memory_order_relaxed)) return false;
        MRH_CPU_RELAX();
        if (drained && (spins & 1023u) == 0 && drained(arg)) {
          // the stream has run dry: everything the launch wrote is visible, or about to be — only a flag that stays away is an error
          const int64_t t = now_ns();
          while (fl[lc] != epoch && now_ns() - t < 200000000) MRH_CPU_RELAX();
          if (fl[lc] == epoch) break;
          abort_widen.store(1, std::memory_order_relaxed);
          return false;
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    const size_t bytes = j.bytes.load(std::memory_order_relaxed);
    const size_t off = lc * kWidenChunk, len = std::min(kWidenChunk, bytes - off);
    const float* src = (const float*) ((part ? j.src2.load(std::memory_order_relaxed) : j.src.load(std::memory_order_relaxed)) + off);
    double* dst = (double*) ((part ? j.dst2.load(std::memory_order_relaxed) : j.dst.load(std::memory_order_relaxed)) + 2 * off);
    widen_floats(dst, src, len / sizeof(float));
    return true;
  }
  // both parts of a widening job through the pool (the calling thread works too); false: gave up on a flag
  bool widen(double* const dst[2], const float* const src[2], const volatile uint32_t* const flags[2], const uint32_t epoch, const size_t nfloat,
             bool (*drained)(void*), void* arg) {
    if (!started) start();
    const size_t bytes = nfloat * sizeof(float);
    const size_t per = (bytes + kWidenChunk - 1) / kWidenChunk, nc = 2 * per;
    if (nc == 0) return true;
    quiesce();  // a straggler of the previous widening job still reads its descriptor
    abort_widen.store(0, std::memory_order_relaxed);
    const uint64_t g = generation.load(std::memory_order_relaxed) + 1;  // one submitter at a time (g_copy_mutex)
    Job& j = jobs[g & 1];
    j.dst.store((char*) dst[0], std::memory_order_relaxed); j.src.store((const char*) src[0], std::memory_order_relaxed);
    j.dst2.store((char*) dst[1], std::memory_order_relaxed); j.src2.store((const char*) src[1], std::memory_order_relaxed);
    j.flags.store(flags[0], std::memory_order_relaxed); j.flags2.store(flags[1], std::memory_order_relaxed);
    j.epoch.store(epoch, std::memory_order_relaxed);
    j.bytes.store(bytes, std::memory_order_relaxed); j.nchunks.store(nc, std::memory_order_relaxed);
    j.widen.store(1, std::memory_order_relaxed);
    const bool stateful = nc <= kMaxStates;
    for (size_t i = 0; i < std::min(nc, kMaxStates); i++) state[i].store(0, std::memory_order_relaxed);
    done.store(0, std::memory_order_relaxed);
    ticket.store((g & 0xFFFFFFFFull) << 32, std::memory_order_release);
    generation.store(g, std::memory_order_release);
    if (sleepers.load(std::memory_order_acquire) > 0) { std::lock_guard<std::mutex> lk(m); cv.notify_all(); }
    {  // work(g) with the stream check in the flag wait
      size_t i;
      while (claim(g, j, i)) {
        if (widen_chunk(j, i, drained, arg) && i < kMaxStates) state[i].store(2, std::memory_order_release);
        done.fetch_add(1, std::memory_order_acq_rel);
      }
    }
    if (stateful) {
      // The tickets are out; at most one chunk per helper is still under way.  In chunk order: wait for it while it can still be
      // on its way (the flag has not arrived, or arrived less than a grace of 40 us ago — a chunk is ~10 us of work), then redo it.
      for (size_t i = 0; i < nc && !abort_widen.load(std::memory_order_relaxed); i++) {
        int64_t landed_at = 0;
        uint32_t spins = 0;
        while (state[i].load(std::memory_order_acquire) != 2) {
          if (abort_widen.load(std::memory_order_relaxed)) break;
          if (!chunk_landed(j, i)) {  // nobody can have widened it yet: the wait is for the device (with the stream check)
            if (drained && (++spins & 1023u) == 0 && drained(arg)) {
              const int64_t t = now_ns();
              while (!chunk_landed(j, i) && now_ns() - t < 200000000) MRH_CPU_RELAX();
              if (!chunk_landed(j, i)) { abort_widen.store(1, std::memory_order_relaxed); break; }
            }
            MRH_CPU_RELAX();
            continue;
          }
          const int64_t now = now_ns();
          if (!landed_at) landed_at = now;
          if (now - landed_at > 40000) {  // its helper lost its core (or is slow): the same bytes, written here
            if (widen_chunk(j, i, drained, arg)) { state[i].store(2, std::memory_order_release); redone.fetch_add(1, std::memory_order_relaxed); }
            break;
          }
          MRH_CPU_RELAX();
        }
      }
    } else {
      while (done.load(std::memory_order_acquire) < nc && !abort_widen.load(std::memory_order_relaxed)) MRH_CPU_RELAX();
    }
    // (the descriptor keeps `widen` set: a straggler reads it after this call has returned; the next job rewrites it behind quiesce())
    return abort_widen.load(std::memory_order_relaxed) == 0;
  }
  // wake the helpers now and keep them spinning for a millisecond: a widening job is on its way
  void prewake() {
    if (!started) start();
    spin_until_ns.store(now_ns() + 1500000, std::memory_order_relaxed);
    if (sleepers.load(std::memory_order_acquire) > 0) {
      quiesce();
      const uint64_t g = generation.load(std::memory_order_relaxed) + 1;  // an empty job: nothing to claim
      Job& j = jobs[g & 1];
      j.widen.store(0, std::memory_order_relaxed);
      j.bytes.store(0, std::memory_order_relaxed); j.nchunks.store(0, std::memory_order_relaxed);
      done.store(0, std::memory_order_relaxed);
      ticket.store((g & 0xFFFFFFFFull) << 32, std::memory_order_release);
      generation.store(g, std::memory_order_release);
      std::lock_guard<std::mutex> lk(m); cv.notify_all();
    }
  }
  void helper() {
    uint64_t seen = generation.load(std::memory_order_acquire);
    for (;;) {
      // wait for the next job: spin ~100 us (a frame loop submits every 40-100 us), then sleep
      const auto t0 = std::chrono::steady_clock::now();
      uint64_t g;
      int spins = 0;
      while ((g = generation.load(std::memory_order_acquire)) == seen) {
        MRH_CPU_RELAX();
        if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(150) &&
            now_ns() > spin_until_ns.load(std::memory_order_relaxed)) {
          std::unique_lock<std::mutex> lk(m);
          sleepers.fetch_add(1);
          cv.wait(lk, [&] { return generation.load(std::memory_order_acquire) != seen; });
          sleepers.fetch_sub(1);
        }
      }
      seen = g;
      work(g);
    }
  }
  void start() {
    started = true;
    int n = 3;
    if (const char* e = getenv("MRH_COPY_THREADS")) n = atoi(e);
    const int hw = (int) std::thread::hardware_concurrency();
    if (hw > 0 && n > hw - 1) n = hw - 1;
    for (int i = 0; i < n; i++) {
      threads.emplace_back([this] { helper(); });
      threads.back().detach();  // they sleep on the condition variable when idle; the pool lives as long as the process
    }
  }
  void copy(void* d, const void* s_, size_t n) {
    if (!started) start();
    if (threads.empty() || n < 4 * kChunk) { copy_chunk(d, s_, n); return; }
    quiesce();  // a straggler of a widening job still reads that job's descriptor
    const size_t nc = (n + kChunk - 1) / kChunk;
    const uint64_t g = generation.load(std::memory_order_relaxed) + 1;  // one submitter at a time (g_copy_mutex)
    Job& j = jobs[g & 1];
    j.dst.store((char*) d, std::memory_order_relaxed); j.src.store((const char*) s_, std::memory_order_relaxed);
    j.bytes.store(n, std::memory_order_relaxed); j.nchunks.store(nc, std::memory_order_relaxed);
    j.widen.store(0, std::memory_order_relaxed);
    done.store(0, std::memory_order_relaxed);  // no ticket of an earlier job is outstanding: they all completed before their copy() returned
    ticket.store((g & 0xFFFFFFFFull) << 32, std::memory_order_release);
    generation.store(g, std::memory_order_release);
    if (sleepers.load(std::memory_order_acquire) > 0) { std::lock_guard<std::mutex> lk(m); cv.notify_all(); }
    size_t i;
    while (claim(g, j, i)) {
      const size_t off = i * kChunk, len = std::min(kChunk, n - off);
      copy_chunk((char*) d + off, (const char*) s_ + off, len);
      done.fetch_add(1, std::memory_order_acq_rel);
    }
    while (done.load(std::memory_order_acquire) < nc) MRH_CPU_RELAX();  // the source is the caller's: nobody may still read it on return
  }
};
CopyPool* copy_pool() {
  static CopyPool* pool = new CopyPool();  // never destroyed: detached helpers may still be parked on it at exit
  return pool;
}
std::mutex g_copy_mutex;  // one job at a time (contexts on different host threads share the pool)
void copy_to_staging(void* dst, const void* src, size_t n) {
  std::lock_guard<std::mutex> lk(g_copy_mutex);
  copy_pool()->copy(dst, src, n);
}
bool widen_from_staging(double* const dst[2], const float* const src[2], const volatile u32* const flags[2], u32 epoch, size_t nfloat,
                        bool (*drained)(void*), void* arg) {
  std::lock_guard<std::mutex> lk(g_copy_mutex);
  return copy_pool()->widen(dst, src, flags, epoch, nfloat, drained, arg);
}
void widen_prewake() {
  std::lock_guard<std::mutex> lk(g_copy_mutex);
  copy_pool()->prewake();
}
// no helper is still reading a staging buffer or writing a result array of an earlier widening job (CopyPool: `inflight`)
void widen_quiesce() {
  std::lock_guard<std::mutex> lk(g_copy_mutex);
  copy_pool()->quiesce();
}
uint64_t widen_redone() { return copy_pool()->redone.load(std::memory_order_relaxed); }
#else
void copy_to_staging(void* dst, const void* src, size_t n);
bool widen_from_staging(double* const dst[2], const float* const src[2], const volatile u32* const flags[2], u32 epoch, size_t nfloat,
                        bool (*drained)(void*), void* arg) { return false; }
void widen_prewake() {}
void widen_quiesce() {}
uint64_t widen_redone() { return 0; }
#endif

// one host image into the next slot of its ring: wait until the slot is free, copy into pinned staging (the caller's
// buffer is free on return), enqueue the H2D on the copy stream
int upload_image(mrh_ctx* c, UpRing& ring, const void* src, const size_t bytes, const void** out_dev) {
  if (!c->copy_ready) {
    // One copy stream per image kind: the depth and the colour image of a frame then move through two SDMA engines side by
    // side (64 us per frame instead of 77 on one stream).  A copy kernel pulling the pinned buffer over PCIe is faster on its
    // own (48 GB/s against 25-30, tools/micro/h2d_paths.hip) but finds no wave slots while k_back fills every SIMD's
    // registers, neither with stream priority nor with CU masks (tools/micro/cu_mask_overlap.hip: a masked stream costs the
    // big kernel 14 %): measured at 73-75 us per frame inside the library, and dropped.
    for (UpRing* r : {&c->up_depth, &c->up_rgb}) HIP_TRY(c, hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    for (hipEvent_t& e : c->frame_done) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->copy_ready = true;
  }
  const int next = (ring.cur + 1) % 3;
  // a frame that mrh_integrate kept back (flush_deferred) has not marked its slots yet: before the ring comes round to one of them, it runs
  if (c->deferred.on && next == c->deferred.ring[&ring == &c->up_rgb ? 1 : 0].cur) {
    const int frc = flush_deferred(c);
    if (frc < 0) return frc;
  }
  UpSlot& u = ring.s[next];
  if (u.last_seq) HIP_TRY(c, hipEventSynchronize(c->frame_done[u.last_seq % 8]));  // this mark or a later one of the same stream
  if (u.copied_rec) HIP_TRY(c, hipEventSynchronize(u.copied));
  if (bytes > u.cap) {
    if (u.h) HIP_TRY(c, hipHostFree(u.h));
    if (u.d) HIP_TRY(c, hipFree(u.d));
    u.h = u.d = nullptr; u.cap = 0;
    HIP_TRY(c, hipHostMalloc(&u.h, bytes, hipHostMallocDefault));
    HIP_TRY(c, hipMalloc(&u.d, bytes));
    u.cap = bytes;
    if (!u.copied) HIP_TRY(c, hipEventCreateWithFlags(&u.copied, hipEventDisableTiming));
  }
  copy_to_staging(u.h, src, bytes);
  // (Round 5 measured the two runtime calls below on a thread of their own, so that the caller is back in its code ~10 us earlier:
  // uploads alone 56 -> 43 us per frame, but a FRAME stays at 62-64 us — mrh_integrate then waits for that thread to have
  // recorded the event before it can enqueue the stream wait, and the chain staging -> copy call -> stream wait -> launches is
  // on the caller's critical path whoever makes the calls.  Removed again; profiles/r05/README.md.)
  HIP_TRY(c, hipMemcpyAsync(u.d, u.h, bytes, hipMemcpyHostToDevice, ring.stream));
  HIP_TRY(c, hipEventRecord(u.copied, ring.stream));
  u.copied_rec = true;
  u.last_seq = 0;
  ring.last_copy = u.copied;  // a ring's copies are ordered on its stream: the newest event covers the earlier ones
  ring.waited[0] = ring.waited[1] = false;
  ring.cur = next;
  *out_dev = u.d;
  return MRH_OK;
}

// before kernels that read the images: `reader` — the stream those kernels are launched on: the front stream for a pipelined
// frame (its integration reads the cleaned copy the front half wrote), the main stream otherwise — waits for the newest uploads
int send_uploads(mrh_ctx* c, hipStream_t reader) {
  const int w = (reader == c->stream) ? 0 : 1;
  for (UpRing* r : {&c->up_depth, &c->up_rgb})
    if (r->last_copy && !r->waited[w]) {
      // a transfer the host already sees complete needs no wait packet (a kernel launched from here on reads what it wrote)
      const hipError_t q = hipEventQuery(r->last_copy);
      if (q == hipErrorNotReady) {
        (void) hipGetLastError();
        HIP_TRY(c, hipStreamWaitEvent(reader, r->last_copy, 0));
      } else if (q != hipSuccess) {
        return fail(c, MRH_ERR_DEVICE, "image transfer: %s", hipGetErrorString(q));
      }
      r->waited[w] = true;
    }
  return MRH_OK;
}

// after the kernels of a frame (or of a seeding call) are enqueued: mark the ring slots they read, report the pool level
int mark_frame(mrh_ctx* c) {
  UpSlot* used[2] = {nullptr, nullptr};
  if (c->up_depth.cur >= 0 && c->d_depth == c->up_depth.s[c->up_depth.cur].d) used[0] = &c->up_depth.s[c->up_depth.cur];
  if (c->up_rgb.cur >= 0 && c->d_rgb == c->up_rgb.s[c->up_rgb.cur].d) used[1] = &c->up_rgb.s[c->up_rgb.cur];
  if (!used[0] && !used[1] && !c->peek_enabled) return MRH_OK;
  const uint64_t seq = c->frame_seq++;
  if (!c->frame_done[0])
    for (hipEvent_t& e : c->frame_done) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const bool lazy = c->last_frame_lazy && c->npend;  // the frame's integration is not enqueued yet (launch_pending)
  if (c->peek_enabled) {
    if (!c->peek_done[0])
      for (hipEvent_t& e : c->peek_done) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (lazy) {
      // The report of a pipelined frame is written behind its integration, by launch_pending; until then the mark does not
      // exist for the peeks (they fall back to an older one and say how many frames behind it is).  A report launched here would
      // sit behind the integration of an EARLIER frame only, and an event on the front stream says nothing about it at all.
      c->peek_seq[seq % 8] = 0;
      c->pendq[c->npend - 1].report_seq = seq;
    } else {
      k_report<<<1, 64, 0, c->stream>>>(&c->tab.ctr[CTR_HEAP_FINE], c->h_peek + 8 * (seq % 8));  // ctr[0 .. 4]
      HIP_TRY(c, hipGetLastError());
      HIP_TRY(c, hipEventRecord(c->peek_done[seq % 8], c->stream));
      c->peek_seq[seq % 8] = seq;
    }
  }
  if (used[0] || used[1]) {
    // the raw images of a pipelined frame are read by its front half, on the front stream (its integration reads the cleaned copy)
    HIP_TRY(c, hipEventRecord(c->frame_done[seq % 8], lazy ? c->stream_front : c->stream));
    for (UpSlot* u : used) if (u) u->last_seq = seq;
  }
  return MRH_OK;
}

}  // namespace

int mrh_upload_depth(mrh_ctx* c, const float* depth, int rows, int cols) {
  int rc = ensure_device(c, "mrh_upload_depth");
  if (rc) return rc;
  if (!depth || rows <= 0 || cols <= 0) return fail(c, MRH_ERR_INVALID_ARG, "mrh_upload_depth: bad argument");
  const void* dev = nullptr;
  rc = upload_image(c, c->up_depth, depth, (size_t) rows * cols * sizeof(float), &dev);
  if (rc) return rc;
  c->d_depth = (const float*) dev;
  c->depth_rows = rows; c->depth_cols = cols;
  return MRH_OK;
}

int mrh_upload_rgb(mrh_ctx* c, const uint8_t* rgb, int rows, int cols) {
  int rc = ensure_device(c, "mrh_upload_rgb");
  if (rc) return rc;
  if (!rgb || rows <= 0 || cols <= 0) return fail(c, MRH_ERR_INVALID_ARG, "mrh_upload_rgb: bad argument");
  const void* dev = nullptr;
  rc = upload_image(c, c->up_rgb, rgb, (size_t) rows * cols * 3, &dev);
  if (rc) return rc;
  c->d_rgb = (const uint8_t*) dev;
  c->rgb_rows = rows; c->rgb_cols = cols;
  return MRH_OK;
}

int mrh_set_depth_device(mrh_ctx* c, const float* d_depth, int rows, int cols) {
  if (!c || !d_depth || rows <= 0 || cols <= 0) return fail(c, MRH_ERR_INVALID_ARG, "mrh_set_depth_device: bad argument");
  if (c->deferred.on) { (void) hipSetDevice(c->device); const int drc = flush_deferred(c); if (drc < 0) return drc; }
  c->d_depth = d_depth; c->depth_rows = rows; c->depth_cols = cols;
  c->up_depth.cur = -1;
  return MRH_OK;
}

int mrh_set_rgb_device(mrh_ctx* c, const uint8_t* d_rgb, int rows, int cols) {
  if (!c || !d_rgb || rows <= 0 || cols <= 0) return fail(c, MRH_ERR_INVALID_ARG, "mrh_set_rgb_device: bad argument");
  if (c->deferred.on) { (void) hipSetDevice(c->device); const int drc = flush_deferred(c); if (drc < 0) return drc; }
  c->d_rgb = d_rgb; c->rgb_rows = rows; c->rgb_cols = cols;
  c->up_rgb.cur = -1;
  return MRH_OK;
}

// voxel_data_structures.cpp:90-110 VoxelContainer::integrate, as one sync-free kernel chain
static int integrate_frame(mrh_ctx* c, int n_frames_invalidate);

static int integrate_checks(mrh_ctx* c);
static void prewarm_maybe(mrh_ctx* c);
int mrh_integrate(mrh_ctx* c, int n_frames_invalidate) {
  int rc = ensure_device(c, "mrh_integrate");
  if (rc) return rc;
  rc = flush_deferred(c);  // the frame of the previous call: its images have landed meanwhile
  if (rc < 0) return rc;
  // is this a frame of host images whose transfers are still on their way?
  const bool up_d = c->up_depth.cur >= 0 && c->d_depth == c->up_depth.s[c->up_depth.cur].d && !c->up_depth.waited[0] && !c->up_depth.waited[1];
  const bool up_c = c->up_rgb.cur >= 0 && c->d_rgb == c->up_rgb.s[c->up_rgb.cur].d && !c->up_rgb.waited[0] && !c->up_rgb.waited[1];
  if (c->defer_uploads && (up_d || up_c) && c->p.shard_count <= 1 && !c->comm && !c->profile) {
    rc = integrate_checks(c);  // what can be wrong with the call is reported by the call
    if (rc) return rc;
    mrh_ctx::DeferredFrame& d = c->deferred;
    d.on = true;
    d.n_inval = n_frames_invalidate;
    d.cam = c->cam;
    d.d_depth = c->d_depth; d.d_rgb = c->d_rgb;
    d.depth_rows = c->depth_rows; d.depth_cols = c->depth_cols; d.rgb_rows = c->rgb_rows; d.rgb_cols = c->rgb_cols;
    const UpRing* rings[2] = {&c->up_depth, &c->up_rgb};
    for (int i = 0; i < 2; i++) d.ring[i] = {rings[i]->cur, rings[i]->last_copy, {rings[i]->waited[0], rings[i]->waited[1]}};
    return MRH_OK;
  }
  rc = integrate_frame(c, n_frames_invalidate);
  if (rc < 0) return rc;
  const int mrc = mark_frame(c);
  if (!mrc && rc == MRH_OK) prewarm_maybe(c);
  return mrc ? mrc : rc;
}

extern "C++" {
namespace {
// runs the frame mrh_integrate kept back, with the inputs it was issued under; the context's current inputs (the next frame's
// pose and images may have arrived meanwhile) are put back afterwards
int flush_deferred(mrh_ctx* c) {
  mrh_ctx::DeferredFrame& d = c->deferred;
  if (!d.on) return MRH_OK;
  d.on = false;
  UpRing* rings[2] = {&c->up_depth, &c->up_rgb};
  mrh_ctx::DeferredFrame now;
  now.cam = c->cam;
  now.d_depth = c->d_depth; now.d_rgb = c->d_rgb;
  now.depth_rows = c->depth_rows; now.depth_cols = c->depth_cols; now.rgb_rows = c->rgb_rows; now.rgb_cols = c->rgb_cols;
  for (int i = 0; i < 2; i++) now.ring[i] = {rings[i]->cur, rings[i]->last_copy, {rings[i]->waited[0], rings[i]->waited[1]}};
  c->cam = d.cam;
  c->d_depth = d.d_depth; c->d_rgb = d.d_rgb;
  c->depth_rows = d.depth_rows; c->depth_cols = d.depth_cols; c->rgb_rows = d.rgb_rows; c->rgb_cols = d.rgb_cols;
  for (int i = 0; i < 2; i++) { rings[i]->cur = d.ring[i].cur; rings[i]->last_copy = d.ring[i].last_copy; rings[i]->waited[0] = d.ring[i].waited[0]; rings[i]->waited[1] = d.ring[i].waited[1]; }
  int rc = integrate_frame(c, d.n_inval);
  if (rc >= 0) {
    const int mrc = mark_frame(c);
    if (mrc) rc = mrc;
  }
  c->cam = now.cam;
  c->d_depth = now.d_depth; c->d_rgb = now.d_rgb;
  c->depth_rows = now.depth_rows; c->depth_cols = now.depth_cols; c->rgb_rows = now.rgb_rows; c->rgb_cols = now.rgb_cols;
  for (int i = 0; i < 2; i++) {
    rings[i]->cur = now.ring[i].cur;
    if (now.ring[i].last_copy != d.ring[i].last_copy) {  // a newer image of this kind has arrived: its transfer has not been waited for
      rings[i]->last_copy = now.ring[i].last_copy;
      rings[i]->waited[0] = now.ring[i].waited[0]; rings[i]->waited[1] = now.ring[i].waited[1];
    }  // else: the same transfer, and what the frame has waited for stays waited for
  }
  return rc;
}
}  // namespace
}  // extern "C++"

extern "C++" {
namespace {

int take_event_pair(mrh_ctx* c, EvPair& e) {
  if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); return MRH_OK; }
  if (c->ev_pending.size() >= 4096) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int r = drain_events(c);
    if (r) return r;
    e = c->ev_pool.back(); c->ev_pool.pop_back();
    return MRH_OK;
  }
  HIP_TRY(c, hipEventCreate(&e.a)); HIP_TRY(c, hipEventCreate(&e.b));
  return MRH_OK;
}

Lists ring_lists(const mrh_ctx* c, const int i) {
  return Lists{c->ring_vis[i], c->ring_bbox[i], c->ring_cfree[i], c->ring_zmin[i], (u32) c->num_blocks};
}

// Behind the pipelined frames issued so far (their integrations are all on the main stream, each behind its front half), the
// zombies nobody wanted leave the table: k_reclaim runs alone on the main stream — the front stream is idle once the last
// integration has started, and nothing is enqueued on it before the host has seen the main stream drain (front_needs_sync).
// the integration of the newest pipelined frame, behind its front half
int launch_pending(mrh_ctx* c, const bool count_skips = false) {
  if (!c->npend) return MRH_OK;
  const mrh_ctx::PendingBack pb = c->pendq[0];  // the oldest
  for (int i = 1; i < c->npend; i++) c->pendq[i - 1] = c->pendq[i];
  c->npend--;
  hipStream_t s = c->stream;
  static const bool always_wait = getenv("MRH_PIPE_ALWAYS_WAIT") != nullptr;  // A/B: the wait packet whatever the query says
  const hipError_t q = always_wait ? hipErrorNotReady : hipEventQuery(c->ev_front[pb.ring]);
  if (q == hipErrorNotReady) {
    (void) hipGetLastError();
    HIP_TRY(c, hipStreamWaitEvent(s, c->ev_front[pb.ring], 0));
    c->dbg_waits++;
  } else if (q != hipSuccess) {
    return fail(c, MRH_ERR_DEVICE, "mrh_integrate: front half of a pipelined frame: %s", hipGetErrorString(q));
  }
  const Map& m = c->map;
  const Tab& t = c->tab;
  const size_t lds = (size_t) 4 * kTileMaxPx * sizeof(uint2);
  if (count_skips) HIP_TRY(c, hipMemsetAsync(&c->tab.ctr[CTR_ZSKIP], 0, sizeof(int), s));  // mrh_get_stats: M of the last frame, exactly
  if (pb.profile)  // U and M of the frame (device-side counters of the roofline numerator): after its front half, before its integration
    k_count_updates<<<c->fused_grid, 256, 0, s>>>(pb.cam, m, t, pb.f, c->d_cnt_partials, CTR_SET0 + 4 * pb.set, pb.L.vis, pb.L.cfree, pb.stamp, pb.count_zombies ? 1 : 0);
#define MRH_KB(FREE, PROF, SAFE, SPH)                                                                                                                 \
  do {                                                                                                                                                \
    if (pb.profile) hipExtLaunchKernelGGL((k_back<FREE, PROF, false, SAFE, 2, SPH>), dim3(c->pipe_grid), dim3(256), (uint32_t) lds, s, pb.ev.a, pb.ev.b, 0u, pb.cam, m, t, pb.f, \
                                          pb.L, pb.set, pb.zero_set, pb.thr, (const float*) nullptr, (const uint8_t*) nullptr, (u32*) nullptr, pb.stamp, pb.seq); \
    else k_back<FREE, PROF, false, SAFE, 2, SPH><<<c->pipe_grid, 256, lds, s>>>(pb.cam, m, t, pb.f, pb.L, pb.set, pb.zero_set, pb.thr, nullptr, nullptr, nullptr, pb.stamp, pb.seq); \
  } while (0)
#define MRH_KB3(FREE, PROF, SAFE) do { if (pb.sph) MRH_KB(FREE, PROF, SAFE, true); else MRH_KB(FREE, PROF, SAFE, false); } while (0)
#define MRH_KB2(FREE, PROF) do { if (pb.safe_div) MRH_KB3(FREE, PROF, true); else MRH_KB3(FREE, PROF, false); } while (0)
  if (pb.free_ && pb.profile) MRH_KB2(true, true);
  else if (pb.free_) MRH_KB2(true, false);
  else MRH_KB2(false, false);
#undef MRH_KB2
#undef MRH_KB3
#undef MRH_KB
  if (pb.profile) c->ev_pending.push_back(pb.ev);
  if (pb.free_) c->zombies_possible = true;
  if (pb.starve) {
    const int src = launch_starve_fused(c, pb.cam, pb.f, pb.L, pb.set, pb.thr, pb.stamp, 2);
    if (src) return src;
    c->zombies_possible = true;
  }
  if (pb.report_seq && c->peek_enabled) {  // the frame's pool report (mark_frame left it to this launch): behind its integration
    k_report<<<1, 64, 0, s>>>(&c->tab.ctr[CTR_HEAP_FINE], c->h_peek + 8 * (pb.report_seq % 8));
    HIP_TRY(c, hipEventRecord(c->peek_done[pb.report_seq % 8], s));
    c->peek_seq[pb.report_seq % 8] = pb.report_seq;  // from here on the mark exists for the peeks
  }
  HIP_TRY(c, hipGetLastError());
  return MRH_OK;
}

int strict_point(mrh_ctx* c) {
  while (c->npend) {
    const int rc = launch_pending(c, c->npend == 1);
    if (rc) return rc;
  }
  if (!c->zombies_possible) return MRH_OK;
  k_reclaim<<<64, 256, 0, c->stream>>>(c->tab, c->fast);
  k_reclaim_done<<<1, 1, 0, c->stream>>>(c->tab);
  // the pool report of the newest mark now understates the free list by the zombies that have just left: written again behind the
  // reclaim, so that a peek after mrh_sync (or after any other flush) reads the level the flush left (with the reclaim period at 64
  // frames the difference is no longer a handful of blocks)
  if (c->peek_enabled && c->frame_seq > 1) {
    const uint64_t seq = c->frame_seq - 1;
    if (c->peek_seq[seq % 8] == seq && c->peek_done[seq % 8]) {
      k_report<<<1, 64, 0, c->stream>>>(&c->tab.ctr[CTR_HEAP_FINE], c->h_peek + 8 * (seq % 8));
      HIP_TRY(c, hipEventRecord(c->peek_done[seq % 8], c->stream));
    }
  }
  c->zombies_possible = false;
  c->lazy_run = 0;
  c->front_needs_sync = true;
  HIP_TRY(c, hipGetLastError());
  return MRH_OK;
}

int ensure_pipe_buffers(mrh_ctx* c, const size_t npix) {
  if (!c->stream_front) {
    const size_t cap = c->num_blocks;
    // (a high-priority front stream, a ring of eight and integrations deferred by two calls were measured: no difference)
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream_front, hipStreamNonBlocking));
    // (hipEventDisableSystemFence on these events — they order two streams of one device — was measured in round 5: no difference)
    for (hipEvent_t& e : c->ev_front) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->ring_vis[0] = c->tab.compact; c->ring_bbox[0] = c->fast.bbox; c->ring_cfree[0] = c->d_cfree; c->ring_zmin[0] = c->d_zmin;
    for (int i = 1; i < kPipeRing; i++) {
      HIP_TRY(c, hipMalloc((void**) &c->ring_vis[i], cap * sizeof(int4)));
      HIP_TRY(c, hipMalloc((void**) &c->ring_bbox[i], cap * sizeof(int4)));
      HIP_TRY(c, hipMalloc((void**) &c->ring_cfree[i], cap * sizeof(int4)));
      HIP_TRY(c, hipMalloc((void**) &c->ring_zmin[i], cap * sizeof(float)));
    }
    HIP_TRY(c, hipMalloc((void**) &c->fast.zlist, cap * sizeof(int4)));
    HIP_TRY(c, hipMalloc((void**) &c->want_ring, (size_t) kPipeRing * c->slots * sizeof(u32)));
    HIP_TRY(c, hipMemsetAsync(c->want_ring, 0, (size_t) kPipeRing * c->slots * sizeof(u32), c->stream));  // stamps start at 1
    if (!c->h_levels) HIP_TRY(c, hipHostMalloc((void**) &c->h_levels, 4 * sizeof(int), hipHostMallocDefault));
    c->h_levels[0] = (int) c->num_blocks - 1; c->h_levels[1] = 0; c->h_levels[2] = -1;
    c->tab.h_levels = c->h_levels;
    c->front_needs_sync = true;  // the memset above
  }
  if (c->pipe_npix < npix) {
    {
      const int rc = strict_point(c);  // the pending integration reads the buffers that are about to go
      if (rc) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream_front));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (uint2*& d : c->pipe_dcx) { if (d) HIP_TRY(c, hipFree(d)); d = nullptr; }
    c->pipe_npix = 0;
    for (uint2*& d : c->pipe_dcx) HIP_TRY(c, hipMalloc((void**) &d, npix * sizeof(uint2)));
    c->pipe_npix = npix;
  }
  return MRH_OK;
}

// One single-resolution pinhole frame of a pipelining context.
//   pipelined: front stream: k_front<LAZY> (+ event) | main stream: wait for that event, k_back<LZ = 2>.  The front stream never
//              waits for the main one, so this frame's front half runs next to the integration of the frame(s) before it;
//              the host only holds back when it is kPipeRing - 1 frames ahead of the integration that has started.
//   serial:    [k_reclaim] -> k_front -> k_back (-> the starve passes), all on the main stream, no zombies anywhere.
// A serial frame comes after anything else touched the map, every `pipe_period` frames (the reclaim bounds the zombies), on
// starve frames, and while the pool is short of room: zombies hold their pool slots until the reclaim, so a pool that is
// nearly full is fused serially — the reference's accounting, exactly.
int integrate_lazy(mrh_ctx* c, const int max_num_frames, const bool starve_now) {
  int rc = MRH_OK;
  hipStream_t s = c->stream;
  const Cam& k = c->cam;
  const Map& m = c->map;
  const Tab& t = c->tab;
  const size_t npix = (size_t) k.rows * k.cols;
  rc = ensure_pipe_buffers(c, npix);
  if (rc) return rc;
  if (c->fast_summaries_stale) {  // a general frame (spherical camera) ran since: rebuild the GC summaries once
    rc = strict_point(c);
    if (rc) return rc;
    k_summarize_all<<<2048, 256, 0, s>>>(t, c->fast);
    c->fast_summaries_stale = false;
    c->front_needs_sync = true;
  }
  const int tiles_x = (k.cols + kRayTile - 1) / kRayTile, tiles_y = (k.rows + kRayTile - 1) / kRayTile;
  const int n_tiles = tiles_x * tiles_y;
  // room in the pool, as the last integration launch reported it (a few frames old: the margins are generous)
  const int64_t free_known = (int64_t) ((volatile int*) c->h_levels)[0] + 1, zombies_known = ((volatile int*) c->h_levels)[1];
  const bool roomy = free_known >= (int64_t) (c->num_blocks / 4) && zombies_known <= (int64_t) (c->num_blocks / 8);
  // a caller that synchronises (or asks for statistics, a mesh, ...) after EVERY frame gains nothing from the pipeline and would pay
  // for its flush each time: three frames in a row that found the pipeline flushed switch to serial frames, the first frame that
  // follows another frame directly switches back
  c->sync_streak = c->flushed_since_frame ? c->sync_streak + 1 : 0;
  c->flushed_since_frame = false;
  // images that come from the host (mrh_upload_*) make the frame loop host- and link-bound (staging copy + 2.15 MB over PCIe: ~60 us
  // per frame at 640x480 against ~40 us of GPU work): nothing to gain from overlapping kernels, and the second stream's events
  // only add to the host's bill (measured: 77 us per frame pipelined, 64 serial) — such frames are fused serially unless
  // MRH_PIPE_UPLOADS=1 says otherwise (the test-suite sets it, so that its upload-fed streams exercise the pipeline)
  const bool resident_inputs = c->up_depth.cur < 0 && c->up_rgb.cur < 0;
  // a starve frame stays a frame of the pipeline when its starve step can take the fused launches (round 6; before: every starve
  // frame flushed the pipeline, ran serially and left a host synchronisation in front of the next pipelined frame)
  const bool starve_in_pipe = starve_now && starve_fused_ok(c) && !getenv("MRH_STARVE_SERIAL");
  const bool lazy = c->pipe && (!starve_now || starve_in_pipe) && roomy && c->lazy_run < c->pipe_period && c->sync_streak < 3 && (resident_inputs || c->pipe_uploads) &&
                    !getenv("MRH_PIPE_SERIAL");
  if (!lazy) {
    rc = strict_point(c);
    if (rc) return rc;
  }
  rc = send_uploads(c, lazy ? c->stream_front : s);  // the raw images are read by the front half
  if (rc) return rc;
  const int seq = (int) (c->pipe_seq & 0x3FFFFFFF);
  const int ring = lazy ? (int) (c->pipe_seq % kPipeRing) : 0;  // a serial frame runs behind everything on the main stream: any slot
  c->pipe_seq++;                                                // is free for it, and the starve passes walk slot 0's lists
  const int set = (int) (c->fast_frames % kListSets), zero_set = (set + kListSets - 1) % kListSets;
  c->frame_parity = set;
  c->fast_frames++;
  c->fast.dcx = c->pipe_dcx[ring];
  c->fast.want = c->want_ring + (size_t) ring * c->slots;
  const Fast f = c->fast;
  const u32 stamp = (u32) ((c->frames + 1) & 0x3FFFFFFFu);
  const float gc_thr = m.trunc + m.trunc_scale * k.max_depth;  // getTruncation(camera.maxDepth(), ...), vds.cu:1720
  const bool safe_div = m.half_vs_two_steps || m.wsum_two_steps;  // the short divisions failed their check at mrh_create
  const int gc_on = max_num_frames > 0 ? 1 : 0;
  const bool sph = c->spherical;
  c->frame_gc_inline = max_num_frames > 0 && !starve_now;
  const Lists L = ring_lists(c, ring);
  const size_t lds = (size_t) 4 * kTileMaxPx * sizeof(uint2);
  EvPair ev = {nullptr, nullptr}, evf = {nullptr, nullptr};
  if (c->profile) {
    rc = take_event_pair(c, evf);
    if (rc) return rc;
    rc = take_event_pair(c, ev);
    if (rc) return rc;
  }
  if (lazy) {
    if (c->front_needs_sync) {  // the main stream erased keys / pushed the free list (reclaim, a serial frame, any other entry
      HIP_TRY(c, hipStreamSynchronize(s));  // point) after the front stream last looked: the front half must see all of it
      c->front_needs_sync = false;
      c->pipe_base = c->pipe_seq - 1;
    }
    // ring slot `ring` was last used by frame seq - kPipeRing; its integration is complete once the one after it has started,
    // and that one has also cleared the list-counter set this frame appends to
    {
      const int64_t need = (int64_t) seq - kPipeRing + 2;
      if (need > (int64_t) c->pipe_base) {
        const auto t0 = std::chrono::steady_clock::now();
        while ((int64_t) ((volatile int*) c->h_levels)[2] < need) {
          MRH_CPU_RELAX();
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) return fail(c, MRH_ERR_DEVICE, "mrh_integrate: the integration of frame %lld never started", (long long) need);
        }
        c->dbg_spin_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      }
    }
    c->dbg_lazy_frames++;
    const auto t_api = std::chrono::steady_clock::now();
    hipStream_t a = c->stream_front;
#define MRH_KF(PROF, LAZY, SPH, STREAM)                                                                                                                \
  do {                                                                                                                                                \
    if (PROF) hipExtLaunchKernelGGL((k_front<PROF, false, LAZY, SPH>), dim3(n_tiles + c->sweep_wgs), dim3(256), 0, STREAM, evf.a, evf.b, 0u, k, m, t, f, L, c->d_depth, c->d_rgb, \
                                    tiles_x, n_tiles, stamp, set, gc_on, gc_thr, 0, 0, (const int*) nullptr);                                          \
    else k_front<PROF, false, LAZY, SPH><<<n_tiles + c->sweep_wgs, 256, 0, STREAM>>>(k, m, t, f, L, c->d_depth, c->d_rgb, tiles_x, n_tiles, stamp, set, gc_on, gc_thr, 0, 0, nullptr); \
  } while (0)
#define MRH_KF2(LAZY, STREAM)                                                                                                                         \
  do {                                                                                                                                                \
    if (c->profile && sph) MRH_KF(true, LAZY, true, STREAM); else if (c->profile) MRH_KF(true, LAZY, false, STREAM);                                   \
    else if (sph) MRH_KF(false, LAZY, true, STREAM); else MRH_KF(false, LAZY, false, STREAM);                                                          \
  } while (0)
    MRH_KF2(true, a);
    HIP_TRY(c, hipEventRecord(c->ev_front[ring], a));
    if (c->profile) c->ev_pending_front.push_back(evf);
    // the integration of the PREVIOUS pipelined frame goes out now (its front half ran a frame ago: usually no wait), this
    // frame's is left for the next call
    const bool zombies_before = c->zombies_possible;
    while (c->npend >= c->pipe_defer) {
      rc = launch_pending(c);
      if (rc) return rc;
    }
    mrh_ctx::PendingBack& pb = c->pendq[c->npend++];
    pb.on = true;
    pb.cam = k; pb.f = f; pb.L = L;
    pb.set = set; pb.zero_set = zero_set; pb.ring = ring; pb.seq = seq; pb.stamp = stamp; pb.thr = gc_thr;
    pb.free_ = c->frame_gc_inline; pb.profile = c->profile != 0; pb.safe_div = safe_div; pb.sph = sph;
    pb.count_zombies = zombies_before || c->zombies_possible || c->frame_gc_inline;
    pb.starve = starve_now;
    pb.ev = ev;
    pb.report_seq = 0;
    c->last_frame_lazy = true;
    c->lazy_run++;
    c->frames++;  // frame_tail's bookkeeping; nothing else of it applies (GC runs inside the integration)
    c->dbg_api_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_api).count();
    HIP_TRY(c, hipGetLastError());
    return MRH_OK;
  }
  // ---- serial frame, all on the main stream (strict_point above has flushed and reclaimed)
  c->last_frame_lazy = false;
  MRH_KF2(false, s);
#undef MRH_KF2
#undef MRH_KF
  if (c->profile) {
    c->ev_pending_front.push_back(evf);
    k_count_updates<<<c->fused_grid, 256, 0, s>>>(k, m, t, f, c->d_cnt_partials, CTR_SET0 + 4 * set, L.vis, L.cfree, stamp, 0);
  }
#define MRH_KB(FREE, PROF, SAFE, SPH)                                                                                                                 \
  do {                                                                                                                                                \
    if (c->profile) hipExtLaunchKernelGGL((k_back<FREE, PROF, false, SAFE, 0, SPH>), dim3(c->fused_grid), dim3(256), (uint32_t) lds, s, ev.a, ev.b, 0u, k, m, t, f, L, \
                                          set, zero_set, gc_thr, (const float*) nullptr, (const uint8_t*) nullptr, (u32*) nullptr, stamp, seq);        \
    else k_back<FREE, PROF, false, SAFE, 0, SPH><<<c->fused_grid, 256, lds, s>>>(k, m, t, f, L, set, zero_set, gc_thr, nullptr, nullptr, nullptr, stamp, seq); \
  } while (0)
#define MRH_KB3(FREE, PROF, SAFE) do { if (sph) MRH_KB(FREE, PROF, SAFE, true); else MRH_KB(FREE, PROF, SAFE, false); } while (0)
#define MRH_KB2(FREE, PROF) do { if (safe_div) MRH_KB3(FREE, PROF, true); else MRH_KB3(FREE, PROF, false); } while (0)
  if (c->frame_gc_inline && c->profile) MRH_KB2(true, true);
  else if (c->frame_gc_inline) MRH_KB2(true, false);
  else MRH_KB2(false, false);
#undef MRH_KB2
#undef MRH_KB3
#undef MRH_KB
  c->front_needs_sync = true;  // direct frees on the main stream
  if (c->profile) c->ev_pending.push_back(ev);
  if (starve_now && starve_fused_ok(c)) {
    rc = launch_starve_fused(c, k, f, L, set, gc_thr, stamp, 0);
    if (rc) return rc;
    c->frames++;  // frame_tail's bookkeeping: the summaries and the garbage collection ran inside the tail launch
    return MRH_OK;
  }
  return starve_and_tail(c, max_num_frames);
}

}  // namespace
}  // extern "C++"

// what mrh_integrate rejects before it touches the device
static int integrate_checks(mrh_ctx* c) {
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_integrate: an exchange is pending (call mrh_integrate_resume)");
  if (c->halo_upper) return fail(c, MRH_ERR_STATE, "mrh_integrate: halo blocks of other shards are present (call mrh_drop_blocks(MRH_DROP_HALO) after the extraction)");
  if (!c->has_camera) return fail(c, MRH_ERR_STATE, "mrh_integrate: set_camera has not been called");
  if (c->comm && c->p.shard_count > 1) {  // the starve all-reduce runs over the communicator's ranks: they must be this map's shards
    int cr = 0, cw = 1;
    // MRH_COMM_ALLOW_SHARD_MISMATCH=1 is a test hook for one-GPU boxes (a one-rank group reducing the buffer of one of two shards)
    if (!comm_matches_sharding(c, &cr, &cw) && !getenv("MRH_COMM_ALLOW_SHARD_MISMATCH"))
      return fail(c, MRH_ERR_STATE, "mrh_integrate: the context is shard %d of %d, the attached communicator rank %d of %d", c->p.shard_rank, c->p.shard_count, cr, cw);
  }
  if (!c->d_depth || !c->d_rgb) return fail(c, MRH_ERR_STATE, "mrh_integrate: depth and rgb images are required");
  const Cam& k = c->cam;
  if (c->depth_rows != k.rows || c->depth_cols != k.cols || c->rgb_rows != k.rows || c->rgb_cols != k.cols)
    return fail(c, MRH_ERR_INVALID_ARG, "mrh_integrate: image shape does not match the camera");
  return MRH_OK;
}

// see mrh_ctx::prewarm_on
static void prewarm_maybe(mrh_ctx* c) {
  if (!c->prewarm_on || c->prewarm_done || c->frames != 3 || c->n_extractions || c->f64_link || c->mesh_on_host || c->pending) return;
  c->prewarm_done = true;
  int lev = 0;
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&lev, &c->tab.ctr[CTR_HEAP_FINE], sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
    (void) hipGetLastError();
    return;
  }
  const uint64_t live = (uint64_t) std::max<int64_t>((int64_t) c->num_blocks - ((int64_t) lev + 1), 0);  // fine slots in use (coarse units live in fine slots)
  const uint64_t nv = live * 64;
  if (nv < 65536) return;  // a mesh this small costs its first extraction next to nothing
  try {
    const size_t nf = (size_t) (nv + nv / 4);  // faces: a little above the vertices (closed surfaces: twice; what is seen of a room: ~1.1 x)
    c->V32.resize_discard((size_t) nv * 3 + 4); c->C32.resize_discard((size_t) nv * 3 + 4);
    c->F.resize_discard(nf * 3 + 4);
    c->stage_ctl.resize_discard(kStageHdrWords + 2 * ((std::min(c->V32.cap, c->C32.cap) * 4 + kStageChunk - 1) / kStageChunk + 2));
    if (c->stage_ctl.data()) memset(c->stage_ctl.data(), 0, c->stage_ctl.cap * sizeof(u32));  // no epoch, no flag of an earlier life
    c->V.reserve_unpinned((size_t) nv * 3); c->C.reserve_unpinned((size_t) nv * 3);  // never pinned: mapped and faulted in
    c->V32.clear(); c->C32.clear(); c->F.clear(); c->V.clear(); c->C.clear();  // capacity, not content: the getters still answer "no mesh"
  } catch (...) {
    // no memory for it: the first extraction sizes its buffers itself, as it always did
  }
  (void) hipGetLastError();
}

static int integrate_frame(mrh_ctx* c, int n_frames_invalidate) {
  int rc = integrate_checks(c);
  if (rc) return rc;
  const Cam& k = c->cam;
  const int max_num_frames = n_frames_invalidate < 0 ? c->p.n_frames_invalidate_voxels : n_frames_invalidate;
  hipStream_t s = c->stream;
  const Tab& t = c->tab;
  const Map& m = c->map;
  // the table upkeep rebuilds from the descriptors: the zombies of the pipelined frames leave first
  if (c->zombies_possible && c->census_period >= 0 && (c->table_dirty || c->frames_since_census >= (uint64_t) c->census_period)) {
    rc = strict_point(c);
    if (rc) return rc;
  }
  rc = maintain_table(c, false);
  if (rc) return rc;
  c->frames_since_census++;

  // Multi-resolution maps take the same two launches when that is exact: the fused kernel checks the variance of a
  // fine block right after updating it, which covers every block the reference's checkVarSDF can newly decide on —
  // EXCEPT blocks that changed without being checked (frame 0 is never checked, voxel_data_structures.cpp:99; the starve
  // step decrements weights after the check; imported blocks) and are then outside the image on the next frame.  Those
  // frames, and the starve frames themselves, go through the general kernels.
  const bool starve_now = max_num_frames > 0 && c->frames > 0 && c->frames % (uint64_t) max_num_frames == 0;
  c->frame_fused_mr = t.multi_res && c->mr_fused && !c->profile && max_num_frames > 0 && !starve_now && !c->mr_next_general &&
                      c->frames >= 2 && !c->spherical;
  c->frame_general = t.multi_res && !c->frame_fused_mr;  // (round 4: single-resolution maps take the two launches under the spherical model, too)
  if (t.multi_res && !c->frame_fused_mr) {
    c->mr_summaries_valid = false;
    c->mr_next_general = starve_now || c->frames == 0;
    c->refill_flag_valid = false;
  }
  const bool pipe_frame = (c->pipe || c->spherical) && !c->frame_general && !t.multi_res;  // integrate_lazy also carries the spherical model's serial frames
  if (max_num_frames > 0 && starve_fused_ok(c) && c->zfused_n < (size_t) k.rows * k.cols) {
    // the z-buffers of the starve frames, both pairs empty, while the context is still allocating (not inside its first starve frame)
    const size_t npix = (size_t) k.rows * k.cols;
    HIP_TRY(c, hipStreamSynchronize(s));
    if (c->d_zfused) HIP_TRY(c, hipFree(c->d_zfused));
    c->d_zfused = nullptr; c->zfused_n = 0;
    HIP_TRY(c, hipMalloc((void**) &c->d_zfused, 4 * npix * sizeof(u64)));
    c->zfused_n = npix;
    k_fill_u64<<<512, 256, 0, s>>>(c->d_zfused, 4 * npix, 0x7FFFFFFFFFFFFFFFull);
    c->zfused_clean[0] = c->zfused_clean[1] = true;
    c->zfused_clean_npix = npix;
  }
  if (!pipe_frame) {  // a frame of another kind follows pipelined ones
    rc = strict_point(c);
    if (rc) return rc;
  }
  if (pipe_frame) return integrate_lazy(c, max_num_frames, starve_now);
  rc = send_uploads(c, s);  // the frame's kernels read the images on the main stream
  if (rc) return rc;
  if (!c->frame_general) {
    // ---- fast path: alloc + sweep -> fused integrate / summary / GC (mrh_fast2.h)
    if (!t.multi_res && c->fast_summaries_stale) {  // a general frame (spherical camera) ran since: rebuild the GC summaries once
      k_summarize_all<<<2048, 256, 0, s>>>(t, c->fast);
      c->fast_summaries_stale = false;
    }
    const size_t npix = (size_t) k.rows * k.cols;
    const int tiles_x = (k.cols + kRayTile - 1) / kRayTile, tiles_y = (k.rows + kRayTile - 1) / kRayTile;
    if (c->fast_npix < npix) {
      HIP_TRY(c, hipStreamSynchronize(s));
      if (c->dcx_buf) HIP_TRY(c, hipFree(c->dcx_buf));
      c->dcx_buf = nullptr;
      HIP_TRY(c, hipMalloc((void**) &c->dcx_buf, npix * sizeof(uint2)));
      c->fast_npix = npix;
    }
    c->fast.dcx = c->dcx_buf;
    const Fast& f = c->fast;
    const int parity = (int) (c->fast_frames % kListSets), zero_set = (parity + kListSets - 1) % kListSets;  // the frame's list-counter set, and the one to clear
    c->frame_parity = parity;
    c->fast_frames++;
    const u32 stamp = (u32) ((c->frames + 1) & 0x3FFFFFFFu);
    const float gc_thr = m.trunc + m.trunc_scale * k.max_depth;  // getTruncation(camera.maxDepth(), ...), vds.cu:1720
    const Lists L = {t.compact, c->fast.bbox, c->d_cfree, c->d_zmin, (u32) c->num_blocks};
    const bool safe_div = m.half_vs_two_steps || m.wsum_two_steps;  // the short divisions failed their check at mrh_create
    const int n_tiles = tiles_x * tiles_y;
    const size_t lds = (size_t) 4 * kTileMaxPx * sizeof(uint2);
    if (c->frame_fused_mr) {
      if (!c->mr_summaries_valid) {
        k_summarize_all<<<2048, 256, 0, s>>>(t, f);
        c->mr_summaries_valid = true;
      }
      c->frame_gc_inline = true;
      // the coarse-list refill (vds.cu:885-891) rides in k_front; its test was taken by the previous frame's k_mr_tail
      // unless something else touched the coarse list since (general frames, import, stream-out, reset)
      if (!c->refill_flag_valid) k_refill_decide<<<1, 64, 0, s>>>(t, c->low_blocks_to_allocate, c->d_flag);
      const int n_refill = (c->low_blocks_to_allocate + 255) / 256;
      k_front<false, true><<<n_tiles + c->sweep_wgs_mr + n_refill, 256, 0, s>>>(k, m, t, f, L, c->d_depth, c->d_rgb, tiles_x, n_tiles, stamp, parity, 1, gc_thr,
                                                                                   n_refill, c->low_blocks_to_allocate, c->d_flag);
      if (safe_div) k_back<true, false, true, true><<<c->fused_grid, 256, lds, s>>>(k, m, t, f, L, parity, zero_set, gc_thr, c->d_depth, c->d_rgb, (u32*) c->d_reint, 0u, 0);
      else k_back<true, false, true, false><<<c->fused_grid, 256, lds, s>>>(k, m, t, f, L, parity, zero_set, gc_thr, c->d_depth, c->d_rgb, (u32*) c->d_reint, 0u, 0);
      k_mr_tail<<<1, 256, 0, s>>>(t, (const u32*) c->d_reint, c->low_blocks_to_allocate, c->d_flag);
      rc = starve_and_tail(c, max_num_frames);
      c->refill_flag_valid = rc == MRH_OK;
      return rc;
    }
    // GC runs inside k_back unless this is a starve frame (the starve step changes weights after the integrate pass)
    c->frame_gc_inline = max_num_frames > 0 && !starve_now;
    auto take_events = [&](EvPair& e) -> int {
      if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); return MRH_OK; }
      if (c->ev_pending.size() >= 4096) {
        HIP_TRY(c, hipStreamSynchronize(s));
        const int r = drain_events(c);
        if (r) return r;
        e = c->ev_pool.back(); c->ev_pool.pop_back();
        return MRH_OK;
      }
      HIP_TRY(c, hipEventCreate(&e.a)); HIP_TRY(c, hipEventCreate(&e.b));
      return MRH_OK;
    };
    if (c->profile) {  // event pair attached to the launch, as for k_back below
      EvPair evf;
      rc = take_events(evf);
      if (rc) return rc;
      hipExtLaunchKernelGGL((k_front<true, false>), dim3(n_tiles + c->sweep_wgs), dim3(256), 0, s, evf.a, evf.b, 0u, k, m, t, f, L, c->d_depth, c->d_rgb, tiles_x, n_tiles, stamp, parity,
                            max_num_frames > 0 ? 1 : 0, gc_thr, 0, 0, (const int*) nullptr);
      c->ev_pending_front.push_back(evf);
    }
    else k_front<false, false><<<n_tiles + c->sweep_wgs, 256, 0, s>>>(k, m, t, f, L, c->d_depth, c->d_rgb, tiles_x, n_tiles, stamp, parity, max_num_frames > 0 ? 1 : 0, gc_thr, 0, 0, nullptr);
    EvPair ev;
    if (c->profile) {
      k_count_updates<<<c->fused_grid, 256, 0, s>>>(k, m, t, f, c->d_cnt_partials, CTR_SET0 + 4 * parity, L.vis, L.cfree, 0u, 0);
      rc = take_events(ev);
      if (rc) return rc;
    }
    // Profile mode: the event pair is attached to the launch itself (hipExtLaunchKernelGGL), so it holds the kernel's own
    // begin / end timestamps — the duration rocprofv3 reports — instead of a hipEventRecord bracket, which adds the
    // dispatch latency of a dependent launch (~3.5 us here) to every sample.
#define MRH_K_BACK_V(FREE, PROF, SAFE)                                                                                                   \
  do {                                                                                                                                   \
    if (c->profile) hipExtLaunchKernelGGL((k_back<FREE, PROF, false, SAFE>), dim3(c->fused_grid), dim3(256), (uint32_t) lds, s, ev.a, ev.b, 0u, k, m, t, f, L,   \
                                          parity, zero_set, gc_thr, (const float*) nullptr, (const uint8_t*) nullptr, (u32*) nullptr, 0u, 0);  \
    else k_back<FREE, PROF, false, SAFE><<<c->fused_grid, 256, lds, s>>>(k, m, t, f, L, parity, zero_set, gc_thr, nullptr, nullptr, nullptr, 0u, 0); \
  } while (0)
#define MRH_K_BACK(FREE, PROF) do { if (safe_div) MRH_K_BACK_V(FREE, PROF, true); else MRH_K_BACK_V(FREE, PROF, false); } while (0)
    if (c->frame_gc_inline && c->profile) MRH_K_BACK(true, true);
    else if (c->frame_gc_inline) MRH_K_BACK(true, false);
    else MRH_K_BACK(false, false);
#undef MRH_K_BACK
#undef MRH_K_BACK_V
    if (c->profile) c->ev_pending.push_back(ev);
    if (starve_now && starve_fused_ok(c)) {
      rc = launch_starve_fused(c, k, f, L, parity, gc_thr, stamp, 0);
      if (rc) return rc;
      c->frames++;
      return MRH_OK;
    }
    return starve_and_tail(c, max_num_frames);
  }

  if (t.multi_res) {
    // vds.cu:885-891 (coarse free-list refill), decided on the device
    k_refill_decide<<<1, 64, 0, s>>>(t, c->low_blocks_to_allocate, c->d_flag);
    k_refill<<<(c->low_blocks_to_allocate + 255) / 256, 256, 0, s>>>(t, c->low_blocks_to_allocate, c->d_flag);
  }
  // the image every kernel below reads as "depth": the raw image (pinhole: cloud z == depth, cleaned on the fly) or, for
  // the spherical model, getDepth(cloud) computed once per frame
  const float* depth_img = c->d_depth;
  if (c->spherical) {
    const size_t npix = (size_t) k.rows * k.cols;
    if (c->cloud_n < npix) {
      HIP_TRY(c, hipStreamSynchronize(s));
      if (c->d_cloud) HIP_TRY(c, hipFree(c->d_cloud));
      c->d_cloud = nullptr;
      HIP_TRY(c, hipMalloc((void**) &c->d_cloud, npix * sizeof(float)));
      c->cloud_n = npix;
    }
    k_cloud_depth<<<(int) ((npix + 255) / 256), 256, 0, s>>>(k, c->d_depth, c->d_cloud);
    depth_img = c->d_cloud;
  }
  const dim3 tiles((k.cols + kTile - 1) / kTile, (k.rows + kTile - 1) / kTile);
  if (c->profile) k_alloc<true><<<tiles, dim3(kTile, kTile), 0, s>>>(k, m, t, depth_img);
  else k_alloc<false><<<tiles, dim3(kTile, kTile), 0, s>>>(k, m, t, depth_img);
  k_compact<<<512, 256, 0, s>>>(k, m, t, 1);

  if (c->profile) {
    EvPair ev;
    if (!c->ev_pool.empty()) { ev = c->ev_pool.back(); c->ev_pool.pop_back(); }
    else {
      if (c->ev_pending.size() >= 4096) { HIP_TRY(c, hipStreamSynchronize(s)); rc = drain_events(c); if (rc) return rc; ev = c->ev_pool.back(); c->ev_pool.pop_back(); }
      else { HIP_TRY(c, hipEventCreate(&ev.a)); HIP_TRY(c, hipEventCreate(&ev.b)); }
    }
    HIP_TRY(c, hipEventRecord(ev.a, s));
    k_integrate<true><<<c->integrate_grid, 512, 0, s>>>(k, m, t, depth_img, c->d_rgb, c->d_upd_partials);
    HIP_TRY(c, hipEventRecord(ev.b, s));
    c->ev_pending.push_back(ev);
  } else {
    k_integrate<false><<<c->integrate_grid, 512, 0, s>>>(k, m, t, depth_img, c->d_rgb, c->d_upd_partials);
  }

  if (t.multi_res && c->frames > 0) {
    // checkVarSDF -> reallocBlocks -> flatAndReduceHashTable(camera) -> reintegrateDepthMap
    HIP_TRY(c, hipMemsetAsync(&t.ctr[CTR_NREALLOC], 0, 2 * sizeof(int), s));  // NREALLOC, NREINT
    k_check_var<<<2048, 64, 0, s>>>(m, t, c->d_realloc);
    k_realloc<<<64, 256, 0, s>>>(t, c->d_realloc, c->d_reint);
    k_compact<<<512, 256, 0, s>>>(k, m, t, 1);
    k_reintegrate<<<1024, 64, 0, s>>>(k, m, t, depth_img, c->d_rgb, c->d_reint);
  }

  return starve_and_tail(c, max_num_frames);
}

}  // extern "C"

namespace {
// Is this cloud an organised scan — rows of L points each, row-major, neighbours in the array neighbours in direction both along a
// row and from one row to the next?  A few dozen point pairs decide: the candidate L (a power of two that leaves a multiple of 16
// rows) whose points i and i + L lie closest in direction, if that and the step to i + 1 are within a few degrees.  Only a hint
// for the order in which k_scan_walk takes the beams (mrh_scan.h: Scan::patch_log2): a wrong answer costs time, never a bit.
int detect_scan_row_len(const float* xyz, const uint64_t n) {
  if (n < 4096 || n % 256) return 0;
  auto cos_between = [&](uint64_t a, uint64_t b, double* out) {
    const float *p = xyz + 3 * a, *q = xyz + 3 * b;
    const double pp = (double) p[0] * p[0] + (double) p[1] * p[1] + (double) p[2] * p[2], qq = (double) q[0] * q[0] + (double) q[1] * q[1] + (double) q[2] * q[2];
    if (!(pp > 0.0) || !(qq > 0.0)) return false;  // a missing return
    *out = ((double) p[0] * q[0] + (double) p[1] * q[1] + (double) p[2] * q[2]) / std::sqrt(pp * qq);
    return true;
  };
  constexpr int kSamples = 96;
  const double cos_limit = 0.99756;  // 4 degrees (a 16-beam sensor's rows are 2-3 degrees apart)
  int best = 0;
  double best_cos = cos_limit;
  for (uint64_t L = 16; L <= 8192 && L * 16 <= n; L <<= 1) {
    if (n % L || (n / L) % 16) continue;
    double sum_row = 0.0, sum_next = 0.0;
    int ok = 0;
    for (int k = 0; k < kSamples; k++) {
      const uint64_t i = (uint64_t) ((double) k * (double) (n - L - 2) / kSamples);
      double a, b;
      if ((i % L) + 1 < L && cos_between(i, i + 1, &a) && cos_between(i, i + L, &b)) { sum_next += a; sum_row += b; ok++; }
    }
    if (ok < kSamples / 4) continue;
    if (sum_next / ok > cos_limit && sum_row / ok > best_cos) { best_cos = sum_row / ok; best = (int) L; }
  }
  return best;
}
}  // namespace

extern "C" {

int mrh_detect_scan_layout(const float* xyz, uint64_t n) { return xyz ? detect_scan_row_len(xyz, n) : 0; }

int mrh_set_scan_layout(mrh_ctx* c, int row_len) {
  if (!c) return MRH_ERR_INVALID_ARG;
  c->scan_layout_hint = row_len;
  c->scan_row_len = row_len > 0 ? row_len : 0;
  c->scan_detect_n = 0;
  return MRH_OK;
}

int mrh_upload_points(mrh_ctx* c, const float* xyz, uint64_t n) {
  int rc = ensure_ready(c, "mrh_upload_points");
  if (rc) return rc;
  if (n && !xyz) return fail(c, MRH_ERR_INVALID_ARG, "mrh_upload_points: null argument");
  if (n > c->points_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_points) HIP_TRY(c, hipFree(c->d_points));
    c->d_points = nullptr;
    HIP_TRY(c, hipMalloc((void**) &c->d_points, n * 3 * sizeof(float)));
    c->points_cap = n;
  }
  if (n) HIP_TRY(c, hipMemcpyAsync(c->d_points, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller's buffer is free on return (GeoWrapper::setPointCloud copies)
  c->d_points_cur = c->d_points;
  c->num_points = n;
  if (c->scan_layout_hint > 0) c->scan_row_len = c->scan_layout_hint;
  else if (c->scan_layout_hint == 0 && n) {
    if (n != c->scan_detect_n || ++c->scan_detect_age >= 64) {
      c->scan_detect_len = detect_scan_row_len(xyz, n);
      c->scan_detect_n = n;
      c->scan_detect_age = 0;
    }
    c->scan_row_len = c->scan_detect_len;
  } else c->scan_row_len = 0;
  return MRH_OK;
}

int mrh_set_points_device(mrh_ctx* c, const float* d_xyz, uint64_t n) {
  int rc = ensure_ready(c, "mrh_set_points_device");
  if (rc) return rc;
  if (n && !d_xyz) return fail(c, MRH_ERR_INVALID_ARG, "mrh_set_points_device: null argument");
  c->d_points_cur = d_xyz;
  c->num_points = n;
  c->scan_row_len = c->scan_layout_hint > 0 ? c->scan_layout_hint : 0;  // a cloud in device memory is not looked at: mrh_set_scan_layout says how it is laid out
  return MRH_OK;
}

int mrh_upload_normals(mrh_ctx* c, const float* nxyz, uint64_t n) {
  int rc = ensure_ready(c, "mrh_upload_normals");
  if (rc) return rc;
  if (n && !nxyz) return fail(c, MRH_ERR_INVALID_ARG, "mrh_upload_normals: null argument");
  if (n > c->normals_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_normals) HIP_TRY(c, hipFree(c->d_normals));
    c->d_normals = nullptr;
    HIP_TRY(c, hipMalloc((void**) &c->d_normals, n * 3 * sizeof(float)));
    c->normals_cap = n;
  }
  if (n) HIP_TRY(c, hipMemcpyAsync(c->d_normals, nxyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller's buffer is free on return
  c->num_normals = n;
  return MRH_OK;
}

}  // extern "C"

namespace {
// stable radix sort of the scan's (voxel id, sdf) records on key bits [0, end_bit): the padding key (all ones) ends up last,
// equal ids keep their point-major order (mrh_sort.h)
// *out_buf = which of the two buffer pairs holds the sorted records
template <typename K>
int lidar_sort(mrh_ctx* c, K* k0, K* k1, float* v0, float* v1, const size_t n, const int end_bit, int* out_buf) {
  const u32 ntiles = (u32) ((n + kSortTile - 1) / kSortTile);
  const u32 total = 256u * ntiles;
  if (n > 0xFFFFFFFFull - kSortTile || (uint64_t) 256u * ntiles > kSortScanMax)  // records and histogram entries are indexed in 32 bits
    return fail(c, MRH_ERR_CAPACITY, "mrh_integrate_points: %zu records in one scan through the sorted path (limit 2^32 - %d)", n, kSortTile + 1);
  // the scan-sized sort of mrh_sort.h: per 8-bit digit a tile histogram, a one-workgroup scan, a stable scatter
  hipStream_t s = c->stream;
  if ((size_t) total * sizeof(u32) + 1024 > c->sort_tmp_bytes) {
    HIP_TRY(c, hipStreamSynchronize(s));
    if (c->d_sort_tmp) HIP_TRY(c, hipFree(c->d_sort_tmp));
    c->d_sort_tmp = nullptr;
    HIP_TRY(c, hipMalloc(&c->d_sort_tmp, (size_t) total * sizeof(u32) * 2 + 1024));
    c->sort_tmp_bytes = (size_t) total * sizeof(u32) * 2 + 1024;
  }
  u32* hist = (u32*) c->d_sort_tmp + 256;  // [0, 256): the digit totals
  K* ks[2] = {k0, k1};
  float* vs[2] = {v0, v1};
  int src = 0;
  for (int shift = 0; shift < end_bit; shift += 8) {
    k_sort_hist<K><<<ntiles, kSortThreads, 0, s>>>(ks[src], (u32) n, shift, hist, ntiles);
    k_sort_scan<<<256, 256, 0, s>>>(hist, ntiles, (u32*) c->d_sort_tmp);
    k_sort_scatter<K><<<ntiles, kSortThreads, 0, s>>>(ks[src], vs[src], ks[src ^ 1], vs[src ^ 1], (u32) n, shift, hist, ntiles, (const u32*) c->d_sort_tmp);
    src ^= 1;
  }
  HIP_TRY(c, hipGetLastError());
  *out_buf = src;
  return MRH_OK;
}
}  // namespace

extern "C" {

// Scratch and buffers of the voxel-bucket scans (mrh_scan.h).  0: ready, 1: not on this context (the sorted path takes over), < 0: error.
static int scan_prepare(mrh_ctx* c, const uint64_t n, const uint64_t rec_bound, const size_t walk_lds) {
  hipStream_t s = c->stream;
  Scan& sc = c->scan;
  if (c->scan_state == 0) {
    const size_t nb = (size_t) c->num_blocks;
    bool ok = hipMalloc((void**) &sc.vcnt, nb * 512 * sizeof(u32)) == hipSuccess;
    ok = ok && hipMalloc((void**) &sc.bstamp, nb * sizeof(u32)) == hipSuccess;
    ok = ok && hipMalloc((void**) &c->d_scan_ctr, 2 * SC_N * sizeof(u32)) == hipSuccess;
    if (!ok) {  // one counter per voxel slot does not fit next to this map
      (void) hipGetLastError();
      auto F = [](void* p) { if (p) (void) hipFree(p); };
      F(sc.vcnt); F(sc.bstamp); F(c->d_scan_ctr);
      sc.vcnt = sc.bstamp = c->d_scan_ctr = nullptr;
      c->scan_state = -1;
      return 1;
    }
    HIP_TRY(c, hipMemsetAsync(sc.vcnt, 0, nb * 512 * sizeof(u32), s));
    HIP_TRY(c, hipMemsetAsync(sc.bstamp, 0, nb * sizeof(u32), s));
    HIP_TRY(c, hipMemsetAsync(c->d_scan_ctr, 0, 2 * SC_N * sizeof(u32), s));
    c->scan_state = 1;
    c->scan_dirty = false;
  }
  if (c->scan_dirty) {  // a scan that failed half way leaves counters behind
    HIP_TRY(c, hipMemsetAsync(sc.vcnt, 0, (size_t) c->num_blocks * 512 * sizeof(u32), s));
    HIP_TRY(c, hipMemsetAsync(c->d_scan_ctr, 0, 2 * SC_N * sizeof(u32), s));
    c->scan_dirty = false;
  }
  if (rec_bound > c->scan_rec_cap) {
    HIP_TRY(c, hipStreamSynchronize(s));
    for (void* p : {(void*) sc.st_meta, (void*) sc.st_sdf, (void*) sc.st_grp, (void*) sc.rec, (void*) sc.chunks})
      if (p) HIP_TRY(c, hipFree(p));
    sc.st_meta = nullptr; sc.st_sdf = nullptr; sc.st_grp = nullptr; sc.rec = nullptr; sc.chunks = nullptr;
    c->scan_rec_cap = 0;
    const uint64_t cap = rec_bound;
    // chunks: one per touched block + one per kScanChunkWeight of weight (a record weighs at least 32: <= records / 256, taken as
    // records / 128) + two per run beyond kScanLongRun (the run's own chunk and the cut behind it: <= 2 * records / 65)
    const uint64_t chunk_cap = std::min<uint64_t>(c->num_blocks, cap) + cap / 128 + 2 * cap / (kScanLongRun + 1) + 64;
    HIP_TRY(c, hipMalloc((void**) &sc.st_meta, cap * sizeof(uint2)));
    HIP_TRY(c, hipMalloc((void**) &sc.st_sdf, cap * sizeof(float)));
    HIP_TRY(c, hipMalloc((void**) &sc.st_grp, cap * sizeof(uint2)));
    HIP_TRY(c, hipMalloc((void**) &sc.rec, cap * sizeof(uint4)));
    HIP_TRY(c, hipMalloc((void**) &sc.chunks, chunk_cap * sizeof(uint4)));
    sc.rec_cap = (u32) std::min<uint64_t>(cap, 0xFFFFFFF0ull);
    sc.chunk_cap = (u32) std::min<uint64_t>(chunk_cap, 0xFFFFFFF0ull);
    c->scan_rec_cap = cap;
  }
  const uint64_t wgs = (n + 255) / 256;
  if (wgs > c->scan_wg_cap) {
    HIP_TRY(c, hipStreamSynchronize(s));
    if (sc.wgdesc) HIP_TRY(c, hipFree(sc.wgdesc));
    sc.wgdesc = nullptr;
    HIP_TRY(c, hipMalloc((void**) &sc.wgdesc, wgs * sizeof(uint2)));
    c->scan_wg_cap = wgs;
  }
  if (walk_lds > 65536 && walk_lds > c->scan_lds_set) {
    HIP_TRY(c, hipFuncSetAttribute((const void*) k_scan_walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int) walk_lds));
    c->scan_lds_set = walk_lds;
  }
  return 0;
}

// VoxelContainer::integrate(point_cloud, ...) voxel_data_structures.cpp:112-135 (mrh_lidar.h)
int mrh_integrate_points(mrh_ctx* c, int n_frames_invalidate) {
  int rc = ensure_ready(c, "mrh_integrate_points");
  if (rc) return rc;
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_integrate_points: an exchange is pending (call mrh_integrate_resume)");
  if (c->halo_upper) return fail(c, MRH_ERR_STATE, "mrh_integrate_points: halo blocks of other shards are present (call mrh_drop_blocks(MRH_DROP_HALO) after the extraction)");
  if (!c->has_camera) return fail(c, MRH_ERR_STATE, "mrh_integrate_points: set_camera has not been called");
  const int max_num_frames = n_frames_invalidate < 0 ? c->p.n_frames_invalidate_voxels : n_frames_invalidate;
  const uint64_t n = c->num_points;
  if (!c->p.projective_sdf && c->num_normals != n)
    return fail(c, MRH_ERR_STATE, "mrh_integrate_points: the normal-direction SDF needs one normal per point (mrh_upload_normals)");
  if (n >= (1ull << 24)) return fail(c, MRH_ERR_CAPACITY, "mrh_integrate_points: %llu points in one scan (limit 2^24 - 1)", (unsigned long long) n);
  hipStream_t s = c->stream;
  const Cam& k = c->cam;
  const Map& m = c->map;
  const Tab& t = c->tab;
  // the table upkeep rebuilds from the descriptors: the zombies of the pipelined frames leave first
  if (c->zombies_possible && c->census_period >= 0 && (c->table_dirty || c->frames_since_census >= (uint64_t) c->census_period)) {
    rc = strict_point(c);
    if (rc) return rc;
  }
  rc = maintain_table(c, false);
  if (rc) return rc;
  c->frames_since_census++;
  c->frame_general = true;  // GC (and the starve step) of a scan run through the general kernels on the list of ALL live blocks
  c->frame_fused_mr = false;
  if (t.multi_res) { c->mr_summaries_valid = false; c->mr_next_general = true; c->refill_flag_valid = false; }
  const float* normals = c->p.projective_sdf ? nullptr : c->d_normals;
  if (n > 0) {
    const u32 np = (u32) n, grid = (np + 255) / 256;
    const float* pts = c->d_points_cur;
    const u32 stamp = (u32) ((c->frames + 1) & 0x3FFFFFFFu);
    if (t.multi_res) {  // vds.cu:1048-1054: coarse free-list refill, decided on the device
      k_refill_decide<<<1, 64, 0, s>>>(t, c->low_blocks_to_allocate, c->d_flag);
      k_refill<<<(c->low_blocks_to_allocate + 255) / 256, 256, 0, s>>>(t, c->low_blocks_to_allocate, c->d_flag);
    }
    // an organised scan is taken in 2-D patches of beams (mrh_lidar.h: BeamOrder); its row length: mrh_set_scan_layout, the look
    // at a host cloud (mrh_upload_points), or the spherical camera's columns if it has one pixel per point
    BeamOrder order;
    order.patch_log2 = 8; order.patches_per_row = 1; order.row_len = 256;
    {
      const int want = c->scan_patch_log2;
      const uint64_t row_len = c->scan_row_len > 0 ? (uint64_t) c->scan_row_len : (c->scan_layout_hint == 0 && (uint64_t) k.rows * (uint64_t) k.cols == n ? (uint64_t) k.cols : 0);
      if (want < 8 && row_len > 0 && n % row_len == 0 && row_len % (1u << want) == 0 && (n / row_len) % (256u >> want) == 0) {
        order.patch_log2 = (u32) want; order.patches_per_row = (u32) (row_len >> want); order.row_len = (u32) row_len;
      }
    }
    k_alloc3d<<<grid, 256, 0, s>>>(k, m, t, c->fast, pts, normals, np, stamp, order);
    // ---- integrate3D (vds.cu:1215-1410): records of every (point, voxel) in point-major order -> stable sort by voxel -> fold.
    // The record buffers are sized by a bound the host can compute (a beam crosses at most `slots` voxels), so the emit pass
    // needs nothing from the host and runs WHILE the host picks up the scan's one report (record count, high-water mark: they
    // size the sort): count -> scan -> report -> emit are enqueued together, the sort and the fold follow the report.
    // slots: the voxel-level DDA walks from voxel(p_min) to voxel(p_max), |p_max - p_min| <= 2 tr, tr <= trunc + scale *
    // integration distance: at most sum_axis(|end - start|) + 1 steps <= (2 tr / vs) * sqrt(3) + 3, plus the roundings
    const double tr_max = (double) m.trunc + (double) m.trunc_scale * (double) k.max_int_dist;
    const uint64_t slots = std::min<uint64_t>(kMaxDdaIter, (uint64_t) std::floor(2.0 * tr_max / (double) m.vs * 1.7320508) + 10);
    const uint64_t rec_bound = n * slots;
    if (rec_bound >= 0xFFFFFFF0ull) return fail(c, MRH_ERR_CAPACITY, "mrh_integrate_points: %llu points x %llu voxels per beam exceed 2^32 records per scan", (unsigned long long) n, (unsigned long long) slots);
    // key width from the pool capacity: voxel id < cap * 512, one more bit for coarse units
    auto bits_for = [](uint64_t max_value) { int b = 1; while (b < 63 && (max_value >> b)) b++; return b; };
    const int coarse_bit = bits_for((uint64_t) c->num_blocks * 512 - 1);
    const bool wide = coarse_bit + (t.multi_res ? 1 : 0) > 32;
    const size_t key_bytes = wide ? 8 : 4;
    // voxel buckets (mrh_scan.h) unless the map's voxel ids, the beam length or the memory say otherwise: then the sorted records below
    bool buckets = c->lidar_buckets && c->scan_state >= 0 && !wide && (uint64_t) c->num_blocks * 512 < 0x7FFFFE00ull &&
                   slots <= (uint64_t) kScanMaxSlots;
    if (buckets) {
      rc = scan_prepare(c, n, rec_bound, (size_t) (2 * slots * 256 + 2 * kScanSetSize) * sizeof(u32));
      if (rc < 0) return rc;
      buckets = rc == 0;
    }
    auto integrate_scan_buckets = [&]() -> int {
      Scan sc = c->scan;
      sc.seq = ++c->scan2_seq;
      if (sc.seq >= 0x80000000u) {  // a block's stamp is seq * 2 + coarse in 32 bits: the sequence restarts at 1 with clean stamps
        HIP_TRY(c, hipMemsetAsync(sc.bstamp, 0, (size_t) c->num_blocks * sizeof(u32), s));
        // ... and with both counter sets at zero: the restart breaks the alternation that lets a scan zero the next one's set
        HIP_TRY(c, hipMemsetAsync(c->d_scan_ctr, 0, 2 * SC_N * sizeof(u32), s));
        sc.seq = c->scan2_seq = 1;
      }
      sc.ctr = c->d_scan_ctr + (sc.seq & 1u) * SC_N;
      sc.ctr_next = c->d_scan_ctr + ((sc.seq + 1u) & 1u) * SC_N;
      sc.ord_shift = t.multi_res ? 5 : 0;
      sc.narrow = ((uint64_t) np << sc.ord_shift) <= (1ull << 23) && !getenv("MRH_SCAN_WIDE_RECORDS") ? 1 : 0;  // tag + 9 bits of voxel index in one word (MRH_SCAN_WIDE_RECORDS=1: tests)
      sc.order = order;
      const size_t lds = (size_t) (2 * slots * 256 + 2 * kScanSetSize) * sizeof(u32);
      k_scan_walk<<<grid, 256, lds, s>>>(k, m, t, pts, normals, np, sc, (int) slots);
      // the touched blocks are found by their stamps inside k_scan_offsets, windows of kScanWindow blocks; 512 workgroups walk the
      // windows (2 048 of these 1 024-thread workgroups took 9 us to DISPATCH for ~1 us of work each, tools/trace_scan.py)
      k_scan_offsets<<<std::min<u32>(512u, (u32) ((c->num_blocks + kScanWindow - 1) / kScanWindow)), 1024, 0, s>>>(t, sc, t.multi_res ? (u32) c->num_blocks : 0u);
      k_scan_place<<<grid, 256, 0, s>>>(sc, (int) slots);
      k_scan_apply<<<1536, 256, 0, s>>>(m, t, sc, np << sc.ord_shift, c->profile);
      HIP_TRY(c, hipGetLastError());
      return MRH_OK;
    };
    if (!buckets && (rec_bound > c->rec_cap || key_bytes > c->rec_key_bytes)) {
      HIP_TRY(c, hipStreamSynchronize(s));
      for (int b = 0; b < 2; b++) {
        if (c->d_rec_keys[b]) HIP_TRY(c, hipFree(c->d_rec_keys[b]));
        if (c->d_rec_vals[b]) HIP_TRY(c, hipFree(c->d_rec_vals[b]));
        c->d_rec_keys[b] = nullptr; c->d_rec_vals[b] = nullptr;
      }
      const uint64_t cap = std::max<uint64_t>(rec_bound, c->rec_cap);
      for (int b = 0; b < 2; b++) {
        HIP_TRY(c, hipMalloc((void**) &c->d_rec_keys[b], cap * std::max(key_bytes, c->rec_key_bytes)));
        HIP_TRY(c, hipMalloc((void**) &c->d_rec_vals[b], cap * sizeof(float)));
      }
      c->rec_cap = cap;
      c->rec_key_bytes = std::max(key_bytes, c->rec_key_bytes);
    }
    if (!buckets && n > c->pt_cap) {
      HIP_TRY(c, hipStreamSynchronize(s));
      if (c->d_pt_counts) HIP_TRY(c, hipFree(c->d_pt_counts));
      if (c->d_pt_offsets) HIP_TRY(c, hipFree(c->d_pt_offsets));
      c->d_pt_counts = c->d_pt_offsets = nullptr;
      HIP_TRY(c, hipMalloc((void**) &c->d_pt_counts, n * sizeof(u32)));
      HIP_TRY(c, hipMalloc((void**) &c->d_pt_offsets, (n / 256 + 2) * sizeof(u32)));  // one total per count workgroup
      c->pt_cap = n;
    }
    if (!c->h_scan) {
      HIP_TRY(c, hipHostMalloc((void**) &c->h_scan, 4 * sizeof(u32), hipHostMallocDefault));
      memset(c->h_scan, 0, 4 * sizeof(u32));
    }
    auto integrate_scan = [&]() -> int {
      // the one host round trip of a scan: the emit pass derives its offsets from the per-workgroup totals itself, and its LAST
      // workgroup, which knows the grand total before its walk starts, writes {high-water mark, records} and a sequence mark into
      // pinned memory: the host reads it and enqueues the sort while the emit pass runs
      ScanState ss;
      ss.wg_totals = c->d_pt_offsets; ss.host_rec = c->h_scan; ss.seq = ++c->scan_seq;
      const u32 seq = ss.seq;
      k_points_walk<false, u32><<<grid, 256, 0, s>>>(k, m, t, pts, normals, np, c->d_pt_counts, ss, (u32*) nullptr, nullptr, coarse_bit, 0u);
      if (wide) k_points_walk<true, u64><<<grid, 256, 0, s>>>(k, m, t, pts, normals, np, c->d_pt_counts, ss, (u64*) c->d_rec_keys[0], c->d_rec_vals[0], coarse_bit, (u32) std::min<uint64_t>(c->rec_cap, 0xFFFFFFFFull));
      else k_points_walk<true, u32><<<grid, 256, 0, s>>>(k, m, t, pts, normals, np, c->d_pt_counts, ss, (u32*) c->d_rec_keys[0], c->d_rec_vals[0], coarse_bit, (u32) std::min<uint64_t>(c->rec_cap, 0xFFFFFFFFull));
      HIP_TRY(c, hipGetLastError());
      {
        volatile u32* mark = c->h_scan + 3;
        const auto t0 = std::chrono::steady_clock::now();
        int spins = 0;
        while (*mark != seq) {
          MRH_CPU_RELAX();
          if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;  // long scan or a fault
        }
        if (*mark != seq) HIP_TRY(c, hipStreamSynchronize(s));  // reports a device error if that is why the mark never came
        if (*mark != seq) return fail(c, MRH_ERR_DEVICE, "mrh_integrate_points: the scan report did not arrive");
        std::atomic_thread_fence(std::memory_order_acquire);
      }
      const int hwm = (int) c->h_scan[0];
      const uint64_t n_rec = (uint64_t) c->h_scan[1];
      if (n_rec > rec_bound) return fail(c, MRH_ERR_DEVICE, "mrh_integrate_points: %llu records exceed the bound of %llu", (unsigned long long) n_rec, (unsigned long long) rec_bound);
      if (n_rec == 0) return MRH_OK;
      // the sort only looks at the bits a voxel id of THIS map can have (the high-water mark of the pool); it is stable, and the
      // records were emitted in point order: every voxel's run ends up in ascending point index (D6)
      const int end_bit = t.multi_res ? coarse_bit + 1 : bits_for((uint64_t) (hwm > 0 ? hwm : 1) * 512 - 1);
      const u32 agrid = (u32) ((n_rec + kApplyChunk - 1) / kApplyChunk);
      int r, sb = 1;
      if (wide) {
        r = lidar_sort(c, (u64*) c->d_rec_keys[0], (u64*) c->d_rec_keys[1], c->d_rec_vals[0], c->d_rec_vals[1], (size_t) n_rec, end_bit, &sb);
        if (r) return r;
        k_points_apply<u64><<<agrid, 256, 0, s>>>(m, t, (const u64*) c->d_rec_keys[sb], c->d_rec_vals[sb], (u32) n_rec, coarse_bit, c->profile);
      } else {
        r = lidar_sort(c, (u32*) c->d_rec_keys[0], (u32*) c->d_rec_keys[1], c->d_rec_vals[0], c->d_rec_vals[1], (size_t) n_rec, end_bit, &sb);
        if (r) return r;
        k_points_apply<u32><<<agrid, 256, 0, s>>>(m, t, (const u32*) c->d_rec_keys[sb], c->d_rec_vals[sb], (u32) n_rec, coarse_bit, c->profile);
      }
      HIP_TRY(c, hipGetLastError());
      return MRH_OK;
    };
    rc = buckets ? integrate_scan_buckets() : integrate_scan();
    if (rc) return rc;
    if (t.multi_res && c->frames > 0) {
      // checkVarSDF -> reallocBlocks -> flatAndReduceHashTable() -> reintegrate3D, which launches integrate3DKernel again
      // (vds.cu:1561-1580): the whole scan a second time, into fine and coarse blocks alike
      HIP_TRY(c, hipMemsetAsync(&t.ctr[CTR_COMPACT], 0, sizeof(int), s));
      k_compact<<<512, 256, 0, s>>>(k, m, t, 0);
      HIP_TRY(c, hipMemsetAsync(&t.ctr[CTR_NREALLOC], 0, 2 * sizeof(int), s));  // NREALLOC, NREINT
      k_check_var<<<2048, 64, 0, s>>>(m, t, c->d_realloc);
      k_realloc<<<64, 256, 0, s>>>(t, c->d_realloc, c->d_reint);
      rc = buckets ? integrate_scan_buckets() : integrate_scan();
      if (rc) return rc;
    }
  }
  if (max_num_frames > 0) {  // flatAndReduceHashTable() without a camera: every live block (voxel_data_structures.cpp:121, :126)
    HIP_TRY(c, hipMemsetAsync(&t.ctr[CTR_COMPACT], 0, sizeof(int), s));
    k_compact<<<512, 256, 0, s>>>(k, m, t, 0);
  }
  rc = starve_and_tail(c, max_num_frames);  // garbageCollect(camera, max_num_frames); counts the frame
  if (rc < 0) return rc;
  HIP_TRY(c, hipGetLastError());
  if (c->peek_enabled) {
    const int mrc = mark_frame(c);  // pool-level report for mrh_peek_free_blocks
    if (mrc) return mrc;
  }
  return rc;
}

int mrh_integrate_resume(mrh_ctx* c) {
  int rc = ensure_ready(c, "mrh_integrate_resume");
  if (rc) return rc;
  if (c->pending == 1) {
    launch_starve(c, 1);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->pending = 2;
    return MRH_PENDING_EXCHANGE;
  }
  if (c->pending == 2) {
    launch_starve(c, 2);
    c->pending = 0;
    rc = frame_tail(c, true, c->pending_max_frames);
    if (rc < 0) return rc;
    const int mrc = mark_frame(c);  // the tail's kernels read the frame's images too
    return mrc ? mrc : rc;
  }
  return fail(c, MRH_ERR_STATE, "mrh_integrate_resume: no exchange is pending");
}

int mrh_exchange_buffer(mrh_ctx* c, void** out_ptr, uint64_t* out_n, int* out_is_device) {
  if (!c || !out_ptr || !out_n) return MRH_ERR_INVALID_ARG;
  if (c->pending == 0) return fail(c, MRH_ERR_STATE, "mrh_exchange_buffer: no exchange is pending");
  const size_t npix = (size_t) c->cam.rows * c->cam.cols;
  *out_ptr = c->pending == 1 ? (void*) c->d_zbuf : (void*) (c->d_zbuf + npix);
  *out_n = npix;
  if (out_is_device) *out_is_device = 1;
  return MRH_OK;
}

int mrh_sync(mrh_ctx* c) {
  int rc = ensure_ready(c, "mrh_sync");
  if (rc) return rc;
  u32 flags = 0;
  rc = take_device_flags(c, &flags);
  if (rc) return rc;
  HIP_TRY(c, hipGetLastError());
  rc = drain_events(c);
  if (rc) return rc;
  flags |= c->flags_deferred;
  c->flags_deferred = 0;
  c->flags_peeked = 0;
  return check_device_flags(c, flags);
}

// GaussianContainer::extractNodesQTree + checkNodes (gaussian_data_structures.cpp:48-68, .cu:58-84), see mrh_splat.h
int mrh_splat_seeds(mrh_ctx* c, float qtree_thresh, int qtree_min_pixel_size, const mrh_splat_seed** out, uint64_t* out_n) {
  int rc = ensure_ready(c, "mrh_splat_seeds");
  if (rc) return rc;
  if (!out || !out_n) return fail(c, MRH_ERR_INVALID_ARG, "mrh_splat_seeds: null argument");
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_splat_seeds: an exchange is pending (call mrh_integrate_resume)");
  if (!c->has_camera) return fail(c, MRH_ERR_STATE, "mrh_splat_seeds: set_camera has not been called");
  if (c->spherical) return fail(c, MRH_ERR_UNSUPPORTED, "mrh_splat_seeds: pinhole camera only");
  if (qtree_min_pixel_size < 0 || qtree_thresh != qtree_thresh) return fail(c, MRH_ERR_INVALID_ARG, "mrh_splat_seeds: bad quad-tree parameter");
  if (!c->d_depth || !c->d_rgb) return fail(c, MRH_ERR_STATE, "mrh_splat_seeds: no depth / colour image");
  const Cam& k = c->cam;
  if (c->depth_rows != k.rows || c->depth_cols != k.cols || c->rgb_rows != k.rows || c->rgb_cols != k.cols)
    return fail(c, MRH_ERR_INVALID_ARG, "mrh_splat_seeds: image shape differs from the camera");
  if ((uint64_t) k.rows * (uint64_t) k.cols > (1ull << 22)) return fail(c, MRH_ERR_CAPACITY, "mrh_splat_seeds: image above 2^22 pixels");
  hipStream_t s = c->stream;
  // depth of the potential tree: the first level whose largest rectangle (the bottom-right chain of ceil halves) can
  // no longer split (quad_tree.cu:133-149)
  QTree qt = {k.cols, k.rows, 0, qtree_min_pixel_size, 0};
  for (int w = k.cols, h = k.rows; qt.D < kQtMaxDepth && !(w / 2 <= qt.min_px || h / 2 <= qt.min_px); qt.D++) { w -= w / 2; h -= h / 2; }
  qt.total = qt_level_offset(qt.D + 1);
  if (qt.total != c->qt.total || !c->d_qt_sums) {
    HIP_TRY(c, hipStreamSynchronize(s));
    auto F = [](auto*& p) { if (p) (void) hipFree(p); p = nullptr; };
    // only the quad-tree buffers are sized by qt.total: nothing else of the context is released here
    F(c->d_qt_sums); F(c->d_qt_flags); F(c->d_qt_unc); F(c->d_qt_marks); F(c->d_qt_pos); F(c->d_qt_parked); F(c->d_qt_leaves); F(c->d_qt_misc);
    if (c->h_qt_seeds) { (void) hipHostFree(c->h_qt_seeds); c->h_qt_seeds = nullptr; }
    const size_t n = qt.total;
    HIP_TRY(c, hipMalloc((void**) &c->d_qt_sums, n * sizeof(QSum)));
    HIP_TRY(c, hipMalloc((void**) &c->d_qt_flags, n * sizeof(u32)));
    HIP_TRY(c, hipMalloc((void**) &c->d_qt_unc, n * sizeof(u32)));
    HIP_TRY(c, hipMalloc((void**) &c->d_qt_marks, n * sizeof(u64)));
    HIP_TRY(c, hipMalloc((void**) &c->d_qt_pos, (n + (n + kChainTile - 1) / kChainTile + 1) * sizeof(u64)));  // + the tile sums of the marks' scan
    HIP_TRY(c, hipMalloc((void**) &c->d_qt_parked, n * sizeof(mrh_splat_seed)));
    c->qt_seed_cap = (u32) std::min<size_t>(n, (size_t) 1 << 20);  // seeds <= leaves <= 1 000 000 (checked below), or the call fails
    HIP_TRY(c, hipHostMalloc((void**) &c->h_qt_seeds, (size_t) c->qt_seed_cap * sizeof(mrh_splat_seed), hipHostMallocDefault));
    if (!c->h_qt_out) HIP_TRY(c, hipHostMalloc((void**) &c->h_qt_out, 2 * sizeof(u64), hipHostMallocDefault));
    HIP_TRY(c, hipMalloc((void**) &c->d_qt_leaves, n * sizeof(mrh_qtree_leaf)));
    HIP_TRY(c, hipMalloc((void**) &c->d_qt_misc, 2 * sizeof(u64)));
  }
  c->qt = qt;
  rc = send_uploads(c, c->stream);
  if (rc) return rc;
  const u32 grid = (qt.total + 255) / 256;
  u32* unc_count = (u32*) (c->d_qt_misc + 1);
  // exact statistics of every potential node, four tree levels per launch
  int L = qt.D, T = L < 4 ? L : 4;
  k_qt_sums_bottom<<<1u << (2 * (L - T)), 256, 0, s>>>(qt, c->d_rgb, c->d_qt_sums, T, c->d_qt_misc);
  for (L -= T; L > 0; L -= T) {
    T = L < 4 ? L : 4;
    k_qt_sums_up<<<1u << (2 * (L - T)), 256, 0, s>>>(qt, c->d_qt_sums, L, T);
  }
  // exclusive scan of the marks (mrh_sort.h): the tile sums (parked behind the positions) are cleared by k_qt_decide and added up by
  // k_qt_emit's workgroups, then every tile scans on its own
  const u32 tiles = (u32) ((qt.total + kChainTile - 1) / kChainTile);
  static_assert(kChainTile % 256 == 0, "a workgroup of k_qt_emit lies inside one scan tile");
  k_qt_decide<<<grid, 256, 0, s>>>(qt, qtree_thresh, c->d_qt_sums, c->qt_literal, c->d_qt_flags, c->d_qt_unc, unc_count, c->d_qt_pos + qt.total, tiles);
  k_qt_literal<<<512, 256, 0, s>>>(qt, c->d_rgb, qtree_thresh, c->d_qt_unc, unc_count, c->d_qt_flags);
  k_qt_emit<<<grid, 256, 0, s>>>(qt, c->cam, c->map, c->tab, c->d_depth, c->d_rgb, c->d_qt_flags, c->d_qt_marks, c->d_qt_parked, c->d_qt_pos + qt.total);
  k_tile_scan_u64<<<tiles, 1024, 0, s>>>(c->d_qt_marks, (u32) qt.total, c->d_qt_pos + qt.total, c->d_qt_pos);
  k_qt_scatter<<<grid, 256, 0, s>>>(qt, c->d_qt_marks, c->d_qt_pos, c->d_qt_parked, c->d_qt_leaves, c->h_qt_seeds, c->qt_seed_cap, c->d_qt_misc, c->h_qt_out);
  HIP_TRY(c, hipGetLastError());
  rc = mark_frame(c);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(s));
  const u64 h_misc[2] = {((volatile u64*) c->h_qt_out)[0], ((volatile u64*) c->h_qt_out)[1]};
  const uint64_t n_leaves = h_misc[0] & 0xFFFFFFFFull, n_seeds = h_misc[0] >> 32;
  c->qt_last_literal = (uint32_t) h_misc[1];
  if (n_leaves > 1000000ull) return fail(c, MRH_ERR_CAPACITY, "mrh_splat_seeds: %llu leaves, above the reference's capacity of 1000000 (params.h:20-23)", (unsigned long long) n_leaves);
  // the leaves (tens of thousands per frame) stay on the device until mrh_get_qtree_leaves asks: the fusion loop only takes the seeds
  c->qt_n_leaves = n_leaves;
  c->qt_leaves_on_host = n_leaves == 0;
  c->qt_leaves.clear();
  if (getenv("MRH_DEBUG")) {
    fprintf(stderr, "[mrh] splat seeds: %u potential nodes, %u literal evaluations, %llu leaves, %llu seeds\n", qt.total,
            c->qt_last_literal, (unsigned long long) n_leaves, (unsigned long long) n_seeds);
  }
  *out = c->h_qt_seeds;
  *out_n = n_seeds;
  return MRH_OK;
}

int mrh_get_qtree_leaves(mrh_ctx* c, const mrh_qtree_leaf** out, uint64_t* out_n) {
  if (!c || !out || !out_n) return MRH_ERR_INVALID_ARG;
  if (!c->qt_leaves_on_host) {
    c->qt_leaves.resize(c->qt_n_leaves);
    HIP_TRY(c, hipMemcpyAsync(c->qt_leaves.data(), c->d_qt_leaves, c->qt_n_leaves * sizeof(mrh_qtree_leaf), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->qt_leaves_on_host = true;
  }
  *out = c->qt_leaves.data();
  *out_n = c->qt_leaves.size();
  return MRH_OK;
}

int mrh_get_free_blocks(mrh_ctx* c, int64_t* out_free_fine, int64_t* out_free_coarse) {
  int rc = ensure_ready(c, "mrh_get_free_blocks");
  if (rc) return rc;
  int h[2] = {0, 0};  // CTR_HEAP_FINE, CTR_HEAP_COARSE are adjacent: stack tops, free count = top + 1
  HIP_TRY(c, hipMemcpyAsync(h, &c->tab.ctr[CTR_HEAP_FINE], 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (out_free_fine) *out_free_fine = (int64_t) h[0] + 1;
  if (out_free_coarse) *out_free_coarse = (int64_t) h[1] + 1;
  return MRH_OK;
}

int mrh_peek_free_blocks(mrh_ctx* c, int64_t* out_free_fine, int64_t* out_free_coarse, uint64_t* out_frames_behind) {
  int rc = ensure_device(c, "mrh_peek_free_blocks");
  if (rc) return rc;
  if (!c->peek_enabled) {  // first call: reports start with the next frame; answer this one the blocking way
    HIP_TRY(c, hipHostMalloc((void**) &c->h_peek, 64 * sizeof(int), hipHostMallocDefault));
    memset(c->h_peek, 0, 64 * sizeof(int));
    c->peek_enabled = true;
  }
  for (uint64_t back = 1; back <= 8 && back < c->frame_seq; back++) {
    const uint64_t seq = c->frame_seq - back;
    if (c->peek_seq[seq % 8] != seq) continue;
    const hipError_t q = hipEventQuery(c->peek_done[seq % 8]);
    if (q == hipErrorNotReady) continue;
    if (q != hipSuccess) return fail(c, MRH_ERR_DEVICE, "mrh_peek_free_blocks: %s", hipGetErrorString(q));
    if (c->peek_seq[seq % 8] != seq) continue;
    if (out_free_fine) *out_free_fine = (int64_t) c->h_peek[8 * (seq % 8)] + 1;
    if (out_free_coarse) *out_free_coarse = (int64_t) c->h_peek[8 * (seq % 8) + 1] + 1;
    // a host-fed frame that mrh_integrate has kept back (flush_deferred) has no sequence number yet: it counts as one more frame behind
    if (out_frames_behind) *out_frames_behind = back - 1 + (c->deferred.on ? 1 : 0);
    return MRH_OK;
  }
  if (out_frames_behind) *out_frames_behind = 0;
  return mrh_get_free_blocks(c, out_free_fine, out_free_coarse);  // blocking: runs a kept-back frame first (ensure_ready)
}

int mrh_peek_error_flags(mrh_ctx* c, uint32_t* out_new_flags) {
  int rc = ensure_device(c, "mrh_peek_error_flags");
  if (rc) return rc;
  if (!out_new_flags) return MRH_ERR_INVALID_ARG;
  *out_new_flags = 0;
  if (!c->peek_enabled) {  // reports start with the next frame
    HIP_TRY(c, hipHostMalloc((void**) &c->h_peek, 64 * sizeof(int), hipHostMallocDefault));
    memset(c->h_peek, 0, 64 * sizeof(int));
    c->peek_enabled = true;
    return MRH_OK;
  }
  for (uint64_t back = 1; back <= 8 && back < c->frame_seq; back++) {
    const uint64_t seq = c->frame_seq - back;
    if (c->peek_seq[seq % 8] != seq) continue;
    const hipError_t q = hipEventQuery(c->peek_done[seq % 8]);
    if (q == hipErrorNotReady) continue;
    if (q != hipSuccess) return fail(c, MRH_ERR_DEVICE, "mrh_peek_error_flags: %s", hipGetErrorString(q));
    const u32 flags = (u32) c->h_peek[8 * (seq % 8) + CTR_ERROR];
    *out_new_flags = flags & ~c->flags_peeked;
    if (*out_new_flags & ERR_POOL) c->table_dirty = true;  // as in take_device_flags: drop the keys without storage before the next frame
    c->flags_peeked = flags;  // the device clears its flags only in mrh_sync: what is set now has been reported
    return MRH_OK;
  }
  return MRH_OK;
}

int mrh_set_profile(mrh_ctx* c, int enabled) {
  int rc = ensure_ready(c, "mrh_set_profile");
  if (rc) return rc;
  c->profile = enabled ? 1 : 0;
  return MRH_OK;
}

int mrh_get_stats(mrh_ctx* c, mrh_stats* out) {
  int rc = ensure_ready(c, "mrh_get_stats");
  if (rc) return rc;
  if (!out) return MRH_ERR_INVALID_ARG;
  hipStream_t s = c->stream;
  HIP_TRY(c, hipMemsetAsync(&c->tab.ctr[CTR_LIVE_FINE], 0, 2 * sizeof(int), s));
  HIP_TRY(c, hipMemsetAsync(&c->tab.ctr[CTR_MAXPROBE], 0, 2 * sizeof(int), s));  // CTR_MAXPROBE, CTR_TOMBS_NOW
  k_count_live<<<256, 256, 0, s>>>(c->tab);
  k_table_census<<<(int) std::min<uint64_t>(2048, (c->slots + 255) / 256), 256, 0, s>>>(c->tab, (size_t) c->slots);
  int h_ctr[CTR_COUNT];
  u64 h_prof[PROF_COUNT];
  const bool fastp = !c->tab.multi_res;
  std::vector<u64> partials(fastp ? (size_t) 32768 * 4 : (size_t) c->integrate_grid);
  HIP_TRY(c, hipMemcpyAsync(h_ctr, c->tab.ctr, sizeof h_ctr, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(h_prof, c->tab.prof, sizeof h_prof, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(partials.data(), fastp ? c->d_cnt_partials : c->d_upd_partials, partials.size() * sizeof(u64), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  rc = drain_events(c);
  if (rc) return rc;
  u64 total_upd = h_prof[PROF_UPDATED];  // LiDAR scans (k_points_apply); the image paths count through the partials
  for (u64 v : partials) total_upd += v;
  memset(out, 0, sizeof *out);
  out->frames_integrated = c->frames;
  out->num_sdf_blocks = c->num_blocks;
  out->occupied_fine = (uint64_t) h_ctr[CTR_LIVE_FINE];
  out->occupied_coarse = (uint64_t) h_ctr[CTR_LIVE_COARSE];
  out->free_fine = (int64_t) h_ctr[CTR_HEAP_FINE] + 1;
  out->free_coarse = (int64_t) h_ctr[CTR_HEAP_COARSE] + 1;
  out->last_compact_blocks = (uint64_t) h_ctr[CTR_COMPACT] + (fastp ? (uint64_t) h_ctr[CTR_CULLED] + (uint64_t) h_ctr[CTR_FREED_EARLY] : 0);
  if (fastp && c->h_levels && out->last_compact_blocks >= (uint64_t) h_ctr[CTR_ZSKIP]) out->last_compact_blocks -= (uint64_t) h_ctr[CTR_ZSKIP];  // list entries that were unwanted zombies
  out->total_updated_voxels = total_upd;
  out->last_updated_voxels = total_upd - c->prev_total_updated;
  out->last_inserted_blocks = h_prof[PROF_INSERTED] - c->prev_inserted;
  out->last_freed_blocks = h_prof[PROF_FREED] - c->prev_freed;
  c->prev_total_updated = total_upd;
  c->prev_inserted = h_prof[PROF_INSERTED];
  c->prev_freed = h_prof[PROF_FREED];
  out->total_compact_blocks = h_prof[PROF_COMPACT];
  out->last_triangles = c->last_triangles;
  out->last_integrate_kernel_ms = c->last_ms;
  out->sum_integrate_kernel_ms = c->sum_ms;
  out->n_integrate_kernel = c->n_ms;
  out->error_flags = (u32) h_ctr[CTR_ERROR] | c->flags_seen;
  out->hash_slots = c->slots;
  out->tombstones = (uint64_t) h_ctr[CTR_TOMBS_NOW];
  out->max_probe_length = (u32) h_ctr[CTR_MAXPROBE];
  out->rehash_count = (u32) h_ctr[CTR_NREHASH];
  out->last_mc_count_ms = c->last_mc_count_ms;
  out->last_mc_emit_ms = c->last_mc_emit_ms;
  out->last_mc_blocks = c->last_mc_blocks;
  out->sum_front_kernel_ms = c->sum_front_ms;
  out->n_front_kernel = c->n_front_ms;
  HIP_TRY(c, hipMemsetAsync(&c->tab.ctr[CTR_TOMBS_NOW], 0, sizeof(int), s));  // the census accumulator belongs to maintain_table
  return MRH_OK;
}

int mrh_extract_triangles(mrh_ctx* c, const mrh_triangle** out_tris, uint64_t* out_n) {
  int rc = ensure_ready(c, "mrh_extract_triangles");
  if (rc) return rc;
  if (!out_n) return MRH_ERR_INVALID_ARG;
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_extract_triangles: an exchange is pending (call mrh_integrate_resume)");
  // out_tris == NULL: the caller only wants the mesh (mrh_extract_mesh) — the soup stays on the device.  The host
  // restatement of the post-process (MRH_MESH_HOST=1) reads the host copy, so it keeps it.
  const bool want_soup = out_tris != nullptr || c->mesh_on_host;
  uint64_t n_tris = 0;
  hipStream_t s = c->stream;
  const bool dbg = getenv("MRH_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now(), t1 = t0, t2 = t0, t3 = t0, t4 = t0, t5 = t0;
  int n = 0;
  rc = compact_all(c, &n);  // the one scalar the host needs up front: it sizes the sort and the launches
  if (rc) return rc;
  c->n_extractions++;  // from here on the caller may hold pointers into the result buffers: the prewarm leaves them alone
  c->tris.clear();
  c->tri_blocks.clear();
  c->tri_counts.clear();
  c->tri_dev_n = 0;
  c->last_triangles = 0;
  c->soup_n = 0;
  bool processed = false;
  if (n > 0) {
    // Everything between the block count and the triangle total stays on the device: canonical order by a radix sort of the
    // packed keys (key order == (x, y, z) order), the 27-block neighbourhoods resolved by one thread per (block, neighbour),
    // per-block triangle counts -> exclusive scan (k_mc_scan_total) -> exact offsets, the emit pass launched right behind it.  The
    // sorted list and the counts are read back only if somebody asks (mrh_get_triangle_blocks).
    u64 *k_in, *k_out, *d_offsets, *d_total;
    int4* sorted;
    u32 *d_counts, *d_nb, *d_rec_base, *d_rec_n, *d_rec_ctr, *d_partial;
    uint8_t* d_per_voxel;  // triangles per voxel from the count pass: k_mc<emit> (the fallback of the record pass) skips the empty ones
    void* tmp;
    int4* sorted2 = nullptr;
    const bool no_rank_sort = getenv("MRH_MC_RADIX_SORT") != nullptr;  // MRH_MC_RADIX_SORT=1: the radix sort of mrh_sort.h for every list (A/B, tests)
    const bool radix = n > kRankSortMax || no_rank_sort;
    const u32 sort_tiles = (u32) ((n + kSortTile - 1) / kSortTile);
    const size_t tmp_bytes = radix ? ((size_t) 256 * sort_tiles + 256) * sizeof(u32) : 0;  // digit totals + the tile histogram of a pass
    {
      MeshScratch a;
      const size_t rank_words = !radix ? (size_t) n * (size_t) ((n + kRankSlice - 1) / kRankSlice) : 1;  // k_block_rank: one row of partial ranks per slice
      a.bytes = (size_t) n * (8 * 3 + 16 + 4 + 4 * kMcNbStride + 512 + 8) + rank_words * 4 + tmp_bytes + (radix ? (size_t) n * sizeof(int4) + 256 : 0) + 32 * 256;
      rc = arena_get(c, 0, a.bytes, &a.base);
      if (rc) return rc;
      k_in = a.take<u64>((size_t) n); k_out = a.take<u64>((size_t) n); d_offsets = a.take<u64>((size_t) n);
      sorted = a.take<int4>((size_t) n); d_counts = a.take<u32>((size_t) n); d_nb = a.take<u32>((size_t) n * kMcNbStride);
      d_per_voxel = a.take<uint8_t>((size_t) n * 512); d_total = a.take<u64>(2);
      d_rec_base = a.take<u32>((size_t) n); d_rec_n = a.take<u32>((size_t) n); d_rec_ctr = a.take<u32>(2);
      d_partial = a.take<u32>(rank_words);
      tmp = a.take<char>(tmp_bytes ? tmp_bytes : 1);
      if (radix) sorted2 = a.take<int4>((size_t) n);
    }
    if (!c->h_mc) {
      HIP_TRY(c, hipHostMalloc((void**) &c->h_mc, 8 * sizeof(u64), hipHostMallocDefault));
      memset(c->h_mc, 0, 8 * sizeof(u64));
    }
    if (!radix) {  // canonical order by counting (mrh_mc.h: k_block_rank)
      const int slices = (n + kRankSlice - 1) / kRankSlice;
      k_list_keys<<<(n + 255) / 256, 256, 0, s>>>(c->tab.compact, n, k_in);
      k_block_rank<<<dim3((n + 255) / 256, slices), 256, 0, s>>>(k_in, n, d_partial);
      k_block_scatter<<<(n + 255) / 256, 256, 0, s>>>(c->tab.compact, n, d_partial, slices, sorted);
    } else {
      // lists beyond the counting rank: the stable byte-wise radix sort of mrh_sort.h over the 64-bit position keys, the list
      // entries riding along (eight passes of histogram / scan / scatter; the first reads the list itself, the last lands in `sorted`)
      k_list_keys<<<(n + 255) / 256, 256, 0, s>>>(c->tab.compact, n, k_in);
      u32* totals = (u32*) tmp;
      u32* hist = totals + 256;
      u64* ks[2] = {k_in, k_out};
      int4* vs[2] = {sorted, sorted2};
      int src = 0;
      for (int shift = 0; shift < 64; shift += 8) {
        k_sort_hist<u64><<<sort_tiles, kSortThreads, 0, s>>>(ks[src], (u32) n, shift, hist, sort_tiles);
        k_sort_scan<<<256, 256, 0, s>>>(hist, sort_tiles, totals);
        k_sort_scatter<u64, int4><<<sort_tiles, kSortThreads, 0, s>>>(ks[src], shift == 0 ? (const int4*) c->tab.compact : (const int4*) vs[src], ks[src ^ 1], vs[src ^ 1], (u32) n, shift,
                                                                      hist, sort_tiles, totals);
        src ^= 1;
      }
      static_assert((64 / 8) % 2 == 0, "an even number of passes ends in the first buffer pair");
    }
    k_mc_neighbors<<<(int) (((size_t) n * 32 + 255) / 256), 256, 0, s>>>(c->tab, sorted, n, d_nb);
    if (dbg) { HIP_TRY(c, hipStreamSynchronize(s)); t1 = now(); }
    // One workgroup per block, block e = workgroup id: the eight XCDs walk the position-sorted list side by side, so a block's
    // neighbours are staged by other XCDs at about the same time and their rows come out of the memory-side Infinity Cache
    // (rocprofv3 counts 3.3 x the algorithmic bytes on the L2 -> fabric side).  Round 5 measured the alternative — runs of 2^k
    // blocks dealt to the XCDs in turn, so that neighbours share an L2 (mc_first_block; MRH_MC_SLAB_LOG2=k switches it on): the
    // fabric traffic falls to 261 / 253 / 217 / 172 / 160 MB for k = 3 / 5 / 7 / 9 / one run per XCD, and the count pass gets
    // SLOWER with every step, 0.246 -> 0.258 / 0.263 / 0.284 / 0.350 / 0.347 ms (profiles/r05/README.md): the re-reads are
    // Infinity Cache hits, not HBM traffic, and they are not what the kernel waits for.  Off by default.
    int slab_log2 = getenv("MRH_MC_SLAB_LOG2") ? std::min(20, std::max(0, atoi(getenv("MRH_MC_SLAB_LOG2")))) : -1;
    while (slab_log2 > 0 && ((size_t) 8 << slab_log2) > (size_t) n + 8) slab_log2--;  // never more than one run per XCD
    const int grid = slab_log2 < 0 ? n : (int) ((((size_t) n + ((size_t) 8 << slab_log2) - 1) >> (slab_log2 + 3)) << (slab_log2 + 3));
    // largest truncation a stored sample can carry (integration clamps to trunc + scale * depth, depth <= the integration distance)
    const float sdf_bound = c->has_camera && !getenv("MRH_MC_NO_PRESCREEN") ? c->map.trunc + c->map.trunc_scale * c->cam.max_int_dist : 0.f;
    c->last_mc_count_ms = c->last_mc_emit_ms = 0.f;
    c->last_mc_blocks = (uint64_t) n;
    // kernel times for mrh_stats only in profile mode: launches that carry events switch the queue to its profiling mode,
    // which slows every later dispatch of the process
    const bool timed = c->profile != 0;
    if (timed)
      for (hipEvent_t& ev : c->mc_ev)
        if (!ev) HIP_TRY(c, hipEventCreate(&ev));
    // Corner records (mrh_mc.h McRecords): the count pass parks the corner values of every voxel that produces triangles, the
    // emit pass interpolates them.  The buffer is sized from the last extraction's demand (first time: 128 records a block);
    // if a block finds no room the whole extraction is emitted by k_mc<emit> and the buffer grows for the next one.
    // MRH_MC_NO_RECORDS=1 keeps the two-pass evaluation (tests compare the two).
    bool use_records = getenv("MRH_MC_NO_RECORDS") == nullptr;
    if (use_records && c->mc_rec_cap == 0) {
      const char* per_block = getenv("MRH_MC_RECORDS_PER_BLOCK");  // tests: a first buffer too small for the map
      const size_t cap = std::max<size_t>((size_t) n * (size_t) (per_block ? std::max(1, atoi(per_block)) : 128), 16);
      if (hipMalloc((void**) &c->d_mc_recs, cap * kMcRecWords * sizeof(u32)) == hipSuccess) c->mc_rec_cap = cap;
      else { (void) hipGetLastError(); c->d_mc_recs = nullptr; use_records = false; }  // no room for the records: the two-pass emit needs none
    }
    McRecords R;
    R.ctr = d_rec_ctr; R.recs = use_records ? c->d_mc_recs : nullptr; R.base = d_rec_base; R.count = d_rec_n;
    R.cap = (u32) std::min<size_t>(c->mc_rec_cap, 0xFFFFFFF0u);
    McRecords none;
    none.ctr = nullptr; none.recs = nullptr; none.base = nullptr; none.count = nullptr; none.cap = 0;
    if (use_records) HIP_TRY(c, hipMemsetAsync(d_rec_ctr, 0, 2 * sizeof(u32), s));
    const int mc_flags = (getenv("MRH_MC_NO_COARSE_KNOWN") ? 2 : 0)   // bit 1: coarse voxels on the literal evaluation only (A/B, tests)
                         | (slab_log2 < 0 ? 4 : (slab_log2 << 4));    // bit 2: block e = workgroup id; else bits 4..8: log2 of an XCD's run of blocks
    if (timed) hipExtLaunchKernelGGL((k_mc<false>), dim3(grid), dim3(kMcThreads), 0, s, c->mc_ev[0], c->mc_ev[1], 0u, c->map, c->tab, (const int4*) sorted, n, (const u32*) d_nb,
                                     (u32*) d_counts, (const u64*) nullptr, (mrh_triangle*) nullptr, (u64) 0, (uint8_t*) d_per_voxel, sdf_bound, mc_flags, R);
    else k_mc<false><<<grid, kMcThreads, 0, s>>>(c->map, c->tab, sorted, n, d_nb, d_counts, nullptr, nullptr, (u64) 0, d_per_voxel, sdf_bound, mc_flags, R);
    // exact offsets + the total: one workgroup chains tiles of 8 192 counts through a carry (10^6 blocks: 122 tiles, ~0.2 ms
    // next to the ~20 ms of their count pass)
    k_mc_scan_total<<<1, 1024, 0, s>>>(d_counts, n, d_offsets, use_records ? d_rec_ctr : nullptr, d_total);
    HIP_TRY(c, hipMemcpyAsync(c->h_mc, d_total, 2 * sizeof(u64), hipMemcpyDeviceToHost, s));
    auto emit = [&](const u64 cap, const int flag_overflow, const bool from_records) {
      if (from_records) {
        if (timed) hipExtLaunchKernelGGL(k_mc_emit_records, dim3(grid), dim3(kMcThreads), 0, s, c->mc_ev[2], c->mc_ev[3], 0u, c->map, c->tab, (const int4*) sorted, n, R,
                                         (const u64*) d_offsets, (mrh_triangle*) c->d_soup, cap, flag_overflow | mc_flags);
        else k_mc_emit_records<<<grid, kMcThreads, 0, s>>>(c->map, c->tab, sorted, n, R, d_offsets, c->d_soup, cap, flag_overflow | mc_flags);
      } else {
        if (timed) hipExtLaunchKernelGGL((k_mc<true>), dim3(grid), dim3(kMcThreads), 0, s, c->mc_ev[2], c->mc_ev[3], 0u, c->map, c->tab, (const int4*) sorted, n, (const u32*) d_nb,
                                         (u32*) d_counts, (const u64*) d_offsets, (mrh_triangle*) c->d_soup, cap, (uint8_t*) d_per_voxel, 0.f, flag_overflow | mc_flags, none);
        else k_mc<true><<<grid, kMcThreads, 0, s>>>(c->map, c->tab, sorted, n, d_nb, d_counts, d_offsets, c->d_soup, cap, d_per_voxel, 0.f, flag_overflow | mc_flags, none);
      }
    };
    // The emit pass goes out BEFORE the host knows the total, into the soup buffer of the previous extraction (grow-only, 12 %
    // head room): a map that is extracted again — the usual case — needs no round trip between the two passes.  Writes beyond
    // the capacity are suppressed by the kernel; if the total turns out larger, the buffer grows and the pass runs again.
    const u64 spec_cap = std::min<u64>(c->soup_cap, c->max_triangles);
    // the host waits for the TOTAL, not for the emit pass behind it: it sizes and enqueues the post-process while the emit pass
    // runs (a stream synchronisation here left the GPU idle for the ~20 us of the host's round trip and first launch)
    if (!c->ev_mc_total) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_mc_total, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_mc_total, s));
    if (spec_cap > 0) emit(spec_cap, 0, use_records);
    HIP_TRY(c, hipEventSynchronize(c->ev_mc_total));
    const u64 total = c->h_mc[0];
    const u64 rec_demand = c->h_mc[1] & ~(1ull << 63);
    const bool records_ok = use_records && (c->h_mc[1] >> 63) == 0;
    const bool emitted = spec_cap > 0 && total <= spec_cap && (records_ok || !use_records);
    struct GrowRecords {  // on every way out: room for this map's demand (+ 25 %) at the next extraction
      mrh_ctx* c; u64 demand;
      ~GrowRecords() {
        if (demand <= c->mc_rec_cap) return;
        (void) hipStreamSynchronize(c->stream);
        if (c->d_mc_recs) (void) hipFree(c->d_mc_recs);
        c->d_mc_recs = nullptr; c->mc_rec_cap = 0;
        const size_t cap = (size_t) (demand + demand / 4);
        if (hipMalloc((void**) &c->d_mc_recs, cap * kMcRecWords * sizeof(u32)) == hipSuccess) c->mc_rec_cap = cap;
        else (void) hipGetLastError();  // no room: the next extraction starts from the default again
      }
    } grow_records{c, use_records ? rec_demand : 0};
    if (dbg) t2 = now();
    c->tri_dev_n = n;
    c->d_tri_sorted = sorted;
    c->d_tri_counts = d_counts;
    if (total > c->max_triangles) {
      return fail(c, MRH_ERR_CAPACITY, "triangle buffer full: %llu triangles > max_triangles %llu", (unsigned long long) total, (unsigned long long) c->max_triangles);
    }
    if (total > 0) {
      if (!emitted) {
        rc = ensure_soup(c, (size_t) total);
        if (rc) return rc;
        emit(total, 1, records_ok);
        if (use_records && !records_ok) c->mc_rec_fallbacks++;
      }
      mrh_triangle* d_tris = c->d_soup;
      c->soup_n = (size_t) total;
      if (timed) {
        HIP_TRY(c, hipEventSynchronize(c->mc_ev[3]));
        HIP_TRY(c, hipEventElapsedTime(&c->last_mc_count_ms, c->mc_ev[0], c->mc_ev[1]));
        HIP_TRY(c, hipEventElapsedTime(&c->last_mc_emit_ms, c->mc_ev[2], c->mc_ev[3]));
      }
      if (dbg) { HIP_TRY(c, hipStreamSynchronize(s)); t3 = now(); }
      if (want_soup) {
        c->tris.resize_discard(total);
        if (c->tris.dev) {  // pinned: out through the copy kernel (see k_copy_out)
          CopyOut a;
          for (int p = 0; p < 3; p++) { a.src[p] = nullptr; a.dst[p] = nullptr; a.count[p] = nullptr; a.fixed[p] = 0; a.cap[p] = 0; a.unit[p] = 0; }
          a.src[0] = (const uint4*) d_tris; a.dst[0] = (uint4*) c->tris.dev; a.fixed[0] = total; a.cap[0] = total; a.unit[0] = (u32) sizeof(mrh_triangle);
          k_copy_out<<<1024, 256, 0, s>>>(a);
        } else {
          HIP_TRY(c, hipMemcpyAsync(c->tris.data(), d_tris, total * sizeof(mrh_triangle), hipMemcpyDeviceToHost, s));
        }
      }
      if (dbg) { HIP_TRY(c, hipStreamSynchronize(s)); t4 = now(); }
      int prc = MRH_OK;
      if (c->merge_on) {  // the running mesh takes this soup; the post-process runs once, over everything, in mrh_mesh_merge_end
        if (c->acc_n + total > c->acc_cap) {
          const size_t cap = (c->acc_n + total) + (c->acc_n + total) / 2;
          mrh_triangle* grown = nullptr;
          HIP_TRY(c, hipMalloc((void**) &grown, cap * sizeof(mrh_triangle)));
          if (c->acc_n) HIP_TRY(c, hipMemcpyAsync(grown, c->d_acc, c->acc_n * sizeof(mrh_triangle), hipMemcpyDeviceToDevice, s));
          HIP_TRY(c, hipStreamSynchronize(s));
          if (c->d_acc) HIP_TRY(c, hipFree(c->d_acc));
          c->d_acc = grown; c->acc_cap = cap;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_acc + c->acc_n, d_tris, total * sizeof(mrh_triangle), hipMemcpyDeviceToDevice, s));
        c->acc_n += total;
        processed = true;
      } else if (!c->mesh_on_host) { prc = process_triangles_device(c, d_tris, total); processed = true; }
      HIP_TRY(c, hipStreamSynchronize(s));
      if (prc) return prc;
      n_tris = total;
    } else if (timed) {
      HIP_TRY(c, hipEventElapsedTime(&c->last_mc_count_ms, c->mc_ev[0], c->mc_ev[1]));
    }
    HIP_TRY(c, hipGetLastError());
  }
  c->last_triangles = n_tris;
  if (!processed && !c->merge_on) process_triangles(c);
  t5 = now();
  if (dbg) fprintf(stderr, "[mrhash_hip] extract: %d blocks, %llu triangles | list+sort+neighbours %.2f ms, count+scan(+speculative emit) %.2f, emit %.2f, soup D2H %.2f, post-process + V/F/C D2H %.2f, total %.2f\n",
                   n, (unsigned long long) n_tris, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0);
  if (out_tris) *out_tris = c->tris.empty() ? nullptr : c->tris.data();
  *out_n = n_tris;
  return MRH_OK;
}

// MeshExtractor::merge_mesh_ = true (geowrapper.cpp:161) ... the chunk loop ... the final mesh.  The reference runs
// processTriangles after every extraction, on (running mesh + new soup).  That equals ONE processTriangles over the soups back
// to back: the vertex merge keeps first occurrences with their indices and colours, and "drop degenerate faces" / "drop
// repeated faces keeping the first" are order-preserving filters, so applying them to a prefix first changes nothing
// (tests/test_geowrapper_gpu.py compares with the oracle, which restates the incremental form literally).
int mrh_mesh_merge_begin(mrh_ctx* c) {
  if (!c) return MRH_ERR_INVALID_ARG;
  c->n_extractions++;
  c->merge_on = true;
  c->acc_n = 0;
  c->V.clear(); c->C.clear(); c->F.clear();
  return MRH_OK;
}

int mrh_mesh_merge_end(mrh_ctx* c, uint64_t* out_total_triangles) {
  int rc = ensure_ready(c, "mrh_mesh_merge_end");
  if (rc) return rc;
  if (!c->merge_on) return fail(c, MRH_ERR_STATE, "mrh_mesh_merge_end: no merge in progress (mrh_mesh_merge_begin)");
  c->merge_on = false;
  if (out_total_triangles) *out_total_triangles = c->acc_n;
  c->last_triangles = c->acc_n;
  c->tris.clear();
  if (c->acc_n == 0) { c->V.clear(); c->C.clear(); c->F.clear(); return MRH_OK; }
  if (c->mesh_on_host) {  // MRH_MESH_HOST=1: the host restatement of the post-process
    c->tris.resize_discard(c->acc_n);
    HIP_TRY(c, hipMemcpyAsync(c->tris.data(), c->d_acc, c->acc_n * sizeof(mrh_triangle), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    process_triangles(c);
    return MRH_OK;
  }
  rc = process_triangles_device(c, c->d_acc, c->acc_n);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return rc;
}

int mrh_extract_mesh(mrh_ctx* c, const double** v, uint64_t* nv, const int32_t** f, uint64_t* nf, const double** col) {
  if (!c || !v || !nv || !f || !nf || !col) return MRH_ERR_INVALID_ARG;
  c->n_extractions++;  // the caller holds these pointers until the next extraction
  *v = c->V.empty() ? nullptr : c->V.data();
  *nv = c->V.size() / 3;
  *f = c->F.empty() ? nullptr : c->F.data();
  *nf = c->F.size() / 3;
  *col = c->C.empty() ? nullptr : c->C.data();
  return MRH_OK;
}

// Streamer, device half (streamer.cu:11-160): select by distance from the camera, copy out, free.
int mrh_stream_out(mrh_ctx* c, const float center[3], float radius, mrh_block_desc* descs, mrh_voxel* voxels, uint64_t capacity,
                   uint64_t* out_n) {
  int rc = ensure_ready(c, "mrh_stream_out");
  if (rc) return rc;
  if (!out_n || !center) return MRH_ERR_INVALID_ARG;
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_stream_out: an exchange is pending (call mrh_integrate_resume)");
  c->refill_flag_valid = false;  // freed coarse units change the level the next frame's refill test must see
  int n = 0;
  rc = compact_all(c, &n);
  if (rc) return rc;
  *out_n = 0;
  if (n == 0) return MRH_OK;
  // the selection is a handful of flops per live block: done on the host copy of the list, which is needed for the
  // canonical (position) order anyway
  std::vector<int4> list((size_t) n);
  HIP_TRY(c, hipMemcpy(list.data(), c->tab.compact, (size_t) n * sizeof(int4), hipMemcpyDeviceToHost));
  const float vs = c->map.vs;
  std::vector<int4> sel;
  sel.reserve((size_t) n);
  for (const int4& e : list) {
    const float px = (float) (e.x * kBlockSide) * vs, py = (float) (e.y * kBlockSide) * vs, pz = (float) (e.z * kBlockSide) * vs;
    const float dx = px - center[0], dy = py - center[1], dz = pz - center[2];
    const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
    if (radius >= 0.f && !(d >= radius)) continue;
    sel.push_back(e);
  }
  *out_n = sel.size();
  if (!descs || sel.empty()) return MRH_OK;
  if (sel.size() > capacity) return fail(c, MRH_ERR_CAPACITY, "mrh_stream_out: capacity %llu < %zu blocks to stream out", (unsigned long long) capacity, sel.size());
  std::sort(sel.begin(), sel.end(), [](const int4& a, const int4& b) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    return a.z < b.z;
  });
  const int ns = (int) sel.size();
  HIP_TRY(c, hipMemcpy(c->tab.compact, sel.data(), (size_t) ns * sizeof(int4), hipMemcpyHostToDevice));
  const int chunk = 8192;
  DevBuf<int4> d_descs;
  DevBuf<char> d_vox;
  HIP_TRY(c, d_descs.alloc((size_t) chunk));
  HIP_TRY(c, d_vox.alloc((size_t) chunk * kFineBytes));
  for (int first = 0; first < ns; first += chunk) {
    const int cnt = (ns - first) < chunk ? (ns - first) : chunk;
    k_dump<<<cnt < 2048 ? cnt : 2048, 512, 0, c->stream>>>(c->tab, first, cnt, d_descs, d_vox);
    HIP_TRY(c, hipMemcpyAsync(&descs[first], d_descs, (size_t) cnt * sizeof(int4), hipMemcpyDeviceToHost, c->stream));
    if (voxels) HIP_TRY(c, hipMemcpyAsync(&voxels[(size_t) first * 512], d_vox, (size_t) cnt * kFineBytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  // free: garbageCollectFree's kernel over the selected list with every decision set
  const int ctr_n = ns;
  HIP_TRY(c, hipMemcpyAsync(&c->tab.ctr[CTR_COMPACT], &ctr_n, sizeof(int), hipMemcpyHostToDevice, c->stream));
  k_fill_u32<<<256, 256, 0, c->stream>>>(c->d_decision, (size_t) ns, 1u);
  k_gc_free<false><<<256, 256, 0, c->stream>>>(c->tab, c->d_decision);
  c->table_dirty = true;  // a bulk erase: census (and, if due, rebuild) before the next frame
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  return MRH_OK;
}

int mrh_dump_blocks(mrh_ctx* c, mrh_block_desc* descs, mrh_voxel* voxels, uint64_t capacity, uint64_t* out_n) {
  int rc = ensure_ready(c, "mrh_dump_blocks");
  if (rc) return rc;
  if (!out_n) return MRH_ERR_INVALID_ARG;
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_dump_blocks: an exchange is pending (call mrh_integrate_resume)");  // the starve passes still need Tab::compact
  int n = 0;
  rc = compact_all(c, &n);
  if (rc) return rc;
  *out_n = (uint64_t) n;
  if (!descs) return MRH_OK;
  if ((uint64_t) n > capacity) return fail(c, MRH_ERR_CAPACITY, "mrh_dump_blocks: capacity %llu < %d live blocks", (unsigned long long) capacity, n);
  const int chunk = 8192;  // 48 MiB of voxels per round trip
  DevBuf<int4> d_descs;
  DevBuf<char> d_vox;
  HIP_TRY(c, d_descs.alloc((size_t) chunk));
  HIP_TRY(c, d_vox.alloc((size_t) chunk * kFineBytes));
  for (int first = 0; first < n; first += chunk) {
    const int cnt = (n - first) < chunk ? (n - first) : chunk;
    k_dump<<<cnt < 2048 ? cnt : 2048, 512, 0, c->stream>>>(c->tab, first, cnt, d_descs, d_vox);
    HIP_TRY(c, hipMemcpyAsync(&descs[first], d_descs, (size_t) cnt * sizeof(int4), hipMemcpyDeviceToHost, c->stream));
    if (voxels) HIP_TRY(c, hipMemcpyAsync(&voxels[(size_t) first * 512], d_vox, (size_t) cnt * kFineBytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  HIP_TRY(c, hipGetLastError());
  return MRH_OK;
}

int mrh_get_voxel(mrh_ctx* c, int32_t vx, int32_t vy, int32_t vz, mrh_voxel* out, int* out_found) {
  int rc = ensure_ready(c, "mrh_get_voxel");
  if (rc) return rc;
  if (!out) return MRH_ERR_INVALID_ARG;
  k_get_voxel<<<1, 1, 0, c->stream>>>(c->map, c->tab, vx, vy, vz, c->d_misc);
  u32 h[4];
  HIP_TRY(c, hipMemcpyAsync(h, c->d_misc, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  memcpy(&out->sdf, &h[0], 4);
  memcpy(&out->sum_squared, &h[1], 4);
  out->rgb[0] = h[2] & 0xFF; out->rgb[1] = (h[2] >> 8) & 0xFF; out->rgb[2] = (h[2] >> 16) & 0xFF;
  out->weight = (uint8_t) (h[2] >> 24);
  if (out_found) *out_found = (int) h[3];
  return MRH_OK;
}

namespace {
// flags raised by earlier frames are set aside (mrh_sync reports them) so that a call which checks its own outcome on the
// device — import, unpack — answers for itself only
int set_aside_flags(mrh_ctx* c) {
  u32 flags = 0;
  int rc = take_device_flags(c, &flags);
  if (rc) return rc;
  c->flags_deferred |= flags;
  return MRH_OK;
}
void map_changed_in_bulk(mrh_ctx* c) {
  c->mr_next_general = true;  // payload that has not been through a variance check
  c->refill_flag_valid = false;
  c->mr_summaries_valid = false;
  c->table_dirty = true;
}
}  // namespace

// room on the coarse free list for `need` coarse blocks, in allocateMemoryLow's portions (vds.cu:860-871: k_refill): an import or a
// merge into a context whose frames have not refilled the list yet (vds.cu:885-891 does it at the start of a frame).  Blocks.
static int ensure_coarse_units(mrh_ctx* c, const uint64_t need) {
  if (!c->tab.multi_res || need == 0) return MRH_OK;
  hipStream_t s = c->stream;
  int lev[2] = {0, 0};  // CTR_HEAP_FINE, CTR_HEAP_COARSE are adjacent: stack tops, free count = top + 1
  HIP_TRY(c, hipMemcpyAsync(lev, &c->tab.ctr[CTR_HEAP_FINE], 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  int64_t fine_free = (int64_t) lev[0] + 1, coarse_free = (int64_t) lev[1] + 1;
  while (coarse_free < (int64_t) need && c->low_blocks_to_allocate > 0 && fine_free > (int64_t) c->low_blocks_to_allocate) {
    HIP_TRY(c, hipMemsetAsync(c->d_flag, 0xFF, sizeof(int), s));  // any non-zero flag: refill
    k_refill<<<(c->low_blocks_to_allocate + 255) / 256, 256, 0, s>>>(c->tab, c->low_blocks_to_allocate, c->d_flag);
    fine_free -= c->low_blocks_to_allocate;
    coarse_free += 8 * (int64_t) c->low_blocks_to_allocate;
  }
  c->refill_flag_valid = false;
  HIP_TRY(c, hipGetLastError());
  return MRH_OK;
}

int mrh_import_blocks(mrh_ctx* c, const mrh_block_desc* descs, const mrh_voxel* voxels, uint64_t n) {
  int rc = ensure_ready(c, "mrh_import_blocks");
  if (rc) return rc;
  if (n == 0) return MRH_OK;
  if (!descs || !voxels) return fail(c, MRH_ERR_INVALID_ARG, "mrh_import_blocks: null argument");
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_import_blocks: an exchange is pending (call mrh_integrate_resume)");
  rc = set_aside_flags(c);
  if (rc) return rc;
  if (c->tab.multi_res) {
    uint64_t need = 0;
    for (uint64_t k = 0; k < n; k++) need += descs[k].resolution != 0;
    rc = ensure_coarse_units(c, need);
    if (rc) return rc;
  }
  map_changed_in_bulk(c);
  // two staging buffers, the copies on their own stream: the host-to-device copy of chunk i + 1 (the caller's memory is
  // pageable: the runtime stages it) runs under the insert kernel of chunk i; one synchronisation at the end
  const uint64_t chunk = 4096;  // 24 MiB of voxels
  struct Pipe {
    hipStream_t copy = nullptr;
    hipEvent_t copied[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    ~Pipe() {
      if (copy) { (void) hipStreamSynchronize(copy); (void) hipStreamDestroy(copy); }
      for (hipEvent_t e : copied) if (e) (void) hipEventDestroy(e);
      for (hipEvent_t e : done) if (e) (void) hipEventDestroy(e);
    }
  } pipe;
  DevBuf<int4> d_descs[2];
  DevBuf<char> d_vox[2];
  const int nbuf = n > chunk ? 2 : 1;
  HIP_TRY(c, hipStreamCreateWithFlags(&pipe.copy, hipStreamNonBlocking));
  for (int b = 0; b < nbuf; b++) {
    HIP_TRY(c, d_descs[b].alloc(chunk));
    HIP_TRY(c, d_vox[b].alloc(chunk * (size_t) kFineBytes));
    HIP_TRY(c, hipEventCreateWithFlags(&pipe.copied[b], hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&pipe.done[b], hipEventDisableTiming));
  }
  uint64_t it = 0;
  for (uint64_t first = 0; first < n; first += chunk, it++) {
    const uint64_t cnt = (n - first) < chunk ? (n - first) : chunk;
    const int b = (int) (it & 1);
    if (it >= 2) HIP_TRY(c, hipStreamWaitEvent(pipe.copy, pipe.done[b], 0));  // the kernel that read this buffer two chunks ago
    HIP_TRY(c, hipMemcpyAsync(d_descs[b], &descs[first], cnt * sizeof(int4), hipMemcpyHostToDevice, pipe.copy));
    HIP_TRY(c, hipMemcpyAsync(d_vox[b], &voxels[first * 512], cnt * (size_t) kFineBytes, hipMemcpyHostToDevice, pipe.copy));
    HIP_TRY(c, hipEventRecord(pipe.copied[b], pipe.copy));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, pipe.copied[b], 0));
    k_import<kImportPlain><<<(int) (cnt < 2048 ? cnt : 2048), 512, 0, c->stream>>>(c->map, c->tab, c->fast.summary, (int) cnt, (const char*) (int4*) d_descs[b], sizeof(int4),
                                                                                   (const char*) d_vox[b], (size_t) kFineBytes, nullptr, nullptr);
    HIP_TRY(c, hipEventRecord(pipe.done[b], c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  u32 flags = 0;
  rc = take_device_flags(c, &flags);
  if (rc) return rc;
  return check_device_flags(c, flags);
}

// ---- multi-GPU block exchange (include/mrhash_hip.h) -----------------------------------------------------------------

int mrh_set_sharding(mrh_ctx* c, int shard_rank, int shard_count, int shard_chunk_log2) {
  if (!c) return MRH_ERR_INVALID_ARG;
  if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count) return fail(c, MRH_ERR_INVALID_ARG, "mrh_set_sharding: rank %d of %d", shard_rank, shard_count);
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_set_sharding: an exchange is pending (call mrh_integrate_resume)");
  {
    const int frc = ensure_ready(c, "mrh_set_sharding");  // the last pipelined frame is integrated under the ownership it was allocated with
    if (frc) return frc;
  }
  c->p.shard_rank = shard_rank; c->p.shard_count = shard_count; c->p.shard_chunk_log2 = shard_chunk_log2;
  c->map.shard_rank = shard_rank;
  c->map.shard_count = shard_count;
  c->map.shard_chunk_log2 = (shard_chunk_log2 > 0 && shard_chunk_log2 < 16) ? shard_chunk_log2 : 3;
  return MRH_OK;
}

namespace {
// live blocks matching a predicate -> Tab::compact[0, n)
int select_blocks(mrh_ctx* c, int sel_mode, int rank_arg, int* out_n) {
  hipStream_t s = c->stream;
  HIP_TRY(c, hipMemsetAsync(&c->tab.ctr[CTR_COMPACT], 0, sizeof(int), s));
  k_select_blocks<<<512, 256, 0, s>>>(c->map, c->tab, sel_mode, rank_arg);
  int n = 0;
  HIP_TRY(c, hipMemcpyAsync(&n, &c->tab.ctr[CTR_COMPACT], sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  *out_n = n;
  return MRH_OK;
}
// frees Tab::compact[0, n) (garbageCollectFree's kernel with every decision set)
int free_compact(mrh_ctx* c, int n) {
  if (n <= 0) return MRH_OK;
  hipStream_t s = c->stream;
  HIP_TRY(c, hipMemcpyAsync(&c->tab.ctr[CTR_COMPACT], &n, sizeof(int), hipMemcpyHostToDevice, s));
  k_fill_u32<<<256, 256, 0, s>>>(c->d_decision, (size_t) n, 1u);
  k_gc_free<false><<<256, 256, 0, s>>>(c->tab, c->d_decision);
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  map_changed_in_bulk(c);
  return MRH_OK;
}
}  // namespace

int mrh_pack_blocks(mrh_ctx* c, int mode, int rank_arg, const mrh_block_record** out_records, uint64_t* out_n, int* out_is_device_memory) {
  int rc = ensure_ready(c, "mrh_pack_blocks");
  if (rc) return rc;
  if (!out_records || !out_n) return MRH_ERR_INVALID_ARG;
  if (mode != MRH_PACK_HALO && mode != MRH_PACK_OWNER) return fail(c, MRH_ERR_INVALID_ARG, "mrh_pack_blocks: bad mode %d", mode);
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_pack_blocks: an exchange is pending (call mrh_integrate_resume)");
  int n = 0;
  rc = select_blocks(c, mode == MRH_PACK_HALO ? kSelHalo : kSelOwner, rank_arg, &n);
  if (rc) return rc;
  if (out_is_device_memory) *out_is_device_memory = 1;
  *out_n = (uint64_t) n;
  *out_records = nullptr;
  if (n == 0) return MRH_OK;
  const size_t bytes = (size_t) n * sizeof(mrh_block_record);
  if (bytes > c->pack_cap) {
    if (c->d_pack) HIP_TRY(c, hipFree(c->d_pack));
    c->d_pack = nullptr; c->pack_cap = 0;
    const size_t cap = bytes + bytes / 4;
    HIP_TRY(c, hipMalloc((void**) &c->d_pack, cap));
    c->pack_cap = cap;
  }
  k_pack_records<<<n < 4096 ? n : 4096, 512, 0, c->stream>>>(c->tab, 0, n, c->d_pack);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  *out_records = (const mrh_block_record*) c->d_pack;
  return MRH_OK;
}

int mrh_unpack_blocks(mrh_ctx* c, int mode, const mrh_block_record* records, uint64_t n, int is_device_memory, uint64_t* out_taken) {
  int rc = ensure_ready(c, "mrh_unpack_blocks");
  if (rc) return rc;
  if (out_taken) *out_taken = 0;
  if (mode != MRH_UNPACK_HALO && mode != MRH_UNPACK_MERGE) return fail(c, MRH_ERR_INVALID_ARG, "mrh_unpack_blocks: bad mode %d", mode);
  if (n == 0) return MRH_OK;
  if (!records) return fail(c, MRH_ERR_INVALID_ARG, "mrh_unpack_blocks: null argument");
  if (n > 0x7FFFFFFFull) return fail(c, MRH_ERR_CAPACITY, "mrh_unpack_blocks: %llu records in one call", (unsigned long long) n);
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_unpack_blocks: an exchange is pending (call mrh_integrate_resume)");
  rc = set_aside_flags(c);
  if (rc) return rc;
  hipStream_t s = c->stream;
  DevBuf<char> staged;
  const char* d_rec = (const char*) records;
  if (!is_device_memory) {  // host records (tests over gloo): one staged copy
    HIP_TRY(c, staged.alloc((size_t) n * sizeof(mrh_block_record)));
    HIP_TRY(c, hipMemcpyAsync(staged, records, (size_t) n * sizeof(mrh_block_record), hipMemcpyHostToDevice, s));
    d_rec = staged;
  }
  if (!c->d_taken) HIP_TRY(c, hipMalloc((void**) &c->d_taken, sizeof(u32)));
  HIP_TRY(c, hipMemsetAsync(c->d_taken, 0, sizeof(u32), s));
  if (mode == MRH_UNPACK_HALO && c->halo_upper + n > c->halo_cap) {  // room for every record of this call on the halo list
    const size_t cap = (c->halo_upper + n) + (c->halo_upper + n) / 2;
    int4* grown = nullptr;
    HIP_TRY(c, hipMalloc((void**) &grown, cap * sizeof(int4)));
    if (c->d_halo && c->halo_upper) HIP_TRY(c, hipMemcpyAsync(grown, c->d_halo, c->halo_upper * sizeof(int4), hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (c->d_halo) HIP_TRY(c, hipFree(c->d_halo));
    c->d_halo = grown;
    c->halo_cap = cap;
  }
  map_changed_in_bulk(c);
  const int grid = (int) (n < 4096 ? n : 4096);
  const size_t stride = sizeof(mrh_block_record);
  // a merge into a variance-adaptive map: room on the coarse free list for every coarse record of the call, in
  // allocateMemoryLow's portions (vds.cu:860-871: k_refill), and a list for the fine slots that make way for coarse records
  DevBuf<u32> released;
  if (mode == MRH_UNPACK_MERGE && c->tab.multi_res) {
    HIP_TRY(c, released.alloc((size_t) n + 1));
    HIP_TRY(c, hipMemsetAsync(released, 0, sizeof(u32), s));
    k_count_coarse_records<<<64, 256, 0, s>>>(d_rec, stride, (int) n, c->d_taken);  // d_taken doubles as the counter, zeroed again below
    u32 need = 0;
    HIP_TRY(c, hipMemcpyAsync(&need, c->d_taken, sizeof(u32), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipMemsetAsync(c->d_taken, 0, sizeof(u32), s));
    rc = ensure_coarse_units(c, need);
    if (rc) return rc;
  }
  if (mode == MRH_UNPACK_HALO) {
    k_import<kImportHalo><<<grid, 512, 0, s>>>(c->map, c->tab, c->fast.summary, (int) n, d_rec, stride, d_rec + sizeof(mrh_block_desc), stride, c->d_halo, c->d_taken);
    c->halo_upper += n;
  } else {
    k_import<kImportMerge><<<grid, 512, 0, s>>>(c->map, c->tab, c->fast.summary, (int) n, d_rec, stride, d_rec + sizeof(mrh_block_desc), stride, nullptr, c->d_taken,
                                                (u32*) released);
    if (released.p) k_release_fine<<<1, 256, 0, s>>>(c->tab, released);
  }
  u32 taken = 0;
  HIP_TRY(c, hipMemcpyAsync(&taken, c->d_taken, sizeof(u32), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  if (out_taken) *out_taken = taken;
  u32 flags = 0;
  rc = take_device_flags(c, &flags);
  if (rc) return rc;
  return check_device_flags(c, flags);
}

int mrh_drop_blocks(mrh_ctx* c, int mode, uint64_t* out_dropped) {
  int rc = ensure_ready(c, "mrh_drop_blocks");
  if (rc) return rc;
  if (out_dropped) *out_dropped = 0;
  if (c->pending) return fail(c, MRH_ERR_STATE, "mrh_drop_blocks: an exchange is pending (call mrh_integrate_resume)");
  int n = 0;
  if (mode == MRH_DROP_HALO) {
    HIP_TRY(c, hipMemcpyAsync(&n, &c->tab.ctr[CTR_HALO], sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n > 0) HIP_TRY(c, hipMemcpyAsync(c->tab.compact, c->d_halo, (size_t) n * sizeof(int4), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(&c->tab.ctr[CTR_HALO], 0, sizeof(int), c->stream));
    c->halo_upper = 0;
  } else if (mode == MRH_DROP_FOREIGN || mode == MRH_DROP_ALL) {
    rc = select_blocks(c, mode == MRH_DROP_FOREIGN ? kSelForeign : kSelAll, 0, &n);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(&c->tab.ctr[CTR_HALO], 0, sizeof(int), c->stream));  // halo blocks are foreign: they go with the rest
    c->halo_upper = 0;
  } else {
    return fail(c, MRH_ERR_INVALID_ARG, "mrh_drop_blocks: bad mode %d", mode);
  }
  rc = free_compact(c, n);
  if (rc) return rc;
  if (out_dropped) *out_dropped = (uint64_t) n;
  return MRH_OK;
}

int mrh_get_triangle_blocks(mrh_ctx* c, const mrh_block_desc** out_descs, const uint32_t** out_counts, uint64_t* out_n) {
  if (!c || !out_descs || !out_counts || !out_n) return MRH_ERR_INVALID_ARG;
  if (c->tri_dev_n > 0) {  // the list and the counts of the last extraction are still where the kernels left them
    const size_t n = (size_t) c->tri_dev_n;
    std::vector<int4> list(n);
    c->tri_counts.resize(n);
    HIP_TRY(c, hipMemcpyAsync(list.data(), c->d_tri_sorted, n * sizeof(int4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->tri_counts.data(), c->d_tri_counts, n * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->tri_blocks.resize(n);
    for (size_t i = 0; i < n; i++) c->tri_blocks[i] = {list[i].x, list[i].y, list[i].z, (list[i].w & (int) kValCoarseBit) ? 1 : 0};
    c->tri_dev_n = 0;
  }
  *out_descs = c->tri_blocks.empty() ? nullptr : c->tri_blocks.data();
  *out_counts = c->tri_counts.empty() ? nullptr : c->tri_counts.data();
  *out_n = c->tri_blocks.size();
  return MRH_OK;
}

int mrh_process_triangles(mrh_ctx* c, const mrh_triangle* triangles, uint64_t n) {
  if (!c || (n && !triangles)) return MRH_ERR_INVALID_ARG;
  c->tris.assign(triangles, triangles + n);
  c->last_triangles = n;
  if (c->mesh_on_host || n == 0) { process_triangles(c); return MRH_OK; }
  int rc = ensure_ready(c, "mrh_process_triangles");
  if (rc) return rc;
  DevBuf<mrh_triangle> d_tris;
  HIP_TRY(c, d_tris.alloc(n));
  HIP_TRY(c, hipMemcpyAsync(d_tris, triangles, n * sizeof(mrh_triangle), hipMemcpyHostToDevice, c->stream));
  return process_triangles_device(c, d_tris, n);
}

int mrh_get_triangles_device(mrh_ctx* c, const mrh_triangle** out, uint64_t* out_n, int* out_is_device_memory) {
  if (!c || !out || !out_n) return MRH_ERR_INVALID_ARG;
  *out = c->soup_n ? c->d_soup : nullptr;
  *out_n = c->soup_n;
  if (out_is_device_memory) *out_is_device_memory = 1;
  return MRH_OK;
}

int mrh_process_triangle_runs(mrh_ctx* c, const mrh_block_desc* descs, const uint32_t* counts, uint64_t n_blocks, const mrh_triangle* triangles,
                              uint64_t n_triangles, int is_device_memory) {
  int rc = ensure_ready(c, "mrh_process_triangle_runs");
  if (rc) return rc;
  if ((n_blocks && (!descs || !counts)) || (n_triangles && !triangles)) return fail(c, MRH_ERR_INVALID_ARG, "mrh_process_triangle_runs: null argument");
  hipStream_t s = c->stream;
  // runs in input order -> canonical order (block position): a host sort of the few-byte descriptors, a device permutation
  // of the 72-byte triangles
  std::vector<uint64_t> src_off(n_blocks);
  uint64_t total = 0;
  for (uint64_t i = 0; i < n_blocks; i++) { src_off[i] = total; total += counts[i]; }
  if (total != n_triangles) return fail(c, MRH_ERR_INVALID_ARG, "mrh_process_triangle_runs: the counts add up to %llu triangles, %llu given", (unsigned long long) total, (unsigned long long) n_triangles);
  std::vector<uint32_t> order(n_blocks);
  for (uint64_t i = 0; i < n_blocks; i++) order[i] = (uint32_t) i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    if (descs[a].x != descs[b].x) return descs[a].x < descs[b].x;
    if (descs[a].y != descs[b].y) return descs[a].y < descs[b].y;
    return descs[a].z < descs[b].z;
  });
  c->tri_dev_n = 0;
  c->tri_blocks.resize(n_blocks);
  c->tri_counts.resize(n_blocks);
  std::vector<ulonglong2> runs;  // {source offset, destination offset | count << 40}
  runs.reserve(n_blocks);
  uint64_t dst = 0;
  for (uint64_t k = 0; k < n_blocks; k++) {
    const uint32_t i = order[k];
    c->tri_blocks[k] = descs[i];
    c->tri_counts[k] = counts[i];
    if (counts[i]) runs.push_back(make_ulonglong2(src_off[i], dst | ((uint64_t) counts[i] << 40)));
    dst += counts[i];
  }
  c->tris.clear();
  c->last_triangles = n_triangles;
  c->soup_n = 0;
  if (n_triangles == 0) { c->V.clear(); c->C.clear(); c->F.clear(); return MRH_OK; }
  if (n_triangles >= (1ull << 40)) return fail(c, MRH_ERR_CAPACITY, "mrh_process_triangle_runs: too many triangles");
  rc = ensure_soup(c, (size_t) n_triangles);
  if (rc) return rc;
  DevBuf<mrh_triangle> staged;
  const mrh_triangle* d_in = triangles;
  if (!is_device_memory) {
    HIP_TRY(c, staged.alloc(n_triangles));
    HIP_TRY(c, hipMemcpyAsync(staged, triangles, n_triangles * sizeof(mrh_triangle), hipMemcpyHostToDevice, s));
    d_in = staged;
  }
  DevBuf<ulonglong2> d_runs;
  HIP_TRY(c, d_runs.alloc(runs.size()));
  HIP_TRY(c, hipMemcpyAsync(d_runs, runs.data(), runs.size() * sizeof(ulonglong2), hipMemcpyHostToDevice, s));
  k_permute_runs<<<(int) std::min<size_t>(runs.size(), 8192), 256, 0, s>>>((const ulonglong2*) d_runs, (int) runs.size(), (const uint4*) d_in, (uint4*) c->d_soup);
  c->soup_n = (size_t) n_triangles;
  if (c->mesh_on_host) {
    c->tris.resize_discard(n_triangles);
    HIP_TRY(c, hipMemcpyAsync(c->tris.data(), c->d_soup, n_triangles * sizeof(mrh_triangle), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    process_triangles(c);
    return MRH_OK;
  }
  rc = process_triangles_device(c, c->d_soup, n_triangles);
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  return rc;
}

int mrh_selftest_division(mrh_ctx* c, uint64_t samples, uint64_t seed, uint64_t* out_mismatches) {
  int rc = ensure_ready(c, "mrh_selftest_division");
  if (rc) return rc;
  if (!out_mismatches) return MRH_ERR_INVALID_ARG;
  DevBuf<u64> d;
  HIP_TRY(c, d.alloc(1));
  HIP_TRY(c, hipMemsetAsync(d, 0, sizeof(u64), c->stream));
  const u32 threads = 1024 * 256;
  const u32 iters = (u32) ((samples + threads - 1) / threads);
  k_selftest_division<<<1024, 256, 0, c->stream>>>(seed, iters, d);
  u64 h = 0;
  HIP_TRY(c, hipMemcpyAsync(&h, d, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *out_mismatches = h;
  return MRH_OK;
}

}  // extern "C"

#include "mrh_comm.h"
