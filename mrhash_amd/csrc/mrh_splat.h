// mrh_splat.h — 3DGS splat seeds: image quad-tree by colour error + one map lookup per leaf.
//
// Reference: CUDAQTree::subdivide (src/gs/quad_tree.cu:169-223) runs level by level — one 256-thread block per node
// sums the node's pixels twice (computeError, :6-90), a second kernel appends leaves / children through atomic
// counters (:102-167), the host reads the child count back and loops — then processNodesKernel
// (gaussian_data_structures.cu:5-56) looks the centre voxel of every leaf up.  ~20 dependent launches and host
// round trips per frame, and the root alone is a 1 200-step serial chain per thread at 640x480.
//
// Here the tree is not grown, it is DECIDED: the halving rule (w1 = w / 2, w2 = w - w1) fixes every rectangle that
// could ever become a node — (4^(D+1) - 1) / 3 "potential" nodes, numbered level-major by their path of child digits —
// independently of the image.  Per frame:
//   k_qt_sums_bottom / k_qt_sums_up   exact integer statistics (sum p, sum p^2 per channel) of every potential node,
//                                     bottom-up, 4^T nodes per workgroup folded through LDS
//   k_qt_decide     one thread per potential node: the node's error as an exact rational in fp64 against the
//                   threshold.  The reference's fp32 value differs from the exact one by a bounded rounding error
//                   (bound below); outside that band the decision is certain, inside it the node goes on a list
//   k_qt_literal    only listed nodes: the reference's summation order literally (256 strided partial sums + halving
//                   tree; a wave per node when the node has <= 256 pixels) -> the decision the oracle takes
//   k_qt_emit       one thread per potential node: live (all ancestors split) and leaf -> the seed test of
//                   processNodesKernel; flags for a scan, and the workgroup's share of its scan tile's sum
//   k_tile_scan_u64 + k_qt_scatter    leaves and seeds in canonical order (level, then path = the order in which a
//                                     sequential level-by-level subdivision appends them; oracle header D7); the seeds and
//                                     the two counts land in pinned host memory
// No level loop, ONE host synchronisation (behind the last launch; round 6: the counters are cleared by the first launch, the
// tile sums come from k_qt_emit, seeds and counts need no transfer call — nine launches + a memset + two read-backs became
// eight launches), and bit-identical decisions: only `err <= threshold` leaves the error computation, never the error itself.
#pragma once

#include "mrh_mc.h"

namespace mrh {

constexpr int kQtThreads = 256;     // n_threads_subdivide (params.h:18): fixes the summation order of computeError
constexpr int kQtMaxDepth = 11;     // images up to 2^22 pixels
constexpr u32 kQtLeaf = 0, kQtSplit = 1, kQtUncertain = 2, kQtDead = 3;

struct QRect { int x0, y0, w, h; };
struct QSum {  // exact statistics of one potential node
  u64 q[3];    // sum of squares per channel (<= 2^22 * 255^2 < 2^38)
  u32 s[3];    // sum per channel (<= 2^22 * 255 < 2^30)
  u32 pad;
};
struct QTree {
  int W, H, D, min_px;
  u32 total;   // (4^(D+1) - 1) / 3
};

__host__ __device__ __forceinline__ u32 qt_level_offset(int l) { return ((1u << (2 * l)) - 1u) / 3u; }

// child digits from the root, most significant first: bit 1 = right half (w2), bit 0 = bottom half (h2) — the order
// subdivideKernel writes its four children in (quad_tree.cu:163-166)
__device__ __forceinline__ QRect qt_rect(const QTree& q, int level, u32 path) {
  QRect r = {0, 0, q.W, q.H};
  for (int i = level - 1; i >= 0; i--) {
    const u32 k = (path >> (2 * i)) & 3u;
    const int w1 = r.w / 2, h1 = r.h / 2;
    if (k & 2u) { r.x0 += w1; r.w -= w1; } else r.w = w1;
    if (k & 1u) { r.y0 += h1; r.h -= h1; } else r.h = h1;
  }
  return r;
}

__device__ __forceinline__ int qt_level_of(const QTree& q, u32 idx) {
  int l = 0;
  while (l < q.D && qt_level_offset(l + 1) <= idx) l++;
  return l;
}

// ---- exact statistics, bottom-up ----------------------------------------------------------------------------

// LDS layout of a T-level fold: 4^T entries, then 4^(T-1), ... , 1
struct QFold {
  u64 q[3][341];
  u32 s[3][341];
};

template <bool FROM_PIXELS>
__device__ __forceinline__ void qt_fold(const QTree& qt, const uint8_t* __restrict__ rgb, QSum* __restrict__ sums, const int L, const int T,
                                        QFold& f) {
  // workgroup = one node of level L - T; thread = one of its 4^T descendants at level L
  const u32 tid = threadIdx.x, n0 = 1u << (2 * T);
  const u32 path = (blockIdx.x << (2 * T)) | tid;
  if (tid < n0) {
    u64 q[3] = {0, 0, 0};
    u32 s[3] = {0, 0, 0};
    if (FROM_PIXELS) {
      const QRect r = qt_rect(qt, L, path);
      for (int y = 0; y < r.h; y++) {
        const uint8_t* p = rgb + ((size_t) (r.y0 + y) * qt.W + r.x0) * 3;
        for (int x = 0; x < r.w; x++) {
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const u32 v = p[3 * x + k];
            s[k] += v;
            q[k] += v * v;
          }
        }
      }
      QSum o;
#pragma unroll
      for (int k = 0; k < 3; k++) { o.q[k] = q[k]; o.s[k] = s[k]; }
      o.pad = 0;
      sums[qt_level_offset(L) + path] = o;
    } else {
      const QSum o = sums[qt_level_offset(L) + path];
#pragma unroll
      for (int k = 0; k < 3; k++) { q[k] = o.q[k]; s[k] = o.s[k]; }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { f.q[k][tid] = q[k]; f.s[k][tid] = s[k]; }
  }
  __syncthreads();
  u32 src = 0, n = n0;
  for (int t = 1; t <= T; t++) {
    const u32 dst = src + n;
    n >>= 2;
    if (tid < n) {
      QSum o;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const u32 c = src + 4 * tid;
        o.q[k] = f.q[k][c] + f.q[k][c + 1] + f.q[k][c + 2] + f.q[k][c + 3];
        o.s[k] = f.s[k][c] + f.s[k][c + 1] + f.s[k][c + 2] + f.s[k][c + 3];
        f.q[k][dst + tid] = o.q[k];
        f.s[k][dst + tid] = o.s[k];
      }
      o.pad = 0;
      sums[qt_level_offset(L - t) + ((blockIdx.x << (2 * (T - t))) | tid)] = o;
    }
    src = dst;
    __syncthreads();
  }
}

// (the first launch of a seeding call also clears the call's counters: a memset in front of it is a launch of its own)
__global__ __launch_bounds__(256) void k_qt_sums_bottom(const QTree qt, const uint8_t* __restrict__ rgb, QSum* __restrict__ sums, const int T,
                                                        u64* __restrict__ misc) {
  __shared__ QFold f;
  if (blockIdx.x == 0 && threadIdx.x < 2) misc[threadIdx.x] = 0ull;
  qt_fold<true>(qt, rgb, sums, qt.D, T, f);
}
__global__ __launch_bounds__(256) void k_qt_sums_up(const QTree qt, QSum* __restrict__ sums, const int L, const int T) {
  __shared__ QFold f;
  qt_fold<false>(qt, nullptr, sums, L, T, f);
}

// ---- decision ------------------------------------------------------------------------------------------------

// Error of a node as the reference defines it (quad_tree.cu:83-88), from the exact statistics, in fp64:
//   mse_c = (n * Q_c - S_c^2) / n^2;  error = (0.2989 mse_r + 0.5870 mse_g + 0.1140 mse_b) * (W * H) / 9e7
// and a bound on |fp32 value of computeError - exact value|.  With u = 2^-24, K = ceil(n / 256) terms per strided chain:
//   mean:  8 tree additions + 1 division               |mean~ - mean| <= 9u * 255            < 1.4e-4
//   diff:  d~_i = d_i + c + r_i, c the common mean shift, |r_i| <= 255u = 1.6e-5
//   sum of squares: sum (d_i + c + r_i)^2 - sum d_i^2 = 2 sum d_i r_i + sum (c + r_i)^2   (sum d_i = 0 exactly)
//                   <= n (3.2e-5 sqrt(mse) + 3e-8);  chain + tree + division + weighting: relative (K + 20) u
// band = 4 x that bound: well under 0.1 % of the threshold for the shipped parameters, so the literal kernel sees a
// handful of nodes per frame.
__device__ __forceinline__ void qt_exact_error(const QTree& qt, const QSum& s, const int n, double& err, double& band) {
  const double dn = (double) n, inv_n2 = 1.0 / (dn * dn);
  double mse[3], mx = 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const u64 num = (u64) n * s.q[k] - (u64) s.s[k] * (u64) s.s[k];  // exact: both products < 2^60
    mse[k] = (double) num * inv_n2;
    mx = mse[k] > mx ? mse[k] : mx;
  }
  const double scale = (double) ((float) (qt.W * qt.H)) / 90000000.0;
  err = ((double) 0.2989f * mse[0] + (double) 0.5870f * mse[1] + (double) 0.1140f * mse[2]) * scale;
  const double u = 5.9604644775390625e-8, K = (double) ((n + kQtThreads - 1) / kQtThreads);
  band = 4.0 * ((K + 32.0) * u * err + scale * (3.2e-5 * sqrt(mx) + 3e-8));
}

__global__ __launch_bounds__(256) void k_qt_decide(const QTree qt, const float thr, const QSum* __restrict__ sums, const int force_literal,
                                                   u32* __restrict__ flags, u32* __restrict__ unc_list, u32* __restrict__ unc_count,
                                                   u64* __restrict__ tile_sums, const u32 tiles) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < tiles) tile_sums[i] = 0ull;  // k_qt_emit adds its workgroups' sums of marks to them (the first half of the scan)
  if (i >= qt.total) return;
  const int l = qt_level_of(qt, i);
  const QRect r = qt_rect(qt, l, i - qt_level_offset(l));
  const int n = r.w * r.h;
  u32 flag;
  if (n == 0) flag = kQtDead;
  else if (l == qt.D || r.w / 2 <= qt.min_px || r.h / 2 <= qt.min_px) flag = kQtLeaf;  // leaf whatever the error says (quad_tree.cu:133-149)
  else {
    double err, band;
    qt_exact_error(qt, sums[i], n, err, band);
    const double t = (double) thr;
    if (force_literal || !(err > t + band || err < t - band)) {
      flag = kQtUncertain;
      unc_list[atomicAdd(unc_count, 1u)] = i;  // order is irrelevant: the literal kernel writes flags[i]
    } else flag = err > t ? kQtSplit : kQtLeaf;
  }
  flags[i] = flag;
}

// ---- the reference's arithmetic, literally ----------------------------------------------------------------------

// node with <= 256 pixels, one wave: lane holds the reference's threads lane, lane + 64, lane + 128, lane + 192 (each
// owns at most one pixel), so the first two tree levels are lane-local and the rest a shift-down fold
__device__ __forceinline__ float qt_wave_fold(float a0, float a1, float a2, float a3) {
  a0 += a2;  // stride 128
  a1 += a3;
  a0 += a1;  // stride 64
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) a0 += __shfl_down(a0, s);  // lanes >= s hold garbage that no lane < s / 2 reads
  return __shfl(a0, 0);
}

__device__ __forceinline__ float qt_error_wave(const uint8_t* __restrict__ rgb, const int cols, const float norm, const QRect n) {
  const int count = n.w * n.h, lane = (int) lane_id();
  float p[3][4];
  bool has[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int idx = lane + 64 * j;
    has[j] = idx < count;
    const int ly = has[j] ? idx / n.w : 0, lx = has[j] ? idx - ly * n.w : 0;
    const uint8_t* px = rgb + ((size_t) (n.y0 + ly) * cols + (n.x0 + lx)) * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) p[k][j] = has[j] ? (float) px[k] : 0.f;
  }
  float fin[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float mean = qt_wave_fold(p[k][0], p[k][1], p[k][2], p[k][3]) / (float) count;
    float sq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float d = p[k][j] - mean;
      sq[j] = has[j] ? d * d : 0.f;
    }
    fin[k] = qt_wave_fold(sq[0], sq[1], sq[2], sq[3]) / (float) count;
  }
  const float error = fin[0] * 0.2989f + fin[1] * 0.5870f + fin[2] * 0.1140f;
  return error * norm / 90000000.0f;
}

// any node, the whole workgroup of 256 = computeError as written (quad_tree.cu:6-90)
__device__ __forceinline__ float qt_error_wg(const uint8_t* __restrict__ rgb, const int cols, const float norm, const QRect n,
                                             float (*sh)[kQtThreads]) {
  const int t = (int) threadIdx.x, count = n.w * n.h;
  const int dq = kQtThreads / n.w, dr = kQtThreads - dq * n.w;  // idx += 256 as (row, column) steps
  float mean[3];
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    float acc[3] = {0.f, 0.f, 0.f};
    int ly = t / n.w, lx = t - ly * n.w;
    for (int idx = t; idx < count; idx += kQtThreads) {
      const uint8_t* px = rgb + ((size_t) (n.y0 + ly) * cols + (n.x0 + lx)) * 3;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float v = (float) px[k];
        if (pass == 0) acc[k] += v;
        else {
          const float d = v - mean[k];
          acc[k] += d * d;
        }
      }
      lx += dr;
      ly += dq;
      if (lx >= n.w) { lx -= n.w; ly++; }
    }
    __syncthreads();  // the previous pass's readers are done with sh
#pragma unroll
    for (int k = 0; k < 3; k++) sh[k][t] = acc[k];
    __syncthreads();
    for (int stride = kQtThreads / 2; stride > 0; stride >>= 1) {
      if (t < stride) {
#pragma unroll
        for (int k = 0; k < 3; k++) sh[k][t] += sh[k][t + stride];
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 3; k++) mean[k] = sh[k][0] / (float) count;  // pass 1: the three mse_final values
  }
  const float error = mean[0] * 0.2989f + mean[1] * 0.5870f + mean[2] * 0.1140f;
  return error * norm / 90000000.0f;
}

__global__ __launch_bounds__(256) void k_qt_literal(const QTree qt, const uint8_t* __restrict__ rgb, const float thr,
                                                    const u32* __restrict__ unc_list, const u32* __restrict__ unc_count, u32* __restrict__ flags) {
  __shared__ float sh[3][kQtThreads];
  const u32 n_unc = *unc_count;
  const float norm = (float) (qt.W * qt.H);
  const int wave = (int) (threadIdx.x >> 6);
  for (u32 base = blockIdx.x * 4u; base < n_unc; base += gridDim.x * 4u) {
    // four listed nodes per round: small ones one wave each, large ones by the whole workgroup (uniform control flow)
    {
      const u32 e = base + (u32) wave;
      if (e < n_unc) {
        const u32 i = unc_list[e];
        const int l = qt_level_of(qt, i);
        const QRect r = qt_rect(qt, l, i - qt_level_offset(l));
        if (r.w * r.h <= kQtThreads) {
          const float err = qt_error_wave(rgb, qt.W, norm, r);
          if (lane_id() == 0) flags[i] = err <= thr ? kQtLeaf : kQtSplit;
        }
      }
    }
    for (u32 e = base; e < base + 4u && e < n_unc; e++) {
      const u32 i = unc_list[e];
      const int l = qt_level_of(qt, i);
      const QRect r = qt_rect(qt, l, i - qt_level_offset(l));
      if (r.w * r.h <= kQtThreads) continue;
      const float err = qt_error_wg(rgb, qt.W, norm, r, sh);
      if (threadIdx.x == 0) flags[i] = err <= thr ? kQtLeaf : kQtSplit;
    }
  }
}

// ---- leaves and seeds ----------------------------------------------------------------------------------------

// processNodesKernel (gaussian_data_structures.cu:5-56) for one leaf; false = no seed
__device__ __forceinline__ bool splat_seed_of(const Cam& c, const Map& m, const Tab& t, const float* __restrict__ depth,
                                              const uint8_t* __restrict__ rgb, const QRect n, mrh_splat_seed& out) {
  const float p2x = (float) n.x0 + 0.5f * (float) n.w, p2y = (float) n.y0 + 0.5f * (float) n.h;
  const int px = f2i(p2x + 0.5f), py = f2i(p2y + 0.5f);
  if (px < 0 || py < 0 || px >= c.cols || py >= c.rows) return false;
  const float d = depth[(size_t) py * c.cols + px];
  if (d < c.min_depth) return false;
  const f3 center = se3_apply(c.R, c.t, inverse_projection(c, (u32) py, (u32) px, d));
  const Neigh none = neigh_none();
  const VoxSample v = get_voxel_f(m, t, none, center);
  if ((v.rgbw >> 24) != 1u) return false;  // a miss reads as weight 0
  const float half_w = 0.5f * (float) n.w, half_h = 0.5f * (float) n.h;
  const float scale = (d * sqrtf(half_w * half_w + half_h * half_h)) / c.fx;
  if (scale <= 0.0f) return false;
  out.p[0] = center.x; out.p[1] = center.y; out.p[2] = center.z;
  out.scale = scale;
  const uint8_t* q = rgb + ((size_t) py * c.cols + px) * 3;
  out.rgb[0] = q[0]; out.rgb[1] = q[1]; out.rgb[2] = q[2]; out.pad = 0;
  return true;
}

// marks[i] = leaf | seed << 32 for the scan; seeds are parked at their potential index
__global__ __launch_bounds__(256) void k_qt_emit(const QTree qt, const Cam c, const Map m, const Tab t, const float* __restrict__ depth,
                                                 const uint8_t* __restrict__ rgb, const u32* __restrict__ flags, u64* __restrict__ marks,
                                                 mrh_splat_seed* __restrict__ parked, u64* __restrict__ tile_sums) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  u64 mark = 0;
  if (i < qt.total && flags[i] == kQtLeaf) {
    const int l = qt_level_of(qt, i);
    const u32 path = i - qt_level_offset(l);
    bool live = true;
    for (int a = l - 1; a >= 0 && live; a--) live = flags[qt_level_offset(a) + (path >> (2 * (l - a)))] == kQtSplit;
    if (live) {
      mark = 1ull;
      mrh_splat_seed s;
      if (splat_seed_of(c, m, t, depth, rgb, qt_rect(qt, l, path), s)) {
        parked[i] = s;
        mark |= 1ull << 32;
      }
    }
  }
  if (i < qt.total) marks[i] = mark;
  // this workgroup's share of its scan tile's sum (kChainTile = 16 workgroups): (leaves, seeds) as two 32-bit counts in one word
  __shared__ u64 s_sum[4];
  u64 acc = mark;
  for (int off = 32; off > 0; off >>= 1) acc += (u64) __shfl_down((unsigned long long) acc, off);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const u64 all = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    if (all) atomicAdd((unsigned long long*) &tile_sums[(blockIdx.x * 256u) / kChainTile], (unsigned long long) all);
  }
}

__global__ __launch_bounds__(256) void k_qt_scatter(const QTree qt, const u64* __restrict__ marks, const u64* __restrict__ pos,
                                                    const mrh_splat_seed* __restrict__ parked, mrh_qtree_leaf* __restrict__ leaves,
                                                    mrh_splat_seed* __restrict__ seeds, const u32 seed_cap, const u64* __restrict__ misc,
                                                    u64* __restrict__ host_out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= qt.total) return;
  const u64 mark = marks[i], at = pos[i];
  if (i == qt.total - 1) {  // for the host (pinned): (leaves | seeds << 32) of the whole tree, the literal evaluations of this call
    host_out[0] = at + mark;
    host_out[1] = misc[1];
  }
  if (!(mark & 1ull)) return;
  const int l = qt_level_of(qt, i);
  const QRect r = qt_rect(qt, l, i - qt_level_offset(l));
  mrh_qtree_leaf o;
  o.x0 = r.x0; o.y0 = r.y0; o.width = r.w; o.height = r.h;
  leaves[(u32) at] = o;
  if ((mark >> 32) && (u32) (at >> 32) < seed_cap) seeds[at >> 32] = parked[i];  // `seeds` is pinned host memory (beyond the cap the call fails)
}

}  // namespace mrh
