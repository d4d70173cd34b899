// mrh_mesh.h — MeshExtractor::processTriangles on the device.
//
// Reference (mesh_extractor.cpp:9-76, :156-259): a single CPU thread walks the triangle soup through two
// std::unordered_maps — vertex merge (exact position, or floor(v / eps) cells; the first occurrence keeps index and
// colour), then degenerate faces dropped and repeated faces dropped keeping the first.  At a million triangles that
// is ~0.4 s of host time after a 12 ms extraction.  "First occurrence wins" is a statement about ORDER, so it maps
// onto stable sorts and scans and gives the same arrays, element for element:
//
//   vertices  key = 3 x 32 bits (position bits, or the eps cell), value = soup index
//             two stable LSD radix passes (rocPRIM) -> equal keys contiguous, soup indices ascending inside a run
//             run head = representative;   is_first = (rep == self);   new index = exclusive scan of is_first
//             -> V / C in order of first occurrence, face corner -> index of its representative
//   faces     degenerate = two equal corners;  the same sort on the (a, b, c) index triples, run head = first
//             occurrence, keep = head && !degenerate, position = exclusive scan of keep in soup order
//
// NaN positions never compare equal in the reference (Vector3dEqual): every vertex with a NaN coordinate is its own
// run head.  Used by mrh_extract_triangles (soup already on the device) and mrh_process_triangles (soup uploaded).
#pragma once

#include <rocprim/rocprim.hpp>

#include "mrh_device.h"

namespace mrh {

// soup vertex i = corner (i % 3) of triangle (i / 3); mrh_triangle = 3 x {p[3], c[3]} floats
__global__ __launch_bounds__(256) void k_mesh_vertex_keys(const float* __restrict__ soup, const u32 n, const double eps, const double inv_eps,
                                                          u32* __restrict__ kx, u64* __restrict__ kyz, u32* __restrict__ idx,
                                                          u32* __restrict__ nanflag) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* v = soup + (size_t) i * 6;
  const float p[3] = {v[0], v[1], v[2]};
  u32 k[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (eps == 0.0) k[a] = __float_as_uint(p[a]);  // (double) p is injective on the bit pattern, -0 and +0 stay apart
    else k[a] = (u32) (int) floor((double) p[a] * inv_eps);
  }
  kx[i] = k[0];
  kyz[i] = ((u64) k[1] << 32) | (u64) k[2];
  idx[i] = i;
  nanflag[i] = (p[0] != p[0] || p[1] != p[1] || p[2] != p[2]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_gather_u32(const u32* __restrict__ src, const u32* __restrict__ idx, const u32 n, u32* __restrict__ dst) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[j] = src[idx[j]];
}

// sorted position j: head if its 96-bit key differs from the previous one (or the element never merges)
__global__ __launch_bounds__(256) void k_mesh_heads(const u32* __restrict__ kx, const u64* __restrict__ kyz, const u32* __restrict__ never,
                                                    const u32* __restrict__ order, const u32 n, u32* __restrict__ headpos) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const u32 i = order[j];
  bool head = j == 0 || (never && never[i]);
  if (!head) {
    const u32 p = order[j - 1];
    head = kx[i] != kx[p] || kyz[i] != kyz[p];
  }
  headpos[j] = head ? j : 0u;
}

__global__ __launch_bounds__(256) void k_mesh_rep(const u32* __restrict__ order, const u32* __restrict__ headpos, const u32 n,
                                                  u32* __restrict__ rep, u32* __restrict__ is_first) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const u32 i = order[j], r = order[headpos[j]];
  rep[i] = r;
  is_first[i] = (r == i) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_mesh_emit_vertices(const float* __restrict__ soup, const u32* __restrict__ rep,
                                                            const u32* __restrict__ is_first, const u32* __restrict__ vid, const u32 n,
                                                            double* __restrict__ V, double* __restrict__ C, u32* __restrict__ corner) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  corner[i] = vid[rep[i]];
  if (is_first[i]) {
    const float* v = soup + (size_t) i * 6;
    const size_t o = (size_t) vid[i] * 3;
    V[o] = (double) v[0]; V[o + 1] = (double) v[1]; V[o + 2] = (double) v[2];
    C[o] = (double) v[3]; C[o + 1] = (double) v[4]; C[o + 2] = (double) v[5];
  }
}

__global__ __launch_bounds__(256) void k_mesh_face_keys(const u32* __restrict__ corner, const u32 nt, u32* __restrict__ ka, u64* __restrict__ kbc,
                                                        u32* __restrict__ idx, u32* __restrict__ degenerate) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const u32 a = corner[3 * t], b = corner[3 * t + 1], c = corner[3 * t + 2];
  ka[t] = a;
  kbc[t] = ((u64) b << 32) | (u64) c;
  idx[t] = t;
  degenerate[t] = (a == b || a == c || b == c) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_mesh_face_keep(const u32* __restrict__ order, const u32* __restrict__ headpos,
                                                        const u32* __restrict__ degenerate, const u32 nt, u32* __restrict__ keep) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nt) return;
  const u32 t = order[j];
  keep[t] = (headpos[j] == j && !degenerate[t]) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_mesh_emit_faces(const u32* __restrict__ corner, const u32* __restrict__ keep, const u32* __restrict__ fpos,
                                                         const u32 nt, int* __restrict__ F) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt || !keep[t]) return;
  const size_t o = (size_t) fpos[t] * 3;
  F[o] = (int) corner[3 * t]; F[o + 1] = (int) corner[3 * t + 1]; F[o + 2] = (int) corner[3 * t + 2];
}

// ---- host driver ------------------------------------------------------------------------------------------

struct MeshScratch {
  void* base = nullptr;
  size_t bytes = 0, used = 0;
  template <typename T>
  T* take(size_t n) {
    used = (used + 255) & ~(size_t) 255;
    T* p = (T*) ((char*) base + used);
    used += n * sizeof(T);
    return p;
  }
};

// stable sort of `n` elements by the 96-bit key (hi32, lo64): order_out[j] = element at sorted position j.
// order_in must be 0..n-1 ascending.  All buffers device; tmp/tmp_bytes is rocPRIM's scratch (sized by the caller
// through mesh_sort_tmp_bytes).
inline size_t mesh_sort_tmp_bytes(const u32 n) {
  size_t a = 0, b = 0, c = 0, d = 0;
  (void) rocprim::radix_sort_pairs(nullptr, a, (u64*) nullptr, (u64*) nullptr, (u32*) nullptr, (u32*) nullptr, n);
  (void) rocprim::radix_sort_pairs(nullptr, b, (u32*) nullptr, (u32*) nullptr, (u32*) nullptr, (u32*) nullptr, n);
  (void) rocprim::inclusive_scan(nullptr, c, (u32*) nullptr, (u32*) nullptr, n, rocprim::maximum<u32>());
  (void) rocprim::exclusive_scan(nullptr, d, (u32*) nullptr, (u32*) nullptr, 0u, n, rocprim::plus<u32>());
  size_t m = a > b ? a : b;
  m = m > c ? m : c;
  return m > d ? m : d;
}

inline hipError_t mesh_sort96(void* tmp, size_t tmp_bytes, const u32* hi, const u64* lo, u32* order_in, u32* order_mid, u32* order_out,
                              u64* lo_sorted, u32* hi_gathered, u32* hi_sorted, const u32 n, hipStream_t s) {
  hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, lo, lo_sorted, order_in, order_mid, n, 0, 64, s);
  if (e != hipSuccess) return e;
  const u32 grid = (n + 255) / 256;
  k_gather_u32<<<grid, 256, 0, s>>>(hi, order_mid, n, hi_gathered);
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, hi_gathered, hi_sorted, order_mid, order_out, n, 0, 32, s);
}

// per-block triangle runs of several ranks -> one buffer in canonical block order (mrh_process_triangle_runs): run r copies
// count triangles (72 B = 4.5 x uint4, moved as 9 x 8-byte words) from src to dst
__global__ __launch_bounds__(256) void k_permute_runs(const ulonglong2* __restrict__ runs, const int n_runs, const uint4* __restrict__ in, uint4* __restrict__ out) {
  for (int r = blockIdx.x; r < n_runs; r += gridDim.x) {
    const ulonglong2 run = runs[r];
    const size_t src = (size_t) run.x, dst = (size_t) (run.y & ((1ull << 40) - 1)), cnt = (size_t) (run.y >> 40);
    const unsigned long long* a = (const unsigned long long*) in + src * 9;
    unsigned long long* b = (unsigned long long*) out + dst * 9;
    for (size_t w = threadIdx.x; w < cnt * 9; w += 256) b[w] = a[w];
  }
}

}  // namespace mrh
