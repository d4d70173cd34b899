// mrh_mesh.h — MeshExtractor::processTriangles on the device.
//
// Reference (mesh_extractor.cpp:9-76, :156-259): a single CPU thread walks the triangle soup through two
// std::unordered_maps — vertex merge (exact position, or floor(v / eps) cells; the first occurrence keeps index and
// colour), then degenerate faces dropped and repeated faces dropped keeping the first.  At a million triangles that
// is ~0.4 s of host time after a millisecond of extraction.  "First occurrence wins" means: the representative of a
// key is the SMALLEST soup index that carries it.  That is a min-reduction per key, which an open-address table of
// soup indices computes in one pass with no ordering at all:
//
//   vertices  slot = hash(96-bit key) ...: empty -> CAS my index in; occupied by index j -> compare MY key with the key of
//             soup vertex j (12 bytes, read where the soup lies): equal -> atomicMin(slot, my index), else next slot.
//             Only indices of ONE key ever replace each other in a slot, so the comparison through whichever index is
//             stored is stable.  Second pass: rep[i] = the slot's final index; is_first = (rep == self); new index =
//             exclusive scan of is_first in soup order -> V / C in order of first occurrence.
//   faces     degenerate = two equal corners; the same table on the (a, b, c) index triples; keep = (rep == self) &&
//             !degenerate; position = exclusive scan of keep in soup order.
//
// Round 2 did this with two stable 96-bit LSD radix sorts (12 + 12 passes over the soup per extraction); the arrays are
// the same, element for element (tests compare with the host restatement and the oracle), at a fraction of the passes.
// NaN positions never compare equal in the reference (Vector3dEqual): a vertex with a NaN coordinate stays out of the
// table and is its own representative.  Used by mrh_extract_triangles (soup already on the device), the run merge of a
// sharded extraction and mrh_process_triangles (soup uploaded).
#pragma once


#include "mrh_device.h"

namespace mrh {

constexpr u32 kMeshEmpty = 0xFFFFFFFFu;
struct Key96 { u32 a, b, c; };
__device__ __forceinline__ bool key_eq(const Key96 x, const Key96 y) { return x.a == y.a && x.b == y.b && x.c == y.c; }
__device__ __forceinline__ u32 key_hash(const Key96 k) {
  u64 h = ((u64) k.a << 32 | k.b) * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  h += (u64) k.c * 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32; h *= 0x94D049BB133111EBull; h ^= h >> 29;
  return (u32) h;
}
// soup vertex i = corner (i % 3) of triangle (i / 3); mrh_triangle = 3 x {p[3], c[3]} floats.
// Key: the position's bit pattern ((double) p is injective on it, -0 and +0 stay apart), or the eps cell (mesh_extractor.cpp:196-203)
__device__ __forceinline__ Key96 vertex_key(const float* __restrict__ soup, const u32 i, const double eps, const double inv_eps, bool& has_nan) {
  const float* v = soup + (size_t) i * 6;
  const float p0 = v[0], p1 = v[1], p2 = v[2];
  has_nan = p0 != p0 || p1 != p1 || p2 != p2;
  Key96 k;
  if (eps == 0.0) { k.a = __float_as_uint(p0); k.b = __float_as_uint(p1); k.c = __float_as_uint(p2); }
  else { k.a = (u32) (int) floor((double) p0 * inv_eps); k.b = (u32) (int) floor((double) p1 * inv_eps); k.c = (u32) (int) floor((double) p2 * inv_eps); }
  return k;
}
__device__ __forceinline__ Key96 face_key(const u32* __restrict__ corner, const u32 t) {
  Key96 k;
  k.a = corner[3 * t]; k.b = corner[3 * t + 1]; k.c = corner[3 * t + 2];
  return k;
}

// A vertex of the soup occurs about six times, and most of its occurrences lie a few hundred entries apart (the voxels that
// share the edge belong to one block, and a block's triangles are contiguous): a workgroup first finds the first occurrences
// inside its own tile of kMeshTile entries in an LDS table (same min-index rule), and only those go to the global table — a
// later occurrence inside the tile cannot lower the minimum its tile's first occurrence already stands for.  The tile-local
// representative is left in rep[] for k_mesh_vertex_rep, which uses the same tiles.  (Round 4: 85 + 22 us -> see DESIGN §4.3
// for 1.6 M soup vertices; every soup vertex used to pay two or three dependent trips to the memory-side atomics.)
constexpr int kMeshTile = 512;
constexpr u32 kMeshLocalSlots = 1024;
__global__ __launch_bounds__(kMeshTile) void k_mesh_vertex_insert(const float* __restrict__ soup, const u32 n, const double eps, const double inv_eps,
                                                                  u32* __restrict__ table, const u32 mask, u32* __restrict__ rep) {
  __shared__ u32 s_ka[kMeshTile], s_kb[kMeshTile], s_kc[kMeshTile];
  __shared__ u32 s_tab[kMeshLocalSlots];
  const u32 tl = threadIdx.x, i = blockIdx.x * kMeshTile + tl;
  for (u32 k = tl; k < kMeshLocalSlots; k += kMeshTile) s_tab[k] = kMeshEmpty;
  bool nan_i = true;
  Key96 key;
  key.a = key.b = key.c = 0;
  if (i < n) key = vertex_key(soup, i, eps, inv_eps, nan_i);
  s_ka[tl] = key.a; s_kb[tl] = key.b; s_kc[tl] = key.c;
  __syncthreads();
  const u32 h = key_hash(key);
  u32 myslot = kMeshEmpty;
  if (!nan_i) {
    u32 s = (h >> 20) & (kMeshLocalSlots - 1);
    for (;;) {
      u32 cur = __hip_atomic_load(&s_tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (cur == kMeshEmpty) {
        cur = atomicCAS(&s_tab[s], kMeshEmpty, tl);
        if (cur == kMeshEmpty) { myslot = s; break; }
      }
      if (s_ka[cur] == key.a && s_kb[cur] == key.b && s_kc[cur] == key.c) {  // only indices of ONE key ever replace each other in a slot
        if (tl < cur) atomicMin(&s_tab[s], tl);
        myslot = s;
        break;
      }
      s = (s + 1) & (kMeshLocalSlots - 1);
    }
  }
  __syncthreads();
  if (i >= n) return;
  const u32 lr = myslot != kMeshEmpty ? s_tab[myslot] : tl;
  rep[i] = blockIdx.x * kMeshTile + lr;
  if (lr != tl || nan_i) return;  // a later occurrence within the tile, or a NaN position (its own representative, never in a table)
  u32 s = h & mask;
  for (;;) {
    // the table is at most half full and most probes meet an empty slot: the CAS goes first (one trip to the memory-side
    // atomics instead of a load and a CAS)
    const u32 cur = atomicCAS(&table[s], kMeshEmpty, i);
    if (cur == kMeshEmpty) return;
    bool nan_c;
    if (key_eq(vertex_key(soup, cur, eps, inv_eps, nan_c), key)) {
      if (i < cur) atomicMin(&table[s], i);
      return;
    }
    s = (s + 1) & mask;
  }
}
// exclusive rank of `flag` among the threads of the workgroup (NW waves) in thread order, and the workgroup's total
template <int NW>
__device__ __forceinline__ u32 tile_rank(const bool flag, u32* s_w, u32& total) {
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u64 bal = __ballot(flag);
  if (lane == 0) s_w[wave] = (u32) __popcll(bal);
  __syncthreads();
  u32 before = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) { if ((u32) w < wave) before += s_w[w]; total += s_w[w]; }
  return before + (u32) __popcll(bal & ((1ull << lane) - 1ull));
}
// The new index of a vertex / the position of a kept face = exclusive scan of a 0 / 1 flag in soup order.  The kernels that
// produce the flags work in tiles anyway: they leave the rank INSIDE the tile per element and one count per tile, one workgroup
// scans the tile counts (k_tile_scan; 3 123 tiles for 1.6 M soup vertices) and the consumers add the two — where rocPRIM's
// device scan took two launches and a pass over the flags each time (2 x 14 us per extraction).
__global__ __launch_bounds__(1024) void k_tile_scan(const u32* __restrict__ counts, const u32 n, u32* __restrict__ offsets, u64* __restrict__ total_out) {
  __shared__ u32 s_w[16];
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 carry = 0;
  for (u32 base = 0; base < n; base += 1024) {
    const u32 i = base + threadIdx.x;
    const u32 c = i < n ? counts[i] : 0u;
    u32 incl = c;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 o = __shfl_up(incl, off);
      if ((int) lane >= off) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    u32 before = 0, all = 0;
    for (u32 w = 0; w < 16; w++) { if (w < wave) before += s_w[w]; all += s_w[w]; }
    if (i < n) offsets[i] = carry + before + incl - c;
    carry += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = (u64) carry;
}

// rep[] comes in holding the tile-local representatives and leaves holding the global ones; vloc[i] = first occurrences before i
// inside i's tile, tcount[tile] = first occurrences of the tile
__global__ __launch_bounds__(kMeshTile) void k_mesh_vertex_rep(const float* __restrict__ soup, const u32 n, const double eps, const double inv_eps,
                                                               const u32* __restrict__ table, const u32 mask, u32* __restrict__ rep, u32* __restrict__ vloc,
                                                               u32* __restrict__ tcount) {
  __shared__ u32 s_rep[kMeshTile];
  __shared__ u32 s_w[kMeshTile / 64];
  const u32 tl = threadIdx.x, i = blockIdx.x * kMeshTile + tl;
  const u32 lr = i < n ? rep[i] : i;
  const bool local_first = i < n && lr == i;
  u32 r = i;
  if (local_first) {
    bool nan_i;
    const Key96 key = vertex_key(soup, i, eps, inv_eps, nan_i);
    if (!nan_i) {
      u32 s = key_hash(key) & mask;
      for (;;) {
        const u32 cur = table[s];  // never empty before the key's own slot: this vertex was inserted
        bool nan_c;
        if (key_eq(vertex_key(soup, cur, eps, inv_eps, nan_c), key)) { r = cur; break; }
        s = (s + 1) & mask;
      }
    }
  }
  s_rep[tl] = r;
  __syncthreads();
  if (i < n && !local_first) r = s_rep[lr - blockIdx.x * kMeshTile];
  u32 total;
  const u32 rank = tile_rank<kMeshTile / 64>(i < n && r == i, s_w, total);
  if (tl == 0) tcount[blockIdx.x] = total;
  if (i >= n) return;
  rep[i] = r;
  vloc[i] = rank;
}

// T = float: V and C leave the device as the fp32 values marching cubes computed (the host widens them while the link is still
// busy, mrh_capi.hip: k_stage_out / widen_from_staging); T = double: widened here (MRH_MESH_F64_LINK=1, the round-3 path)
template <typename T>
__global__ __launch_bounds__(256) void k_mesh_emit_vertices(const float* __restrict__ soup, const u32* __restrict__ rep,
                                                            const u32* __restrict__ vloc, const u32* __restrict__ toff, const u32 n,
                                                            T* __restrict__ V, T* __restrict__ C, u32* __restrict__ corner) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 r = rep[i];
  const u32 id = toff[r / kMeshTile] + vloc[r];  // new index of the representative (k_mesh_vertex_rep's tiles + k_tile_scan)
  corner[i] = id;
  if (r == i) {
    const float* v = soup + (size_t) i * 6;
    const size_t o = (size_t) id * 3;
    V[o] = (T) v[0]; V[o + 1] = (T) v[1]; V[o + 2] = (T) v[2];
    C[o] = (T) v[3]; C[o + 1] = (T) v[4]; C[o + 2] = (T) v[5];
  }
}

__global__ __launch_bounds__(256) void k_mesh_face_insert(const u32* __restrict__ corner, const u32 nt, u32* __restrict__ table, const u32 mask) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const Key96 key = face_key(corner, t);
  u32 s = key_hash(key) & mask;
  for (;;) {
    const u32 cur = atomicCAS(&table[s], kMeshEmpty, t);  // CAS first: repeated faces are rare, almost every probe claims its slot
    if (cur == kMeshEmpty) return;
    if (key_eq(face_key(corner, cur), key)) {
      if (t < cur) atomicMin(&table[s], t);
      return;
    }
    s = (s + 1) & mask;
  }
}
// keep = first occurrence of its (a, b, c) triple and not degenerate (mesh_extractor.cpp:57-75, :156-178)
__global__ __launch_bounds__(256) void k_mesh_face_keep(const u32* __restrict__ corner, const u32 nt, const u32* __restrict__ table, const u32 mask,
                                                        u32* __restrict__ floc, u32* __restrict__ fcount) {
  __shared__ u32 s_w[4];
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  bool kept = false;
  if (t < nt) {
    const Key96 key = face_key(corner, t);
    u32 s = key_hash(key) & mask, r = t;
    for (;;) {
      const u32 cur = table[s];
      if (key_eq(face_key(corner, cur), key)) { r = cur; break; }
      s = (s + 1) & mask;
    }
    const bool degenerate = key.a == key.b || key.a == key.c || key.b == key.c;
    kept = r == t && !degenerate;
  }
  u32 total;
  const u32 rank = tile_rank<4>(kept, s_w, total);
  if (threadIdx.x == 0) fcount[blockIdx.x] = total;
  if (t < nt) floc[t] = rank | (kept ? 0x80000000u : 0u);  // rank inside the 256-face tile | kept
}

__global__ __launch_bounds__(256) void k_mesh_emit_faces(const u32* __restrict__ corner, const u32* __restrict__ floc, const u32* __restrict__ foff,
                                                         const u32 nt, int* __restrict__ F) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const u32 fl = floc[t];
  if (!(fl & 0x80000000u)) return;
  const size_t o = (size_t) (foff[blockIdx.x] + (fl & 0x7FFFFFFFu)) * 3;
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
