"""Pins the CPU oracle against everything the reference's own tests hold for this path:
host-computable known answers from voxel_hash_utils.cuh, and the invariants asserted by
tests/test_hash_utils.cu, tests/test_projections.cu and tests/test_marching_cubes.cpp, restated on the
oracle's literal data structures.  (The reference stores no golden values for integrate / GC / marching
cubes, so those stay "parity unpinned against reference GPU output"; see oracle/mrh_oracle.c header.)"""
import ctypes as C

import numpy as np
import pytest

import parity_utils as pu
from mrhash_amd import capi, synth

HASH_ENTRY = np.dtype([("pos", "<i4", (3,)), ("offset", "<u4"), ("ptr", "<i4"), ("resolution", "<i4")])
FREE_ENTRY = -2


def _table(orc, eng, fn="orc_hash_table", dtype=HASH_ENTRY):
    f = getattr(orc, fn)
    f.restype = C.c_void_p
    f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    n = C.c_uint64()
    p = f(eng._ctx, C.byref(n))
    return np.frombuffer((C.c_char * (n.value * dtype.itemsize)).from_address(p), dtype=dtype).copy()


def _heap(orc, eng):
    orc.orc_heap_high.restype = C.c_void_p
    orc.orc_heap_high.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    n, ctr = C.c_uint64(), C.c_int()
    p = orc.orc_heap_high(eng._ctx, C.byref(n), C.byref(ctr))
    return np.frombuffer((C.c_char * (n.value * 4)).from_address(p), dtype=np.uint32).copy(), ctr.value


def test_struct_layouts():
    # voxel_hash_utils.cuh:8-64 (verified by host compile in SURVEY.md §8c): Voxel 12, HashEntry 24, Vertex 24, Triangle 72
    assert capi.VOXEL_DTYPE.itemsize == 12 and capi.VOXEL_DTYPE.fields["rgb"][1] == 8 and capi.VOXEL_DTYPE.fields["weight"][1] == 11
    assert HASH_ENTRY.itemsize == 24 and HASH_ENTRY.fields["ptr"][1] == 16
    assert capi.TRI_DTYPE.itemsize * 3 == 72


def test_known_answers_from_reference_host_functions(oracle):
    # captured by compiling voxel_hash_utils.cuh's __host__ functions
    oracle.orc_kat_voxel_to_block_index.restype = C.c_uint
    assert oracle.orc_kat_voxel_to_block_index(-9, 17, 3, 8) == 207
    assert oracle.orc_kat_voxel_to_block_index(-9, 17, 3, 4) == 67
    out = (C.c_int * 3)()
    oracle.orc_kat_delinearize(77, 8, out)
    assert list(out) == [5, 1, 1]


@pytest.mark.parametrize("block_size", [8, 4, 2])
def test_voxel_roundtrips(oracle, block_size):
    # tests/test_hash_utils.cu:40-163 VOXEL.*: world -> voxel -> world within 1e-4 at voxel size 1e-6 is not
    # representable for every point; the reference draws points of magnitude ~1e-5..1e-3. Same construction here.
    rng = np.random.default_rng(0)
    vs = np.float32(1e-6)
    oracle.orc_kat_world_to_voxel.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    for _ in range(200):
        p = rng.uniform(-1e-3, 1e-3, 3).astype(np.float32)
        out = (C.c_int * 3)()
        oracle.orc_kat_world_to_voxel(vs, p.ctypes.data_as(C.POINTER(C.c_float)), out)
        back = np.array(list(out), dtype=np.float32) * vs
        assert np.all(np.abs(back - p) <= 1e-4)
    # world -> voxel-index -> local position round trip (delinearize o linearize == id)
    for idx in range(block_size ** 3):
        o = (C.c_int * 3)()
        oracle.orc_kat_delinearize(idx, block_size, o)
        assert o[2] * block_size * block_size + o[1] * block_size + o[0] == idx


def test_voxel_to_block_is_floor_division(oracle):
    # vhu.cuh:75-103 goes through float world coordinates; for |pw| < 256 m it must equal floor(v / 8)
    oracle.orc_kat_voxel_to_block.argtypes = [C.POINTER(C.c_int), C.c_float, C.POINTER(C.c_int)]
    rng = np.random.default_rng(1)
    for vs in (0.01, 0.02, 0.005, 0.2):
        v = rng.integers(-20000, 20000, size=(500, 3)).astype(np.int32)
        v[:8] = [[0, 0, 0], [7, 8, -1], [-8, -9, 15], [-7, -16, 16], [63, 64, 65], [-64, -63, -65], [1, -1, 0], [8, -8, 7]]
        for row in v:
            if np.max(np.abs(row)) * vs >= 256:
                continue
            out = (C.c_int * 3)()
            oracle.orc_kat_voxel_to_block((C.c_int * 3)(*row.tolist()), vs, out)
            assert list(out) == [int(np.floor(x / 8)) for x in row], (row, vs)


def test_buffer_initialisation(oracle):
    # tests/test_hash_utils.cu:306-376 HASHTABLE.BufferInitialization
    N = 4096
    e = pu.make_engine(oracle, synth.CFG1, synth.CFG1_PARAMS, N)
    heap, ctr = _heap(oracle, e)
    assert ctr == N - 1
    assert np.array_equal(heap, N - 1 - np.arange(N, dtype=np.uint32))
    for fn in ("orc_hash_table", "orc_compact_table"):
        t = _table(oracle, e, fn)
        assert len(t) == N * 10
        assert np.all(t["pos"] == 0) and np.all(t["offset"] == 0) and np.all(t["ptr"] == FREE_ENTRY)
    oracle.orc_bucket_mutex.restype = C.c_void_p
    oracle.orc_bucket_mutex.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    n = C.c_uint64()
    p = oracle.orc_bucket_mutex(e._ctx, C.byref(n))
    assert np.all(np.frombuffer((C.c_char * (n.value * 4)).from_address(p), dtype=np.int32) == FREE_ENTRY)
    s = e.stats()
    assert s.last_compact_blocks == 0 and s.free_fine == N and s.free_coarse == 0


def test_heap_sanity_after_one_frame(oracle):
    # tests/test_hash_utils.cu:378-526 HASHTABLE.HeapSanityCheck, scaled down: 400x400 plane at 1 m, K = 400,
    # voxel 5 mm, truncation 0.02 + 0.01 d, weight 3 -> here 200x200, K = 200 (same geometry, 1/4 the rays)
    K = synth.Intrinsics(200.0, 200.0, 100.0, 100.0, 200, 200)
    N = 60000
    params = dict(synth.CFG1_PARAMS, sdf_truncation=0.02, sdf_truncation_scale=0.01, virtual_voxel_size=0.005,
                  integration_weight_sample=3, min_depth=0.01, max_depth=5.0)
    e = pu.make_engine(oracle, K, params, N)
    depth = np.full((K.rows, K.cols), 1.0, np.float32)
    rgb = np.zeros((K.rows, K.cols, 3), np.uint8)
    rgb[..., 0] = 255
    e.set_pose(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
    e.upload_depth(depth)
    e.upload_rgb(rgb)
    e.integrate(0)
    heap, ctr = _heap(oracle, e)
    n_free = ctr + 1
    free_ptrs = heap[:n_free]
    assert len(np.unique(free_ptrs)) == n_free  # free-heap pointers unique
    t = _table(oracle, e)
    occ = t[t["ptr"] != FREE_ENTRY]
    assert len(occ) > 0
    alloc_ptrs = occ["ptr"] // 512
    assert len(np.intersect1d(alloc_ptrs, free_ptrs)) == 0  # no allocated ptr also on the free heap
    assert len(np.unique(alloc_ptrs)) == len(occ)
    assert len(occ) + n_free == N  # every block is either free or allocated (no leak)
    assert len(np.unique(occ["pos"], axis=0)) == len(occ)  # no duplicate block positions
    assert e.stats().occupied_fine == len(occ)


def test_allocation_deletion(oracle):
    # tests/test_hash_utils.cu:192-304 HASHTABLE.AllocationDeletion: integrate -> zero all weights -> garbageCollect
    N = 20000
    e = pu.make_engine(oracle, synth.CFG1, dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=1000), N)
    f = synth.cfg1_plane()
    pu.feed(e, f, 0)
    before = e.stats()
    assert before.occupied_fine > 0
    compacted = before.last_compact_blocks
    oracle.orc_zero_all_weights(e._ctx)
    # a second frame with an empty depth image allocates nothing and integrates nothing; GC then sees weight 0 everywhere
    e.upload_depth(np.zeros_like(f.depth))
    e.integrate(1000)
    after = e.stats()
    assert after.free_fine + after.occupied_fine == N
    assert after.occupied_fine == 0  # everything that was compacted (in frustum) got freed
    oracle.orc_decisions.restype = C.c_void_p
    oracle.orc_decisions.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    n = C.c_uint64()
    p = oracle.orc_decisions(e._ctx, C.byref(n))
    dec = np.frombuffer((C.c_char * (n.value * 4)).from_address(p), dtype=np.int32)
    assert int((dec > 0).sum()) == compacted  # decisions > 0 == number of blocks that were compacted
    e.upload_depth(f.depth)


def test_inverse_projection_pinhole(oracle):
    # tests/test_projections.cu:41-107 INV_PROJECTION: 480x640, K = (517.3, 516.5, 318.6, 255.3): z == depth
    K = synth.Intrinsics(517.3, 516.5, 318.6, 255.3, 480, 640)
    e = pu.make_engine(oracle, K, dict(synth.CFG1_PARAMS, min_depth=0.1, max_depth=10.0))
    oracle.orc_inverse_projection.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_float, C.POINTER(C.c_float)]
    rng = np.random.default_rng(2)
    out = (C.c_float * 3)()
    for _ in range(2000):
        r, c = int(rng.integers(0, 480)), int(rng.integers(0, 640))
        d = np.float32(rng.uniform(0.2, 9.9))
        oracle.orc_inverse_projection(e._ctx, r, c, d, out)
        assert np.float32(out[2]) == d  # ASSERT_FLOAT_EQ


def test_projection_roundtrip_and_row_col_convention(oracle):
    # tests/test_projections.cu:109-141 PROJECTIONS.Dummy: project then inverse-project within 1e-2; pimg = (row, col)
    K = synth.Intrinsics(517.3, 516.5, 318.6, 255.3, 480, 640)
    e = pu.make_engine(oracle, K, dict(synth.CFG1_PARAMS, min_depth=0.1, max_depth=10.0))
    oracle.orc_inverse_projection.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_float, C.POINTER(C.c_float)]
    oracle.orc_project_point.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    rng = np.random.default_rng(3)
    n_ok = 0
    for _ in range(5000):
        r, c = int(rng.integers(0, 480)), int(rng.integers(0, 640))
        d = np.float32(rng.uniform(0.2, 9.9))
        p = (C.c_float * 3)()
        oracle.orc_inverse_projection(e._ctx, r, c, d, p)
        px = (C.c_int * 2)()
        ok = oracle.orc_project_point(e._ctx, p, 0, px)
        assert ok == 1
        assert abs(px[0] - r) <= 1 and abs(px[1] - c) <= 1  # (row, col) order, test_projections.cu:128-130
        q = (C.c_float * 3)()
        oracle.orc_inverse_projection(e._ctx, px[0], px[1], d, q)
        assert max(abs(q[i] - p[i]) for i in range(3)) <= d / 500.0 + 1e-2
        n_ok += 1
    assert n_ok == 5000
    # (-1, 0) after +0.5 truncates to pixel 0 (camera.cuh:137-140 quirk)
    px = (C.c_int * 2)()
    pt = (C.c_float * 3)(np.float32((-0.7 - 318.6) / 517.3), 0.0, 1.0)
    assert oracle.orc_project_point(e._ctx, pt, 0, px) == 1 and px[1] == 0


def _lidar_engine(oracle, min_depth, max_depth):
    # tests/test_projections.cu:155-161 / :192-197: K = (-1024 / 2pi, -128 / (pi / 2), 512, 64), 128 x 1024, Spherical
    fx, fy = np.float32(-1024 / (2 * np.pi)), np.float32(-128 / (np.pi / 2))
    e = capi.Engine(oracle, capi.Params(num_sdf_blocks=1024, **dict(synth.CFG1_PARAMS, min_depth=min_depth, max_depth=max_depth)))
    e.set_camera(float(fx), float(fy), 512.0, 64.0, 128, 1024, min_depth, max_depth, model=1)
    return e


def test_inverse_projection_spherical(oracle):
    """tests/test_projections.cu:143-188 INV_PROJECTION_LIDAR: the range of the back-projected point equals the depth
    (ASSERT_FLOAT_EQ: within 4 ulp) — with sin / cos from include/mrh_softmath.h instead of CUDA's (deviation D8)."""
    e = _lidar_engine(oracle, 0.0, 10.0)
    oracle.orc_inverse_projection.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_float, C.POINTER(C.c_float)]
    rng = np.random.default_rng(4)
    out = (C.c_float * 3)()
    worst = 0.0
    for _ in range(4000):
        r, c = int(rng.integers(0, 128)), int(rng.integers(0, 1024))
        d = np.float32(rng.uniform(0.05, 10.0))
        oracle.orc_inverse_projection(e._ctx, r, c, d, out)
        x, y, z = np.float32(out[0]), np.float32(out[1]), np.float32(out[2])
        rng_len = np.sqrt(np.float32(np.float32(x * x + y * y) + z * z), dtype=np.float32)
        worst = max(worst, abs(float(rng_len) - float(d)) / float(np.spacing(d)))
    assert worst <= 4.0, worst


def test_projection_roundtrip_spherical(oracle):
    """tests/test_projections.cu:190-222 PROJECTIONS_LIDAR.Dummy: random points in [-1, 1]^3, project (atan2 / asin), back-
    project the pixel with the point's range: every coordinate within 2e-2; pimg = (row, col)."""
    e = _lidar_engine(oracle, 0.2, 50.0)
    oracle.orc_inverse_projection.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_float, C.POINTER(C.c_float)]
    oracle.orc_project_point.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    rng = np.random.default_rng(5)
    n_ok = 0
    for _ in range(20000):
        p = rng.uniform(-1.0, 1.0, 3).astype(np.float32)
        pc = (C.c_float * 3)(*[float(v) for v in p])
        px = (C.c_int * 2)()
        if not oracle.orc_project_point(e._ctx, pc, 0, px):
            continue
        assert 0 <= px[0] < 128 and 0 <= px[1] < 1024
        rng_len = np.sqrt(np.float32(np.float32(p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]), dtype=np.float32)
        q = (C.c_float * 3)()
        oracle.orc_inverse_projection(e._ctx, px[0], px[1], rng_len, q)
        assert max(abs(q[i] - p[i]) for i in range(3)) < 2e-2
        n_ok += 1
    assert n_ok > 10000


# ---- oracle/_ref: the slice of the REFERENCE that compiles for the host as it lies (oracle/Makefile `ref`, oracle/ref_vhu_host.cpp) ----

def _ref_host():
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_vhu_host.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_vhu_host.so not built (needs /root/reference: `make -C oracle ref`)")
    return C.CDLL(path)


def test_reference_compiled_struct_layouts_and_constants():
    """voxel_hash_utils.cuh / params.h of the reference, compiled by g++ from where they lie: the layouts of Voxel, HashEntry,
    Vertex, Triangle and the constants of the path — against the boundary structs and the numbers this repository uses."""
    ref = _ref_host()
    lay = (C.c_int32 * 16)()
    ref.ref_struct_layout(lay)
    assert list(lay[:5]) == [12, 0, 4, 8, 11]          # Voxel: sdf, sum_squared, rgb[3], weight
    assert list(lay[5:10]) == [24, 0, 12, 16, 20]      # HashEntry: pos, offset, ptr, resolution (T2)
    assert list(lay[10:15]) == [24, 12, 72, 24, 48]    # Vertex {p, c}, Triangle {v0, v1, v2} (T3)
    vd = capi.VOXEL_DTYPE
    assert vd.itemsize == lay[0] and [vd.fields[k][1] for k in ("sdf", "sum_squared", "rgb", "weight")] == list(lay[1:5])
    assert capi.TRI_DTYPE.itemsize * 3 == lay[12] or capi.TRI_DTYPE.itemsize == lay[10]
    v = (C.c_uint8 * 12)()
    ref.ref_default_voxel(v)
    assert bytes(v) == bytes(12)
    e = (C.c_uint8 * 24)()
    ref.ref_default_hash_entry(e)
    ent = np.frombuffer(bytes(e), dtype=HASH_ENTRY)[0]
    assert tuple(ent["pos"]) == (0, 0, 0) and ent["offset"] == 0 and ent["ptr"] == -2 and ent["resolution"] == 0
    k = (C.c_int64 * 16)()
    ref.ref_constants(k)
    assert list(k[:3]) == [73856093, 19349669, 83492791]
    assert list(k[3:15]) == [8, 512, 3, 10, 7, 255, 1024, 16, -1, -2, 0, 8]
    ref.ref_float_constant.restype = C.c_double
    assert ref.ref_float_constant(0) == float(np.float32(1e-6)) and ref.ref_float_constant(1) == float(np.float32(0.15))
    assert ref.ref_float_constant(2) == 10.0


def test_reference_compiled_index_helpers_match_the_oracle(oracle):
    """linearizeVoxelPos / virtualVoxelPosToSDFBlockIndex / delinearizeVoxelPos / SDFBlockToVirtualVoxelPos /
    virtualVoxelPosToWorld of the reference, EXECUTED (host build of voxel_hash_utils.cuh), against the oracle's restatement
    and the independent numpy one: every voxel position of a 61^3 cube around the origin for block sizes 8, 4 and 2."""
    import independent as ind

    ref = _ref_host()
    r = np.arange(-30, 31, dtype=np.int32)
    xyz = np.stack(np.meshgrid(r, r, r, indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    xyz = np.ascontiguousarray(xyz)
    n = len(xyz)
    oracle.orc_kat_voxel_to_block_index.restype = C.c_uint
    for bs in (8, 4, 2):
        out = np.zeros(n, np.uint32)
        ref.ref_voxel_to_block_index(xyz.ctypes.data_as(C.c_void_p), C.c_int64(n), bs, out.ctypes.data_as(C.c_void_p))
        # the reference's formula: local coordinate made non-negative, divided by 8 / bs, linearised with stride 8 (vhu.cuh:110-128)
        loc = (xyz % 8) // (8 // bs)
        assert np.array_equal(out, (loc[:, 2] * 64 + loc[:, 1] * 8 + loc[:, 0]).astype(np.uint32))
        step = 7  # the oracle hook is one ctypes call per position: a regular subsample, plus the survey's known answers
        got = np.array([oracle.orc_kat_voxel_to_block_index(int(x), int(y), int(z), bs) for x, y, z in xyz[::step]], np.uint32)
        assert np.array_equal(got, out[::step])
        lin = np.zeros(n, np.uint32)
        ref.ref_linearize(xyz.ctypes.data_as(C.c_void_p), C.c_int64(n), bs, lin.ctypes.data_as(C.c_void_p))
        assert np.array_equal(lin, (xyz[:, 2] * bs * bs + xyz[:, 1] * bs + xyz[:, 0]).astype(np.uint32))
        idx = np.arange(bs ** 3, dtype=np.uint32)
        de = np.zeros((len(idx), 3), np.uint32)
        ref.ref_delinearize(idx.ctypes.data_as(C.c_void_p), C.c_int64(len(idx)), bs, de.ctypes.data_as(C.c_void_p))
        o = (C.c_int * 3)()
        for i in idx[:: max(1, len(idx) // 64)]:
            oracle.orc_kat_delinearize(int(i), bs, o)
            assert tuple(o) == tuple(int(c) for c in de[i])
        assert np.array_equal(de, np.stack([idx % bs, (idx % (bs * bs)) // bs, idx // (bs * bs)], -1))
    assert ref_known(ref, (-9, 17, 3), 8) == 207 and ref_known(ref, (-9, 17, 3), 4) == 67  # the survey's captured answers
    b2v = np.zeros_like(xyz)
    ref.ref_block_to_voxel(xyz.ctypes.data_as(C.c_void_p), C.c_int64(n), b2v.ctypes.data_as(C.c_void_p))
    assert np.array_equal(b2v, xyz * 8)
    for vs in (0.01, 0.02, 0.2, 1e-6):
        w = np.zeros((n, 3), np.float32)
        ref.ref_voxel_to_world(C.c_float(vs), xyz.ctypes.data_as(C.c_void_p), C.c_int64(n), w.ctypes.data_as(C.c_void_p))
        assert np.array_equal(w.view(np.uint32), ind.voxel_to_world(np.float32(vs), xyz).view(np.uint32))


def ref_known(ref, v, bs):
    xyz = np.array([v], np.int32)
    out = np.zeros(1, np.uint32)
    ref.ref_voxel_to_block_index(xyz.ctypes.data_as(C.c_void_p), C.c_int64(1), bs, out.ctypes.data_as(C.c_void_p))
    return int(out[0])
