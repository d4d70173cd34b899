"""Parity of the gfx950 HIP path against the CPU oracle, through the C ABI.  Runs on the GPU box (-m gpu).

Bars: canonical occupancy list, u8 payload (colour, weight) and the triangle index buffer bit-exact; TSDF
values / sum_squared / vertex positions within 1e-5 (north_star).  Both sides share one arithmetic spec, so
the float comparisons are additionally reported as bit-exact or not."""
import hashlib
import json
import os

import numpy as np
import pytest

import parity_utils as pu
from mrhash_amd import capi, synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _pair(hip, oracle, K, params, nb=65536, **extra):
    return pu.make_engine(hip, K, params, nb, **extra), pu.make_engine(oracle, K, params, nb, **extra)


def test_native_library_is_the_one_that_runs(hip):
    # guards against a silent fallback: the context must come from libmrhash_hip.so on a real device
    assert hip.mrh_version().startswith(b"mrhash_hip")
    e = pu.make_engine(hip, synth.CFG1, synth.CFG1_PARAMS, 4096)
    s = e.stats()
    assert s.num_sdf_blocks == 4096 and s.free_fine == 4096 and s.occupied_fine == 0
    maps = open("/proc/self/maps").read()
    assert "libmrhash_hip.so" in maps and "libamdhip64" in maps
    e.close()


def test_shared_reciprocal_division_is_ieee_exact(hip):
    """The fused kernel divides with one refined reciprocal per denominator (mrh_device.h div_rr); it must be
    bit-identical to the correctly rounded fp32 divide the arithmetic spec prescribes."""
    e = pu.make_engine(hip, synth.CFG1, synth.CFG1_PARAMS, 1024)
    for seed in (1, 2, 3):
        assert e.selftest_division(1 << 28, seed) == 0
    e.close()


def test_cfg1_plane_single_frame(hip, oracle):
    a, b = _pair(hip, oracle, synth.CFG1, synth.CFG1_PARAMS)
    f = synth.cfg1_plane()
    pu.feed(a, f)
    pu.feed(b, f)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] == 100 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    m = pu.compare_meshes(a, b)
    assert m["triangles"] > 4000 and m["pos_bit_exact"]
    V, F, C = a.extract_mesh()
    assert np.allclose(V[:, 2], 1.0, atol=1e-6)  # the plane z = 1 m is recovered


def test_cfg1_sphere_multi_frame_weights_accumulate(hip, oracle):
    a, b = _pair(hip, oracle, synth.CFG1, synth.CFG1_PARAMS)
    for _ in range(4):
        f = synth.cfg1_sphere()
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    r = pu.compare_maps(a, b)
    _, v = a.dump_blocks()
    assert v["weight"].max() == 4
    pu.compare_meshes(a, b)


def test_weight_clamp_and_colour_blend(hip, oracle):
    # integration_weight_sample 100 -> weights saturate at 255 on the third frame (vhu.cuh:179)
    params = dict(synth.CFG1_PARAMS, integration_weight_sample=100)
    a, b = _pair(hip, oracle, synth.CFG1, params)
    rng = np.random.default_rng(5)
    for i in range(4):
        f = synth.cfg1_sphere()
        f.rgb = rng.integers(0, 256, size=f.rgb.shape, dtype=np.uint8)
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    pu.compare_maps(a, b)
    _, v = a.dump_blocks()
    assert v["weight"].max() == 255


def test_gc_every_frame_and_starve(hip, oracle):
    # n_frames_invalidate = 2: GC each frame, starve on frames 2 and 4 (voxel_data_structures.cpp:137-145)
    params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2)
    a, b = _pair(hip, oracle, synth.CFG1, params)
    for i in range(6):
        f = synth.cfg1_sphere(zc=1.5 + 0.01 * (i % 3))
        pu.feed(a, f)
        pu.feed(b, f)
        sa, sb = a.stats(), b.stats()
        assert (sa.occupied_fine, sa.free_fine, sa.last_compact_blocks) == (sb.occupied_fine, sb.free_fine, sb.last_compact_blocks)
    a.sync()
    pu.compare_maps(a, b)
    pu.compare_meshes(a, b)


def test_starve_tie_break_is_canonical(hip, oracle):
    # identity pose + plane: many voxels share exactly one camera depth and one pixel
    params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=1, virtual_voxel_size=0.004, sdf_truncation=0.02)
    a, b = _pair(hip, oracle, synth.CFG1, params)
    for i in range(3):
        f = synth.cfg1_plane(z=2.0)
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    pu.compare_maps(a, b)


def test_moving_camera_rotated_poses(hip, oracle):
    K = synth.Intrinsics(160.0, 160.0, 79.5, 59.5, 120, 160)
    params = dict(synth.REPLICA_PARAMS, virtual_voxel_size=0.02, sdf_truncation=0.08)
    a, b = _pair(hip, oracle, K, params, 65536)
    scene = synth.scannet_room()
    for t, q in synth.walk_poses(6, seed=3):
        f = synth.render(scene, K, t, q, depth_scaling=5000.0)
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 500
    pu.compare_meshes(a, b)


@pytest.mark.parametrize("offset", [(0.0, 0.0, 0.0), (163.0, -162.5, 81.0), (420.0, -310.0, 150.0)], ids=["origin", "shift-limit-edge", "far"])
def test_far_from_origin(hip, oracle, offset):
    """The allocation kernel converts voxel -> block with an arithmetic shift where mrh_create proved it equal to the
    reference's float detour (vhu.cuh:75-103) and with the float form beyond that bound: the same walk, translated
    to both sides of the bound, must give the oracle's map."""
    K = synth.Intrinsics(160.0, 160.0, 79.5, 59.5, 120, 160)
    params = dict(synth.REPLICA_PARAMS, virtual_voxel_size=0.02, sdf_truncation=0.08)
    a, b = _pair(hip, oracle, K, params, 65536)
    scene = synth.scannet_room()
    for t, q in synth.walk_poses(4, seed=5):
        f = synth.render(scene, K, t, q, depth_scaling=5000.0)
        f.t = (f.t + np.asarray(offset, np.float32)).astype(np.float32)
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 300
    pu.compare_meshes(a, b)


@pytest.mark.parametrize("eps", [0.0, 0.013], ids=["exact-merge", "eps-cells"])
def test_mesh_postprocess_device_equals_host_and_oracle(hip, oracle, monkeypatch, eps):
    """MeshExtractor::processTriangles runs on the device (mrh_mesh.h: stable sorts + scans); the host restatement
    (MRH_MESH_HOST=1) and the oracle must give the same V / F / C, element for element, in both merge modes."""
    K = synth.Intrinsics(160.0, 160.0, 79.5, 59.5, 120, 160)
    params = dict(synth.REPLICA_PARAMS, virtual_voxel_size=0.02, sdf_truncation=0.08, vertices_merging_threshold=eps,
                  min_weight_threshold=1)
    dev = pu.make_engine(hip, K, params, 65536)
    orc = pu.make_engine(oracle, K, params, 65536)
    monkeypatch.setenv("MRH_MESH_HOST", "1")
    host = pu.make_engine(hip, K, params, 65536)
    monkeypatch.delenv("MRH_MESH_HOST")
    # V / C as doubles over the link (the round-3 read-back) against fp32 staging widened by the host (the default)
    monkeypatch.setenv("MRH_MESH_F64_LINK", "1")
    f64 = pu.make_engine(hip, K, params, 65536)
    monkeypatch.delenv("MRH_MESH_F64_LINK")
    engines = (dev, host, f64, orc)
    scene = synth.scannet_room()
    for t, q in synth.walk_poses(3, seed=11):
        f = synth.render(scene, K, t, q, depth_scaling=5000.0)
        for e in engines:
            pu.feed(e, f)
    out = []
    for e in engines:
        tris = e.extract_triangles()
        V, F, C = e.extract_mesh()
        out.append((tris, V, F, C))
    assert out[0][0].shape[0] > 5000
    for other in out[1:]:
        assert out[0][0].tobytes() == other[0].tobytes()
        for a, b in zip(out[0][1:], other[1:]):
            assert a.shape == b.shape and a.tobytes() == b.tobytes()
    assert out[0][1].shape[0] < out[0][0].shape[0] * 3  # something merged
    # the same mesh when the triangle soup stays on the device (out_triangles == NULL: what GeoWrapper.extractMesh does);
    # this second extraction goes into the staging buffers of the first (the speculative path)
    for e in engines:
        assert e.extract_triangles(soup=False) == out[0][0].shape[0]
        for a, b in zip(out[0][1:], e.extract_mesh()):
            assert a.tobytes() == b.tobytes()
    # a map that outgrows the buffers of its last extraction: the staged copy reports, grows and runs again
    for yaw in (1.3, 2.6, 3.9):
        f = synth.render(scene, K, np.zeros(3, np.float32), synth.yaw_quat(yaw), depth_scaling=5000.0)
        for e in (dev, f64, orc):
            pu.feed(e, f)
    grown = []
    for e in (dev, f64, orc):
        n = e.extract_triangles(soup=False)
        grown.append((n,) + tuple(e.extract_mesh()))
    assert grown[0][0] > out[0][0].shape[0] * 1.3
    for other in grown[1:]:
        assert grown[0][0] == other[0]
        for a, b in zip(grown[0][1:], other[1:]):
            assert a.shape == b.shape and a.tobytes() == b.tobytes()
    for e in engines:
        e.close()


def test_variance_adaptive_multires(hip, oracle):
    params = dict(synth.CFG1_PARAMS, sdf_var_threshold=0.5, n_frames_invalidate_voxels=3)
    a, b = _pair(hip, oracle, synth.CFG1, params, 16384)
    seq = [synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.52), synth.cfg1_sphere(zc=1.49), synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.5)]
    saw_coarse = False
    for f in seq:
        pu.feed(a, f)
        pu.feed(b, f)
        sa, sb = a.stats(), b.stats()
        assert (sa.occupied_fine, sa.occupied_coarse, sa.free_fine, sa.free_coarse) == (sb.occupied_fine, sb.occupied_coarse, sb.free_fine, sb.free_coarse)
        saw_coarse |= sa.occupied_coarse > 0
    assert saw_coarse, "the scenario must actually coarsen blocks"
    a.sync()
    r = pu.compare_maps(a, b)
    d, _ = a.dump_blocks()
    assert set(np.unique(d["resolution"])) == {0, 1}
    pu.compare_meshes(a, b)


@pytest.mark.parametrize("fused", ["1", "0"], ids=["fused", "general"])
def test_multires_fused_path_long_run(hip, oracle, monkeypatch, fused):
    """Multi-resolution frames take the fused two-launch path (inline variance check, in-place coarsening, coarse
    units integrated and collected by the same kernel) except frame 0, starve frames and the frame after each:
    12 moving-camera frames with GC every frame and starve every 5th cross every transition; occupancy, free-list
    counts, payload and mesh must match the oracle after every frame."""
    monkeypatch.setenv("MRH_MR_FUSED", fused)
    K = synth.Intrinsics(160.0, 160.0, 79.5, 59.5, 120, 160)
    params = dict(synth.REPLICA_PARAMS, virtual_voxel_size=0.02, sdf_truncation=0.08, sdf_var_threshold=0.02,
                  n_frames_invalidate_voxels=5)
    a, b = _pair(hip, oracle, K, params, 32768)
    scene = synth.scannet_room()
    rng = np.random.default_rng(3)
    saw_coarse = False
    for i, (t, q) in enumerate(synth.walk_poses(12, seed=9)):
        f = synth.render(scene, K, t, q, depth_scaling=5000.0, noise_sigma=0.003, rng=rng)
        pu.feed(a, f)
        pu.feed(b, f)
        sa, sb = a.stats(), b.stats()
        assert (sa.occupied_fine, sa.occupied_coarse, sa.free_fine, sa.free_coarse) == (sb.occupied_fine, sb.occupied_coarse, sb.free_fine, sb.free_coarse), f"frame {i}"
        saw_coarse |= sa.occupied_coarse > 0
    assert saw_coarse
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 300
    pu.compare_meshes(a, b)


def test_multires_noisy_stream(hip, oracle):
    # the project page's sigma values (0.001 .. 0.01) on a noisy plane: mixed fine/coarse map + mesh
    params = dict(synth.CFG1_PARAMS, sdf_var_threshold=0.01, n_frames_invalidate_voxels=0)
    a, b = _pair(hip, oracle, synth.CFG1, params, 16384)
    rng = np.random.default_rng(7)
    for i in range(5):
        f = synth.cfg1_plane(z=1.0)
        f.depth = (f.depth + rng.normal(0, 0.002, f.depth.shape)).astype(np.float32)
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    pu.compare_maps(a, b)
    pu.compare_meshes(a, b)


def test_edge_inputs(hip, oracle):
    # empty depth image, depth beyond max_depth, NaN / negative pixels, ragged validity mask
    a, b = _pair(hip, oracle, synth.CFG1, dict(synth.CFG1_PARAMS, max_depth=3.0))
    f = synth.cfg1_plane()
    f.depth[:] = 0
    pu.feed(a, f); pu.feed(b, f)
    a.sync()
    assert a.stats().occupied_fine == 0 and b.stats().occupied_fine == 0
    assert a.extract_triangles().shape[0] == 0
    g = synth.cfg1_sphere(background=5.0)  # background beyond max_depth -> ignored (camera.cu:13-14)
    g.depth[::7, ::5] = np.nan
    g.depth[3::11, 2::9] = -1.0
    g.depth[5::13, :] = 0.0
    pu.feed(a, g); pu.feed(b, g)
    a.sync()
    pu.compare_maps(a, b)
    pu.compare_meshes(a, b)


def test_camera_changes_between_frames(hip, oracle):
    """Two sensors feeding one map: the camera (resolution and intrinsics) is set anew between frames, larger then smaller
    again — the per-pixel buffers follow, the map carries on; single- and multi-resolution, GC on."""
    scene = synth.Scene(synth.Box((-2.0, -1.5, -2.0), (2.0, 1.5, 2.0)), [synth.Box((-0.3, -0.2, 1.0), (0.3, 0.4, 1.4))], seed=3)
    cams = [synth.CFG1, synth.Intrinsics(210.0, 205.0, 101.3, 74.2, 150, 200), synth.Intrinsics(95.0, 97.0, 47.5, 36.0, 72, 96)]
    for var in (0.0, 0.5):
        params = dict(synth.CFG1_PARAMS, integration_weight_sample=2, n_frames_invalidate_voxels=3, sdf_var_threshold=var)
        a, b = _pair(hip, oracle, cams[0], params)
        for i in range(7):
            K = cams[i % 3]
            f = synth.render(scene, K, np.array([0.05 * i, 0.0, -0.5], np.float32), synth.yaw_quat(0.1 * i - 0.3), depth_scaling=5000.0)
            for e in (a, b):
                e.set_camera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, params["min_depth"], params["max_depth"])
                pu.feed(e, f)
            a.sync()
            pu.compare_maps(a, b)
        r = pu.compare_maps(a, b)
        assert r["blocks"] > 100 and r["sdf_bit_exact"]
        pu.compare_meshes(a, b)
        a.close(); b.close()


@pytest.mark.parametrize("var", [0.0, 0.5])
def test_depth_frames_and_point_clouds_interleaved_on_one_map(hip, oracle, var):
    """GeoWrapper::compute fuses a depth image AND a point cloud when both are set (geowrapper.cpp:140-147).  Alternating the two
    kinds of input on one map, with garbage collection on, walks the library through every hand-over between its fast path
    (block summaries, visible lists) and the general kernels the scans use."""
    K = synth.CFG1
    params = dict(synth.CFG1_PARAMS, integration_weight_sample=2, n_frames_invalidate_voxels=3, sdf_var_threshold=var, sdf_truncation=0.12)
    a, b = _pair(hip, oracle, K, params)
    scene = synth.Scene(synth.Box((-2.0, -1.5, -2.0), (2.0, 1.5, 2.0)), [synth.Box((-0.3, -0.2, 1.0), (0.3, 0.4, 1.4))], seed=3)
    for i in range(8):
        t = np.array([0.04 * i, 0.0, -0.5], np.float32)
        q = synth.yaw_quat(0.08 * i - 0.2)
        f = synth.render(scene, K, t, q, depth_scaling=5000.0)
        pts = synth.lidar_scan(scene, t, q, rows=24, cols=96, max_range=10.0)
        for e in (a, b):
            if i % 3 != 2:
                pu.feed(e, f)
            if i % 2 == 1:
                e.set_pose(f.R, f.t)
                e.upload_points(pts)
                e.integrate_points()
        a.sync()
        pu.compare_maps(a, b)
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 100 and r["sdf_bit_exact"] and a.stats().error_flags == 0
    pu.compare_meshes(a, b)
    a.close(); b.close()


def test_error_behaviour(hip):
    e = capi.Engine(hip, capi.Params(num_sdf_blocks=4096, **synth.CFG1_PARAMS))
    with pytest.raises(capi.MrhError) as ei:
        e.integrate()  # no camera yet
    assert ei.value.code == capi.MRH_ERR_STATE
    K = synth.CFG1
    e.set_camera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 30.0)
    with pytest.raises(RuntimeError):
        e.upload_depth(np.zeros((4, 4, 1), np.float32))  # GeoWrapper::setDepthImage ndim check
    with pytest.raises(RuntimeError):
        e.upload_rgb(np.zeros((4, 4), np.uint8))
    e.upload_depth(np.ones((64, 64), np.float32))
    e.upload_rgb(np.zeros((64, 64, 3), np.uint8))
    with pytest.raises(capi.MrhError) as ei:
        e.integrate()  # image shape != camera
    assert ei.value.code == capi.MRH_ERR_INVALID_ARG
    e.close()


def test_pool_exhaustion_is_reported(hip):
    e = pu.make_engine(hip, synth.CFG1, synth.CFG1_PARAMS, 32)  # the plane needs 100 blocks
    pu.feed(e, synth.cfg1_plane())
    with pytest.raises(capi.MrhError) as ei:
        e.sync()
    assert ei.value.code == capi.MRH_ERR_CAPACITY
    s = e.stats()
    assert s.occupied_fine == 32 and s.free_fine == 0
    e.close()


def test_reset_returns_to_empty_and_is_reproducible(hip):
    e = pu.make_engine(hip, synth.CFG1, synth.CFG1_PARAMS, 8192)
    f = synth.cfg1_sphere()
    pu.feed(e, f)
    d1, v1 = e.dump_blocks()
    e.reset()
    s = e.stats()
    assert s.occupied_fine == 0 and s.free_fine == 8192 and s.frames_integrated == 0
    pu.feed(e, f)
    d2, v2 = e.dump_blocks()
    assert np.array_equal(d1, d2) and np.array_equal(v1.view(np.uint8), v2.view(np.uint8))
    e.close()


def test_get_voxel_lookup(hip, oracle):
    a, b = _pair(hip, oracle, synth.CFG1, synth.CFG1_PARAMS)
    f = synth.cfg1_plane()
    pu.feed(a, f); pu.feed(b, f)
    for v in [(0, 0, 50), (-3, 7, 49), (10, -10, 52), (0, 0, 10), (1000, 0, 0)]:
        va, fa = a.get_voxel(*v)
        vb, fb = b.get_voxel(*v)
        assert fa == fb and va.tobytes() == vb.tobytes()


def test_replica_640x480_stream(hip, oracle):
    """BASELINE configs[1] at full resolution, replica.cfg parameters (GC every frame), 4 frames."""
    a, b = _pair(hip, oracle, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
    for f in synth.replica_stream(4):
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 5000 and r["sdf_bit_exact"]
    m = pu.compare_meshes(a, b)  # marching cubes + mesh post-process at full resolution
    assert m["triangles"] > 100000 and m["pos_bit_exact"]


def test_roofline_counters_equal_the_oracles_per_frame(hip, oracle):
    """"U, M come from the oracle's counters for the same frame".  bench.py's roofline numerator is
    24 * U + 24 * M + 7 * H * W with U = voxels the integrate launch updates and M = compact (in-frustum) blocks, both counted on the
    device in profile mode (k_count_updates re-evaluates the update predicate; the list counters).  Here those two numbers are
    compared with the oracle's own counters — integrate_voxel's return values summed (vds.cu:1162-1180 reached) and the length of
    the compacted list (vds.cu:447) — frame by frame on the 640x480 stream, GC on."""
    a, b = _pair(hip, oracle, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
    a.set_profile(True)
    tot_u = tot_m = 0
    for f in synth.replica_stream(6):
        pu.feed(a, f)
        pu.feed(b, f)
        sa, sb = a.stats(), b.stats()
        assert sa.last_updated_voxels == sb.last_updated_voxels, (sa.last_updated_voxels, sb.last_updated_voxels)
        assert sa.last_compact_blocks == sb.last_compact_blocks, (sa.last_compact_blocks, sb.last_compact_blocks)
        tot_u += int(sb.last_updated_voxels)
        tot_m += int(sb.last_compact_blocks)
    sa = a.stats()
    assert int(sa.total_updated_voxels) == tot_u and int(sa.total_compact_blocks) == tot_m
    assert tot_u > 10_000_000 and tot_m > 30_000
    assert sa.n_integrate_kernel == 6 and sa.n_front_kernel == 6 and sa.sum_integrate_kernel_ms > 0 and sa.sum_front_kernel_ms > 0
    a.set_profile(False)
    pu.compare_maps(a, b)  # the counting pass changes nothing


def test_replica_640x480_crosses_the_starve_period_of_the_shipped_configuration(hip, oracle):
    """replica.cfg as shipped: GC every frame, starve every 100th.  103 frames of the 640x480 stream, so that frame 100 — the
    first starve frame of the shipped period, with the table churn and the high-water mark of a hundred frames behind it —
    runs on both sides; maps compared right after it and at the end, then the mesh."""
    a, b = _pair(hip, oracle, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
    for i, f in enumerate(synth.replica_stream(103)):
        pu.feed(a, f)
        pu.feed(b, f)
        if i in (99, 100):
            a.sync()
            pu.compare_maps(a, b)
    a.sync()
    s = a.stats()
    assert s.frames_integrated == 103 and s.error_flags == 0
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 20000 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    m = pu.compare_meshes(a, b)
    assert m["triangles"] > 500000


def test_a_full_reference_table_drops_blocks_the_open_address_table_keeps(hip, oracle):
    """Deviation D9 (oracle header): the reference inserts a block into its bucket of 10 slots or, when that is full, into
    an overflow list of at most 7 entries; with both full the block is silently NOT allocated (allocBlock vds.cu:502-624:
    every failure path just returns).  The open-address table of this library has no such local limit — it keeps every
    block while the table has a free slot within the probe limit.  Shown with a table that is nearly full (the shipped
    sizing, 10 slots per pool block, never gets there): the reference's map is a strict subset of this library's, and on
    the common blocks the payloads agree bit for bit."""
    params = dict(synth.CFG1_PARAMS)
    f = synth.cfg1_plane()
    roomy = pu.make_engine(oracle, synth.CFG1, params, 4096)
    pu.feed(roomy, f)
    d_all, v_all = roomy.dump_blocks()
    assert len(d_all) == 100
    tight = pu.make_engine(oracle, synth.CFG1, params, 4096, hash_slots=120)  # 12 buckets x 10 slots for 100 blocks
    pu.feed(tight, f)
    d_t, v_t = tight.dump_blocks()
    assert 0 < len(d_t) < 100 and np.all(np.isin(d_t, d_all)), "the reference's table did not overflow: the case does not show the deviation"
    a = pu.make_engine(hip, synth.CFG1, params, 4096, hash_slots=120)  # rounded up to the library's minimum of 1024 slots
    pu.feed(a, f)
    a.sync()
    d_a, v_a = a.dump_blocks()
    assert np.array_equal(d_a, d_all) and np.array_equal(v_a.view(np.uint8), v_all.view(np.uint8))
    common = np.isin(d_all, d_t)
    assert np.array_equal(v_a[common].view(np.uint8), v_t.view(np.uint8))
    print(f"reference table of 120 slots kept {len(d_t)} of 100 blocks; open-address table kept 100")


def test_scannet_640x480_furnished_walk(hip, oracle):
    """BASELINE configs[3] stand-in at full resolution: furnished room (depth discontinuities), hand-held-like walk
    with rotated poses, ScanNet intrinsics / depth quantisation, scannet.cfg parameters."""
    a, b = _pair(hip, oracle, synth.SCANNET, synth.SCANNET_PARAMS, 131072)
    for f in synth.scannet_stream(5, start=40):
        pu.feed(a, f)
        pu.feed(b, f)
        sa, sb = a.stats(), b.stats()
        assert (sa.occupied_fine, sa.free_fine, sa.last_compact_blocks) == (sb.occupied_fine, sb.free_fine, sb.last_compact_blocks)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 3000 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]


def test_replica_native_1200x680(hip, oracle):
    """The resolution the reference's Replica numbers are quoted on (replica.cfg:20-21)."""
    a, b = _pair(hip, oracle, synth.REPLICA_NATIVE, synth.REPLICA_PARAMS, 262144)
    for f in synth.replica_stream(2, K=synth.REPLICA_NATIVE):
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 5000 and r["sdf_bit_exact"]


def test_multires_640x480(hip, oracle):
    """BASELINE configs[2] stand-in: variance-adaptive map at full resolution with sensor noise (sigma 2 mm), the
    project page's threshold 0.005; occupancy (incl. which blocks went coarse) and payload must match."""
    params = dict(synth.REPLICA_PARAMS, sdf_var_threshold=0.005)
    a, b = _pair(hip, oracle, synth.REPLICA_640, params, 131072)
    saw_coarse = False
    for f in synth.replica_stream(4, noise_sigma=0.002):
        pu.feed(a, f)
        pu.feed(b, f)
        sa, sb = a.stats(), b.stats()
        assert (sa.occupied_fine, sa.occupied_coarse) == (sb.occupied_fine, sb.occupied_coarse)
        saw_coarse |= sa.occupied_coarse > 0
    a.sync()
    pu.compare_maps(a, b)
    assert saw_coarse
    m = pu.compare_meshes(a, b)  # configs[2] end to end: multi-resolution map -> marching cubes across resolution jumps -> mesh
    assert m["triangles"] > 100000


def test_multires_640x480_long_run_and_mesh(hip, oracle):
    """configs[2] over 24 frames (sigma 0.005, noise-free stream): the fused multi-resolution path, GC every frame and
    two starve frames (period 10), then the mesh over thousands of coarse and fine blocks."""
    params = dict(synth.REPLICA_PARAMS, sdf_var_threshold=0.005, n_frames_invalidate_voxels=10)
    a, b = _pair(hip, oracle, synth.REPLICA_640, params, 131072)
    for f in synth.replica_stream(24):
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    sa, sb = a.stats(), b.stats()
    assert (sa.occupied_fine, sa.occupied_coarse) == (sb.occupied_fine, sb.occupied_coarse) and sb.occupied_coarse > 1000
    pu.compare_maps(a, b)
    m = pu.compare_meshes(a, b)
    assert m["triangles"] > 300000


@pytest.mark.parametrize("var", [0.0, 0.005])
def test_emit_from_corner_records_equals_the_two_pass_emit(hip, monkeypatch, var):
    """The emit pass has two forms: interpolation of the corner records the count pass parked (k_mc_emit_records) and the
    second evaluation (k_mc<emit>: MRH_MC_NO_RECORDS=1, and the fallback when the record buffer is too small).  Same
    soup, byte for byte — single- and multi-resolution map — through: a first buffer that is too small (fallback, the
    buffer grows), the grown buffer (records), and records switched off."""
    params = dict(synth.REPLICA_PARAMS, sdf_var_threshold=var, n_frames_invalidate_voxels=10)
    monkeypatch.setenv("MRH_MC_RECORDS_PER_BLOCK", "1")
    e = pu.make_engine(hip, synth.REPLICA_640, params, 131072)
    for f in synth.replica_stream(12):
        pu.feed(e, f)
    e.sync()
    first = e.extract_triangles()       # one record per block: does not fit -> k_mc<emit>
    second = e.extract_triangles()      # buffer grown to the demand: k_mc_emit_records
    third = e.extract_triangles()
    monkeypatch.setenv("MRH_MC_NO_RECORDS", "1")
    plain = e.extract_triangles()
    assert len(plain) > 100000
    assert first.tobytes() == plain.tobytes() and second.tobytes() == plain.tobytes() and third.tobytes() == plain.tobytes()
    e.close()


def test_known_sample_cells_of_coarse_voxels_equal_the_literal_evaluation(hip, monkeypatch):
    """A coarse voxel whose 3^3 fine cells lie in coarse blocks is evaluated with its sample cells as integers
    (trilinear_coarse_known: lattice points the reference's rounding cannot miss); MRH_MC_NO_COARSE_KNOWN=1 keeps the literal
    float conversions for every coarse voxel.  Same soup, byte for byte, from the record path and from the two-pass emit, on a
    multi-resolution map (far from the origin too: the conversions' error grows with the coordinates)."""
    params = dict(synth.REPLICA_PARAMS, sdf_var_threshold=0.005, n_frames_invalidate_voxels=10)
    for offset in ((0.0, 0.0, 0.0), (137.3, -61.7, 44.1)):
        e = pu.make_engine(hip, synth.REPLICA_640, params, 131072)
        for f in synth.replica_stream(10):
            f.t = (f.t + np.asarray(offset, np.float32)).astype(np.float32)
            pu.feed(e, f)
        e.sync()
        assert e.stats().occupied_coarse > 1000
        known = e.extract_triangles().tobytes()
        monkeypatch.setenv("MRH_MC_NO_RECORDS", "1")
        known_two_pass = e.extract_triangles().tobytes()
        monkeypatch.setenv("MRH_MC_NO_COARSE_KNOWN", "1")
        literal_two_pass = e.extract_triangles().tobytes()
        monkeypatch.delenv("MRH_MC_NO_RECORDS")
        literal = e.extract_triangles().tobytes()
        monkeypatch.delenv("MRH_MC_NO_COARSE_KNOWN")
        assert len(literal) > 72 * 100000
        assert known == literal and known_two_pass == literal and literal_two_pass == literal
        e.close()


def test_block_order_by_counting_equals_the_radix_sort(hip, monkeypatch):
    """The extraction's canonical block order comes from a rank-by-counting pass (k_block_rank) with the offsets from a
    one-workgroup scan; lists beyond 32 k blocks take the byte-wise radix sort of mrh_sort.h over the 64-bit position keys
    (MRH_MC_RADIX_SORT=1 forces it).  Same soup, same V / F / C, byte for byte."""
    e = pu.make_engine(hip, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
    for f in synth.replica_stream(8):
        pu.feed(e, f)
    e.sync()
    soup = e.extract_triangles().tobytes()
    mesh = [a.tobytes() for a in e.extract_mesh()]
    again = [a.tobytes() for a in (e.extract_triangles(), *e.extract_mesh())]  # second extraction: the speculative copy
    assert again[0] == soup and again[1:] == mesh
    monkeypatch.setenv("MRH_MC_RADIX_SORT", "1")
    ref = [a.tobytes() for a in (e.extract_triangles(), *e.extract_mesh())]
    assert len(soup) > 72 * 100000
    assert ref[0] == soup and ref[1:] == mesh
    e.close()


def test_block_order_of_a_list_beyond_the_counting_rank(hip):
    """More than 32 768 blocks in one extraction (the bench's room at 4.5 mm voxels): the canonical order comes from the byte-wise
    radix sort of mrh_sort.h over the 64-bit position keys, the offsets from the same one-workgroup scan.  The list the library
    reports is every live block once, in (x, y, z) order, its counts add up to the soup, and a second extraction repeats it."""
    P = dict(synth.REPLICA_PARAMS, virtual_voxel_size=0.0045, sdf_truncation=0.0315)
    e = pu.make_engine(hip, synth.REPLICA_640, P, 262144)
    for f in synth.replica_stream(4):
        pu.feed(e, f)
    e.sync()
    soup = e.extract_triangles()
    d, cnt = e.triangle_blocks()
    assert len(d) > 32768 and int(cnt.sum()) == len(soup) > 1000000
    order = np.lexsort((d["z"], d["y"], d["x"]))
    assert np.array_equal(order, np.arange(len(d))), "blocks not in (x, y, z) order"
    live, _ = e.dump_blocks()
    assert len(live) == len(d) and np.array_equal(np.sort(live, order=["x", "y", "z"]), d)
    again = e.extract_triangles()
    assert again.tobytes() == soup.tobytes()
    e.close()


def test_stream_out_and_import_match_oracle(hip, oracle):
    """Streamer device half (mrh_stream_out / mrh_import_blocks): the same blocks leave, in position order, with the
    same payload; what stays is the same map; importing them back restores the original; fusion continues identically."""
    import test_streamer as ts

    a, b = ts.build(hip, 4), ts.build(oracle, 4)
    a.sync()
    da, va = a.stream_out((0.3, 0.0, -0.2), 2.2)
    db, vb = b.stream_out((0.3, 0.0, -0.2), 2.2)
    assert len(da) > 100 and da.tobytes() == db.tobytes() and va.tobytes() == vb.tobytes()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 10
    sa, sb = a.stats(), b.stats()
    assert (sa.occupied_fine, sa.free_fine) == (sb.occupied_fine, sb.free_fine)
    a.import_blocks(da, va)
    b.import_blocks(db, vb)
    K = synth.Intrinsics(160.0, 160.0, 79.5, 59.5, 120, 160)
    scene = synth.scannet_room()
    for t, q in synth.walk_poses(6, seed=7)[4:]:
        f = synth.render(scene, K, t, q, depth_scaling=5000.0)
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    pu.compare_maps(a, b)
    pu.compare_meshes(a, b)
    n_all, _ = a.stream_out((0, 0, 0), -1.0)
    assert a.stats().occupied_fine == 0 and len(n_all) == len(b.stream_out((0, 0, 0), -1.0)[0])


def test_gc_and_starve_at_full_resolution(hip, oracle):
    """640x480, GC every frame, starve on frames 2 and 4: the frames where GC is decided inside k_back, the starve
    frames (k_back without GC -> k_starve -> k_summarize_all -> k_free_lists) and the hand-over between them."""
    params = dict(synth.REPLICA_PARAMS, n_frames_invalidate_voxels=2)
    a, b = _pair(hip, oracle, synth.REPLICA_640, params, 131072)
    for f in synth.replica_stream(5):
        pu.feed(a, f)
        pu.feed(b, f)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 5000 and r["sdf_bit_exact"]


@pytest.mark.parametrize("fixture", ["cfg1_golden.json", "lidar_golden.json"])
def test_golden_fixtures(hip, fixture):
    """Committed fixtures (generated by tests/golden/make_golden.py from the oracle): canonical buffers hashed with
    SHA-256 plus summary counts, RGB-D (cfg1) and LiDAR cases."""
    import test_golden_cpu as tg

    g = json.load(open(os.path.join(GOLDEN, fixture)))
    runner = tg.run_rgbd if fixture.startswith("cfg1") else tg.run_lidar
    for name, case in g["cases"].items():
        e = runner(hip, case)
        e.sync()
        tg.check(e, case, name)
        e.close()


def test_full_size_properties(hip):
    """Size-independent properties at the bench workload (640x480, 30 frames, no oracle):
    heap conservation, no duplicate blocks, idempotent dump, weights bounded by frame count,
    and a second identical run reproduces the map bit for bit (the engine is deterministic)."""
    def run():
        e = pu.make_engine(hip, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
        for f in synth.replica_stream(30):
            pu.feed(e, f)
        e.sync()
        return e
    e = run()
    s = e.stats()
    assert s.occupied_fine + s.free_fine == 131072  # test_hash_utils.cu:464-520 conservation
    d, v = e.dump_blocks()
    assert len(d) == s.occupied_fine
    assert len(np.unique(d[["x", "y", "z"]])) == len(d)  # no duplicate block positions
    assert v["weight"].max() <= 30
    d2, v2 = e.dump_blocks()
    assert np.array_equal(d, d2) and np.array_equal(v.view(np.uint8), v2.view(np.uint8))  # dump is read-only
    sdf_abs = np.abs(v["sdf"][v["weight"] > 0])
    assert sdf_abs.max() <= 0.07 + 1e-6  # truncation bound
    e2 = run()
    d3, v3 = e2.dump_blocks()
    assert np.array_equal(d, d3) and np.array_equal(v.view(np.uint8), v3.view(np.uint8))
    e.close(); e2.close()


def test_default_capacity_rule_sizes_the_pool_to_hbm(hip):
    """num_sdf_blocks = 0 applies the reference's sizing rule (geowrapper.cpp:37-54) to the free HBM of the device:
    tens of millions of blocks on a 288 GB part.  The map built in such a pool equals the one built in a small pool
    (block indices and byte offsets are 64-bit clean)."""
    K = synth.CFG1
    big = capi.Engine(hip, capi.Params(num_sdf_blocks=0, **synth.CFG1_PARAMS))
    small = pu.make_engine(hip, K, synth.CFG1_PARAMS, 16384)
    big.set_camera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 30.0)
    st = big.stats()
    assert st.num_sdf_blocks > 4_000_000 and st.free_fine == st.num_sdf_blocks
    for f in (synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.51)):
        pu.feed(big, f)
        pu.feed(small, f)
    big.sync()
    pu.compare_maps(big, small)
    pu.compare_meshes(big, small)
    big.close()
    small.close()


def test_reference_compiled_tables_equal_the_generated_header():
    """oracle/_ref/ref_params_dump = the reference's params.h compiled for gfx950 as it lies (its marching-cubes tables are
    `__device__ const` arrays): the values the reference's own source defines, read back through a kernel, against
    include/mrh_mc_tables.h (generated by tools/gen_mc_tables.py) and the constants this repository uses."""
    import json
    import os
    import re
    import subprocess

    exe = os.path.join(pu.ROOT, "oracle", "_ref", "ref_params_dump")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_params_dump not built (needs /root/reference at build time: `make -C oracle ref`)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["p"] == [73856093, 19349669, 83492791] and d["sdf_block_size"] == 8 and d["hash_bucket_size"] == 10 and d["linked_list_size"] == 7
    assert d["integration_weight_max"] == 255 and d["max_dda_iteration_count"] == 1024 and d["n_threads"] == 16
    assert np.float32(d["float_epsilon"]) == np.float32(1e-6) and np.float32(d["stream_threshold"]) == np.float32(0.15)  # printed with 9 digits: exact for binary32
    assert d["vert_offset"] == [c for k in range(8) for c in (7 * ((k >> 2) & 1), 7 * ((k >> 1) & 1), 7 * (k & 1))]
    cls, data, vert = d["regularCellClass"], d["regularCellData"], d["regularVertexData"]
    assert len(cls) == 256 and len(data) == 256 and len(vert) == 256 * 12
    txt = open(os.path.join(pu.ROOT, "include", "mrh_mc_tables.h")).read()
    rows = re.findall(r"\{((?:0x[0-9A-Fa-f]{2},?){16})\}", txt)
    mine = [[int(x, 16) for x in row.split(",") if x] for row in rows[:256]]
    assert len(mine) == 256
    for cube in range(256):
        geo = data[cls[cube] * 16]
        ntri = geo & 0x0F  # RegularCellData::getTriangleCount (params.h:110-112)
        want = [ntri] + [vert[cube * 12 + data[cls[cube] * 16 + 1 + s]] & 0xFF for s in range(3 * ntri)]
        assert mine[cube][: 1 + 3 * ntri] == want, cube


@pytest.mark.parametrize("var,pixel_offset", [(0.0, 0.0), (0.0, 0.5), (0.005, 0.0)])
def test_mesh_accuracy_against_the_analytic_room(hip, var, pixel_offset):
    """A check from OUTSIDE the oracle, at BASELINE's full size: 60 frames of the 640x480 Replica stand-in (depth quantised to
    1/6553.5 m like the dataset PNGs) fused at 1 cm voxels, single- and multi-resolution; the vertices of the extracted mesh
    against the analytic room (an axis-aligned box seen from inside) — the accuracy figure of the reference's project page,
    with ground truth that is known exactly here.
    The reference back-projects pixel c along c - cx - 0.5 (camera.cuh:88) but projects by rounding fx x / z + cx
    (camera.cuh:137-138): the two disagree by half a pixel.  Rendered for the projection the integration uses
    (pixel_offset 0) the mesh sits on the walls to 0.16 mm on average (1 cm voxels); rendered for the back-projection (0.5, what
    every other test and the bench feed) the whole mesh is displaced by that half pixel — ~3 mm at 3 m, the reference's own
    systematic error, reproduced."""
    params = dict(synth.REPLICA_PARAMS, sdf_var_threshold=var)
    e = pu.make_engine(hip, synth.REPLICA_640, params, 131072)
    scene = synth.replica_room()
    for t, q in synth.orbit_poses(60):
        pu.feed(e, synth.render(scene, synth.REPLICA_640, t, q, depth_scaling=6553.5, pixel_offset=pixel_offset))
    e.extract_triangles(soup=False)
    V, F, C = e.extract_mesh()
    lo, hi = np.array(scene.room.lo), np.array(scene.room.hi)
    assert len(V) > 300000 and len(F) > 500000
    dist = np.minimum(np.abs(V - lo), np.abs(V - hi)).min(axis=1)  # distance to the nearest wall plane
    vs = params["virtual_voxel_size"]
    stats = (float(dist.mean()), float(np.quantile(dist, 0.999)), float(dist.max()))
    if pixel_offset == 0.0 and var == 0.0:
        assert stats[0] < 0.03 * vs and stats[1] < 0.5 * vs and stats[2] < 0.75 * vs, stats  # measured: 0.16 mm mean, 5.9 mm at the worst room edge
    elif pixel_offset == 0.0:  # coarse blocks: 2 cm voxels
        assert stats[0] < 0.3 * vs and stats[1] < 1.5 * vs and stats[2] < 2.5 * vs, stats  # measured: 1.9 mm mean, 17 mm max
        assert e.stats().occupied_coarse > 1000
    else:
        assert 0.1 * vs < stats[0] < 0.5 * vs and stats[2] < 1.0 * vs, stats
    e.close()


@pytest.mark.parametrize("kind", ["rgbd_multires", "lidar"])
def test_far_end_of_an_hbm_sized_pool(hip, monkeypatch, kind):
    """BASELINE configs[4] names "288 GB HBM hash-table sizing": with the pool sized to the device (tens of millions of blocks,
    > 100 GB of voxel planes) and the free list handing out its HIGHEST indices first (MRH_DEBUG_HEAP_DESCENDING), a small
    scene lives at byte offsets far beyond 2^32 in every plane, descriptor and summary array.  The map, the mesh and the
    scan results must equal those of a small pool filled from index 0."""
    if kind == "rgbd_multires":
        K = synth.CFG1
        params = dict(synth.CFG1_PARAMS, integration_weight_sample=2, sdf_var_threshold=0.5, n_frames_invalidate_voxels=3)
        small = pu.make_engine(hip, K, params, 16384)
        monkeypatch.setenv("MRH_DEBUG_HEAP_DESCENDING", "1")
        big = capi.Engine(hip, capi.Params(num_sdf_blocks=0, **params))
        monkeypatch.delenv("MRH_DEBUG_HEAP_DESCENDING")
        big.set_camera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, params["min_depth"], params["max_depth"])
        frames = [synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.51), synth.cfg1_sphere(zc=1.49), synth.cfg1_sphere(zc=1.5), synth.cfg1_sphere(zc=1.52)]
        for f in frames:
            pu.feed(big, f)
            pu.feed(small, f)
    else:
        params = dict(synth.VBR_PARAMS, n_frames_invalidate_voxels=2)
        small = pu.make_lidar_engine(hip, params, 100.0)
        monkeypatch.setenv("MRH_DEBUG_HEAP_DESCENDING", "1")
        big = capi.Engine(hip, capi.Params(num_sdf_blocks=0, **params))
        monkeypatch.delenv("MRH_DEBUG_HEAP_DESCENDING")
        big.set_camera(1.0, 1.0, 0.0, 0.0, 1, 1, params["min_depth"], 100.0, model=1)
        scene = synth.street_canyon()
        for t, q in synth.drive_poses(3, step=1.0):
            pts = synth.lidar_scan(scene, t, q, rows=32, cols=512)
            for e in (big, small):
                e.set_pose(synth.quat_to_rot(q), t)
                e.upload_points(pts)
                e.integrate_points()
    big.sync()
    st = big.stats()
    assert st.num_sdf_blocks > 8_000_000 and st.error_flags == 0
    r = pu.compare_maps(big, small)
    assert r["blocks"] > 50 and r["sdf_bit_exact"]
    if kind == "rgbd_multires":
        assert st.occupied_coarse > 0
        pu.compare_meshes(big, small)
    big.close()
    small.close()


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_randomised_parameters_and_shapes(hip, oracle, seed):
    """Parameter / shape fuzz: odd image sizes (partial allocation tiles, clamped footprints), off-centre anisotropic
    intrinsics, range-dependent truncation (sdf_truncation_scale > 0), weight sample / weight max other than 1 / 255,
    depth limits that cut the scene, GC with a starve period — fast path (single resolution) and fused / general
    multi-resolution path, always against the oracle."""
    rng = np.random.default_rng(100 + seed)
    rows, cols = int(rng.integers(90, 200)), int(rng.integers(100, 260))
    f = float(rng.uniform(0.7, 1.3)) * cols
    K = synth.Intrinsics(f, f * float(rng.uniform(0.85, 1.15)), cols * float(rng.uniform(0.35, 0.65)), rows * float(rng.uniform(0.35, 0.65)), rows, cols)
    vs = float(rng.choice([0.015, 0.02, 0.03]))
    params = dict(
        sdf_truncation=float(rng.uniform(3.0, 5.0)) * vs, sdf_truncation_scale=float(rng.choice([0.0, 0.01, 0.03])),
        integration_weight_sample=int(rng.integers(1, 6)), integration_weight_max=int(rng.integers(12, 256)),
        virtual_voxel_size=vs, n_frames_invalidate_voxels=int(rng.choice([0, 1000, 3])), voxel_extents_scale=1,
        marching_cubes_threshold=1.5, min_weight_threshold=int(rng.integers(1, 4)),
        sdf_var_threshold=float(rng.choice([0.0, 0.0, 0.02])), vertices_merging_threshold=0.0,
        min_depth=float(rng.choice([0.01, 0.8])), max_depth=float(rng.choice([30.0, 3.2])),
    )
    a, b = _pair(hip, oracle, K, params, 65536)
    scene = synth.scannet_room()
    noise = np.random.default_rng(seed)
    for t, q in synth.walk_poses(7, seed=20 + seed):
        fr = synth.render(scene, K, t, q, depth_scaling=5000.0, noise_sigma=0.002 if params["sdf_var_threshold"] > 0 else 0.0, rng=noise)
        pu.feed(a, fr)
        pu.feed(b, fr)
    a.sync()
    sa, sb = a.stats(), b.stats()
    assert (sa.occupied_fine, sa.occupied_coarse, sa.free_fine, sa.free_coarse) == (sb.occupied_fine, sb.occupied_coarse, sb.free_fine, sb.free_coarse), params
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 50, params
    # 3DGS splat seeds of the last frame on the same map (random quad-tree parameters)
    thr, min_px = float(rng.choice([0.0, 1e-4, 2e-3, 0.05])), int(rng.integers(0, 4))
    sa_, sb_ = a.splat_seeds(thr, min_px), b.splat_seeds(thr, min_px)
    assert np.array_equal(a.qtree_leaves(), b.qtree_leaves()), (params, thr, min_px)
    assert sa_.tobytes() == sb_.tobytes(), (params, thr, min_px)
    pu.compare_meshes(a, b)
    a.close()
    b.close()


def test_safe_division_fallback_gives_the_same_map(hip, oracle, monkeypatch):
    """The running mean divides with one residual step after checks at mrh_create (vs / 2 exhaustively over the working
    range, the weight sums against correctly rounded reciprocals); MRH_SAFE_DIV=1 forces the instantiation a failed check
    would select (two steps).  Both must reproduce the oracle's IEEE divisions."""
    K = synth.REPLICA_640
    monkeypatch.setenv("MRH_SAFE_DIV", "1")
    safe = pu.make_engine(hip, K, synth.REPLICA_PARAMS, 65536)
    monkeypatch.delenv("MRH_SAFE_DIV")
    fast = pu.make_engine(hip, K, synth.REPLICA_PARAMS, 65536)
    orc = pu.make_engine(oracle, K, synth.REPLICA_PARAMS, 65536)
    for f in synth.replica_stream(4, noise_sigma=0.002):
        for e in (safe, fast, orc):
            pu.feed(e, f)
    pu.compare_maps(safe, orc)
    pu.compare_maps(fast, orc)
    for e in (safe, fast, orc):
        e.close()


def test_plain_c_program_drives_the_abi(hip, tmp_path):
    """examples/c_abi_smoke.c: a C11 program with nothing but include/mrhash_hip.h and -lmrhash_hip reproduces the
    counts of the golden plane case."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_smoke")
    libdir = os.path.join(root, "mrhash_amd", "csrc")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_smoke.c"), "-o", exe,
                    "-L" + libdir, "-lmrhash_hip", "-Wl,-rpath," + libdir, "-lm"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    got = dict(zip(out[-12::2], map(int, out[-11::2])))
    case = json.load(open(os.path.join(GOLDEN, "cfg1_golden.json")))["cases"]["plane_1frame"]
    assert (got["blocks"], got["weighted_voxels"], got["triangles"], got["faces"]) == (case["blocks"], case["weighted_voxels"], case["triangles"], case["faces"])
    assert got["free_fine"] == 16384 - case["blocks"]


@pytest.mark.parametrize("defer", ["1", "0"])
def test_upload_ring_semantics(hip, oracle, monkeypatch, defer):
    """Host uploads go through three-slot rings on a copy stream: an image uploaded once stays current over several
    frames, an upload that is overwritten before any frame used it is harmless, caller-owned device pointers and uploads
    mix, and a burst of frames without any synchronisation reuses slots only after the frame that read them.
    defer = 1 (the default): a frame of host images is launched one mrh_integrate late, with the pose and the images it was
    issued under (flush_deferred) — the ring that wraps before any frame, the frame that keeps an old colour image and the
    pose that changes in between are exactly what that has to survive; defer = 0 launches at once."""
    monkeypatch.setenv("MRH_DEFER_UPLOADS", defer)
    K = synth.CFG1
    a = pu.make_engine(hip, K, synth.CFG1_PARAMS, 16384)
    b = pu.make_engine(oracle, K, synth.CFG1_PARAMS, 16384)
    f0, f1, f2 = synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.52), synth.cfg1_plane(z=1.1)
    rng = np.random.default_rng(3)
    junk = rng.integers(0, 256, f0.rgb.shape, dtype=np.uint8)
    for e in (a, b):
        e.set_pose(f0.R, f0.t)
        e.upload_depth(f0.depth); e.upload_rgb(f0.rgb); assert not e.integrate()
        e.upload_depth(f1.depth); assert not e.integrate()          # colour image of f0 stays current
        e.upload_rgb(junk); e.upload_rgb(f2.rgb); e.upload_rgb(junk); e.upload_rgb(f2.rgb)  # ring wraps before any frame
        e.upload_depth(f2.depth); assert not e.integrate()
        assert not e.integrate()                                     # the same images again
    pu.compare_maps(a, b)
    # burst: 24 frames back to back, alternating images, no sync in between
    seq = [f0, f1, f2] * 8
    for f in seq:
        pu.feed(a, f)
    for f in seq:
        pu.feed(b, f)
    pu.compare_maps(a, b)
    # pool level without a stall: after a sync the newest report is the exact level
    a.peek_free_blocks()
    pu.feed(a, f0); pu.feed(b, f0)
    a.sync()
    fine, coarse, behind = a.peek_free_blocks()
    assert behind == 0 and (fine, coarse) == a.free_blocks() == b.free_blocks()
    pu.compare_maps(a, b)
    a.close()
    b.close()


def test_two_contexts_on_two_host_threads_share_the_copy_pool(hip, oracle):
    """The helper threads that share the setters' staging copies are one pool per process: two contexts driven from two
    host threads upload 640x480 frames concurrently (ctypes releases the GIL inside the calls); both maps must equal the
    oracle's."""
    import threading

    streams = [list(synth.replica_stream(6)), list(synth.scannet_stream(6, start=10))]
    Ks = [synth.REPLICA_640, synth.SCANNET]
    engines = [pu.make_engine(hip, Ks[i], synth.REPLICA_PARAMS, 65536) for i in range(2)]
    errors = []

    def drive(i):
        try:
            for rep in range(3):  # the same frames three times: more uploads in flight, same final weights x3
                for f in streams[i]:
                    pu.feed(engines[i], f)
            engines[i].sync()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for i in range(2):
        b = pu.make_engine(oracle, Ks[i], synth.REPLICA_PARAMS, 65536)
        for rep in range(3):
            for f in streams[i]:
                pu.feed(b, f)
        r = pu.compare_maps(engines[i], b)
        assert r["blocks"] > 3000 and r["sdf_bit_exact"]
        b.close()
        engines[i].close()


def test_marching_cubes_prescreen_changes_nothing(hip, monkeypatch):
    """The count pass skips voxels whose 27 surrounding cells are all clearly positive or all clearly negative; with
    the prescreen off (MRH_MC_NO_PRESCREEN=1) every voxel takes the full path: same triangle buffer, byte for byte —
    also with a range-dependent truncation and with noise that puts samples near zero."""
    K = synth.REPLICA_640
    params = dict(synth.REPLICA_PARAMS, sdf_truncation_scale=0.01, min_weight_threshold=1)
    e = pu.make_engine(hip, K, params, 65536)
    for f in synth.replica_stream(3, noise_sigma=0.004):
        pu.feed(e, f)
    on = e.extract_triangles()
    monkeypatch.setenv("MRH_MC_NO_PRESCREEN", "1")
    off = e.extract_triangles()
    assert on.shape[0] > 20000 and on.tobytes() == off.tobytes()
    e.close()


def test_contexts_release_their_device_memory(hip):
    """Create / use / destroy in a loop (uploads, frames, seeds, extraction — every lazily allocated buffer gets
    allocated): the free device memory afterwards is what it was before."""
    from mrhash_amd import hipmem

    K = synth.CFG1
    f = synth.cfg1_sphere()

    def cycle():
        e = pu.make_engine(hip, K, dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2), 32768)
        for _ in range(3):
            pu.feed(e, f)
        e.splat_seeds(0.0025, 1)
        e.peek_free_blocks()
        pu.feed(e, f)
        e.extract_triangles()
        e.extract_mesh()
        e.stream_out((0.0, 0.0, 0.0), -1.0)
        e.close()

    for _ in range(3):  # the first uses pay one-off runtime allocations (code objects, queues, scratch)
        cycle()
    hipmem.synchronize()
    free0, _ = hipmem.mem_get_info()
    for _ in range(25):
        cycle()
    hipmem.synchronize()
    free1, _ = hipmem.mem_get_info()
    assert free0 - free1 < 32 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 25 create/destroy cycles"


_PIPE40 = {}


def _pipe40_reference(hip, oracle, monkeypatch):
    """The 40 frames of the 640x480 orbit through the oracle and through the library fusing serially (MRH_PIPE=0), once for the
    three reclaim periods below (the oracle's 40 frames are most of a test's time): the two engines, kept open, and their pool
    levels after frame 17."""
    if not _PIPE40:
        monkeypatch.setenv("MRH_PIPE", "0")
        s = pu.make_engine(hip, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
        monkeypatch.delenv("MRH_PIPE")
        b = pu.make_engine(oracle, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
        frames = list(synth.replica_stream(40))
        for i, f in enumerate(frames):
            pu.feed(s, f)
            pu.feed(b, f)
            if i == 17:
                _PIPE40["level17"] = (s.free_blocks(), b.free_blocks())
        _PIPE40.update(s=s, b=b, frames=frames, ss=s.stats(), sb=b.stats())  # statistics before anything else compacts the block list
    return _PIPE40


@pytest.mark.parametrize("period", ["2", "5", "1000"])
def test_pipelined_frames_build_the_serial_map(hip, oracle, monkeypatch, period):
    """Pipelined frames (mrh_capi.hip integrate_lazy: front half on a second stream next to the integrations before it, lazy
    garbage collection through zombies and per-frame want stamps, k_reclaim every `period` frames) against the oracle AND against
    the same library fusing serially (MRH_PIPE=0): 40 frames of the 640x480 orbit without a synchronisation in between, GC on
    every frame — the blocks of the truncation band's near edge are collected and wanted again every single frame, blocks that
    leave the band stay zombies until the reclaim.  Occupancy, payload and mesh must be the serial ones, bit for bit; the
    statistics too (blocks freed / inserted are counted where the reference frees / inserts)."""
    ref = _pipe40_reference(hip, oracle, monkeypatch)
    s, b = ref["s"], ref["b"]
    monkeypatch.setenv("MRH_PIPE_PERIOD", period)
    a = pu.make_engine(hip, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
    for i, f in enumerate(ref["frames"]):
        pu.feed(a, f)
        if i == 17:  # an entry point in the middle of the pipeline: the pending integration and the reclaim come first
            assert a.free_blocks() == ref["level17"][0] == ref["level17"][1]
    sa, ss, sb = a.stats(), ref["ss"], ref["sb"]
    for k in ("occupied_fine", "free_fine", "frames_integrated", "last_compact_blocks", "error_flags"):
        assert getattr(sa, k) == getattr(ss, k) == getattr(sb, k), (k, getattr(sa, k), getattr(ss, k), getattr(sb, k))
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 15000 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    pu.compare_maps(a, s)
    m = pu.compare_meshes(a, b)
    assert m["triangles"] > 400000 and m["pos_bit_exact"]
    a.close()


@pytest.mark.parametrize("mode", ["pipelined", "serial-fused", "pipelined-old-starve", "synced-every-frame"])
def test_starve_frames_stay_in_the_pipeline(hip, oracle, monkeypatch, mode):
    """Round 6: a starve frame (voxel_data_structures.cpp:139) no longer flushes the pipeline — behind its integration (which
    collects nothing) run the two min-passes of the z-buffer and one tail launch (the winner's weight, the block summaries, the
    garbage collection through zombies, the other z-buffer pair cleared): k_starve_z / k_starve_tail, mrh_fast2.h.  Period 3 on
    the moving-camera walk at 160x120: every third frame starves while the front half of the next frame runs next to it, blocks
    are collected by starve frames and wanted again, zombies meet the z-buffer passes (they must not take part).  Against the
    oracle, bit for bit, in four ways of running the same frames: pipelined, serial with the fused launches (MRH_PIPE=0), pipelined
    with the eight launches of rounds 1-5 (MRH_STARVE_FUSED=0: those frames flush the pipeline), and with a synchronisation after
    every frame (the library falls back to serial frames after three)."""
    if mode == "serial-fused":
        monkeypatch.setenv("MRH_PIPE", "0")
    if mode == "pipelined-old-starve":
        monkeypatch.setenv("MRH_STARVE_FUSED", "0")
    K = synth.Intrinsics(160.0, 160.0, 79.5, 59.5, 120, 160)
    params = dict(synth.REPLICA_PARAMS, virtual_voxel_size=0.02, sdf_truncation=0.08, n_frames_invalidate_voxels=3)
    a, b = _pair(hip, oracle, K, params, 65536)
    scene = synth.scannet_room()
    for i, (t, q) in enumerate(synth.walk_poses(20, seed=11)):
        f = synth.render(scene, K, t, q, depth_scaling=5000.0)
        pu.feed(a, f)
        pu.feed(b, f)
        if mode == "synced-every-frame":
            a.sync()
    a.sync()
    sa, sb = a.stats(), b.stats()
    for k in ("occupied_fine", "free_fine", "frames_integrated", "error_flags"):
        assert getattr(sa, k) == getattr(sb, k), (k, getattr(sa, k), getattr(sb, k))
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 500 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    pu.compare_meshes(a, b)
    a.close()
    b.close()


def test_peeks_next_to_pipelined_frames_read_reports_that_were_written(hip, oracle):
    """the pool report of a pipelined frame is written behind its integration (launch_pending), and only
    then does the mark exist for the non-blocking peeks — an event on the front stream, or a report launched before the
    integration, would let a peek read a slot nobody has written yet (zeros: "1 free block, no flags") or the report of eight
    marks earlier while claiming to be 0-1 frames behind."""
    pool = 16384
    params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=1000)
    a, b = _pair(hip, oracle, synth.CFG1, params, pool)
    levels = {}
    a.peek_free_blocks()  # switches the reports on
    for i in range(40):
        f = synth.cfg1_sphere(zc=1.3 + 0.01 * i)
        pu.feed(a, f)
        pu.feed(b, f)
        levels[i] = b.free_blocks()[0]
        fine, coarse, behind = a.peek_free_blocks()
        assert a.peek_error_flags() == 0
        assert 0 <= behind <= 8
        # an honest report: the level of a recent frame (blocks emptied by a pipelined frame keep their pool slot until the
        # reclaim, so the device's free list may be some hundred blocks below the reference's) — not the zeros of an unwritten slot
        window = [levels[j] for j in range(max(0, i - behind - 3), i + 1)]
        assert min(window) - 1024 <= fine <= max(window), (i, fine, behind, window)
    a.sync()
    fine, coarse, behind = a.peek_free_blocks()
    assert behind == 0 and (fine, coarse) == a.free_blocks() == b.free_blocks()
    pu.compare_maps(a, b)
    a.close()
    b.close()


@pytest.mark.parametrize("zlist_cap", [0, 8])
def test_pipelined_frames_with_a_moving_scene_and_a_small_pool(hip, oracle, monkeypatch, zlist_cap):
    """The same on the 128x128 case with things that stress the zombie rules: a sphere that jumps back and forth (blocks are
    collected, stay unwanted for some frames, and are wanted again before or after the reclaim), a pool so small that the
    context leaves the pipelined mode when room gets short (free < pool / 4), and interleaved entry points.
    zlist_cap = 8: the list of emptied blocks overflows every few frames (the append is bounded and the reclaim
    then finds the zombies by their flag)."""
    monkeypatch.setenv("MRH_PIPE_PERIOD", "4" if not zlist_cap else "16")
    if zlist_cap:
        monkeypatch.setenv("MRH_ZLIST_CAP", str(zlist_cap))
    params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=1000)  # GC every frame, no starve
    a, b = _pair(hip, oracle, synth.CFG1, params, 2048)
    rng = np.random.default_rng(11)
    for i in range(60):
        zc = 1.3 + 0.4 * float(rng.random()) if i % 3 else 1.5
        f = synth.cfg1_sphere(zc=zc, radius=0.3 + 0.2 * float(rng.random()))
        pu.feed(a, f)
        pu.feed(b, f)
        if i in (7, 31):
            pu.compare_maps(a, b)
        if i == 44:
            assert a.peek_free_blocks()[:2] is not None
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 50 and r["sdf_bit_exact"]
    pu.compare_meshes(a, b)
    assert a.stats().error_flags == 0
