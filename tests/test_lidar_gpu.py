"""LiDAR scans on the GPU (mrh_lidar.h) against the oracle: bit-exact occupancy, payload and mesh."""
import numpy as np
import pytest

import parity_utils as pu
from mrhash_amd import capi, synth

pytestmark = pytest.mark.gpu
K1 = synth.Intrinsics(1.0, 1.0, 0.0, 0.0, 1, 1)


@pytest.fixture(scope="module")
def hip():
    return capi.load_hip()


@pytest.fixture(scope="module")
def oracle():
    return pu.oracle_lib()


def _pair(hip, oracle, params=None, blocks=131072, max_depth=100.0):
    p = dict(synth.VBR_PARAMS, **(params or {}))
    out = []
    for lib in (hip, oracle):
        e = capi.Engine(lib, capi.Params(num_sdf_blocks=blocks, **p))
        e.set_camera(K1.fx, K1.fy, K1.cx, K1.cy, K1.rows, K1.cols, p["min_depth"], max_depth, model=1)
        out.append(e)
    return out


def _feed(engines, pts, t, q):
    for e in engines:
        e.set_pose(synth.quat_to_rot(q), t)
        e.upload_points(pts)
        e.integrate_points()


def test_single_scan_matches_oracle(hip, oracle):
    a, b = _pair(hip, oracle)
    scene = synth.street_canyon()
    (t, q), = synth.drive_poses(1)
    _feed((a, b), synth.lidar_scan(scene, t, q, rows=32, cols=512), t, q)
    a.sync()
    sa, sb = a.stats(), b.stats()
    assert (sa.occupied_fine, sa.free_fine) == (sb.occupied_fine, sb.free_fine)
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 1000 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]


@pytest.mark.parametrize("var_threshold", [0.0, 0.05])
def test_scan_layout_is_a_hint_for_the_beam_order_only(hip, oracle, var_threshold):
    """mrh_set_scan_layout / the look at host clouds (mrh_detect_scan_layout) decide which 256 beams a walk workgroup takes — 16 x 16
    patches of an organised scan instead of 256 in a row.  Whatever is said, true or false, the map is the oracle's: a voxel's
    records are put into point order before they are folded."""
    from mrhash_amd import hipmem

    scene = synth.street_canyon()
    poses = synth.drive_poses(3, step=1.5)
    scans = [synth.lidar_scan(scene, t, q, rows=32, cols=512) for t, q in poses]
    assert hip.mrh_detect_scan_layout(np.ascontiguousarray(scans[0]).ctypes.data, len(scans[0])) == 512
    p = dict(synth.VBR_PARAMS, sdf_var_threshold=var_threshold)
    ref = capi.Engine(oracle, capi.Params(num_sdf_blocks=131072, **p))
    ref.set_camera(K1.fx, K1.fy, K1.cx, K1.cy, K1.rows, K1.cols, p["min_depth"], 100.0, model=1)
    for (t, q), pts in zip(poses, scans):
        _feed((ref,), pts, t, q)
    for hint, on_device in ((0, False), (-1, False), (512, True), (0, True), (256, True), (2048, False), (16, True), (4096, True)):
        e = capi.Engine(hip, capi.Params(num_sdf_blocks=131072, **p))
        e.set_camera(K1.fx, K1.fy, K1.cx, K1.cy, K1.rows, K1.cols, p["min_depth"], 100.0, model=1)
        e.set_scan_layout(hint)
        for (t, q), pts in zip(poses, scans):
            e.set_pose(synth.quat_to_rot(q), t)
            if on_device:
                d_pts = hipmem.DeviceBuffer.from_numpy(np.ascontiguousarray(pts, dtype=np.float32))
                e.set_points_device(d_pts.ptr, len(pts))
                e.integrate_points()
                e.sync()
            else:
                e.upload_points(pts)
                e.integrate_points()
        e.sync()
        r = pu.compare_maps(e, ref)
        assert r["blocks"] > 1000 and r["sdf_bit_exact"] and r["sumsq_bit_exact"], (hint, on_device)
        e.close()
    ref.close()


@pytest.mark.parametrize("var_threshold", [0.0, 0.05])
def test_wide_records_build_the_same_map(hip, oracle, monkeypatch, var_threshold):
    """The placed records are 8 bytes when the order tag leaves room for the voxel's 9 bits (every test above), 16 otherwise
    (scans beyond 8 M points; 256 k on variance-adaptive maps): MRH_SCAN_WIDE_RECORDS=1 sends a small scan through the wide form."""
    monkeypatch.setenv("MRH_SCAN_WIDE_RECORDS", "1")
    a, b = _pair(hip, oracle, dict(sdf_var_threshold=var_threshold))
    scene = synth.street_canyon()
    for t, q in synth.drive_poses(3, step=1.5):
        _feed((a, b), synth.lidar_scan(scene, t, q, rows=32, cols=512), t, q)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 1000 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]


def test_scans_across_the_wrap_of_the_block_stamps(hip, oracle, monkeypatch):
    """a block's stamp is `2 * sequence + coarse` in 32 bits, so the sequence restarts at 2^31 — with clean stamps
    and both counter sets at zero.  Six scans whose sequence numbers straddle the restart build the oracle's map."""
    monkeypatch.setenv("MRH_SCAN_SEQ_START", str(0x7FFFFFFC))
    a, b = _pair(hip, oracle, dict(min_weight_threshold=1))
    scene = synth.street_canyon()
    for t, q in synth.drive_poses(6, step=1.0):
        _feed((a, b), synth.lidar_scan(scene, t, q, rows=32, cols=512), t, q)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 1000 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    a.close()
    b.close()


def test_drive_with_noise_dropouts_and_mesh(hip, oracle):
    """8 scans along the street, 2 cm range noise, 5 % missing returns, integration distance shorter than the street so
    that the clipping branches (range > max distance, clipped far end) are exercised; then the mesh."""
    a, b = _pair(hip, oracle, dict(min_weight_threshold=1), max_depth=40.0)
    scene = synth.street_canyon()
    rng = np.random.default_rng(2)
    for t, q in synth.drive_poses(8, step=1.5):
        _feed((a, b), synth.lidar_scan(scene, t, q, rows=32, cols=512, noise_sigma=0.02, rng=rng, dropout=0.05), t, q)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 1500 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    m = pu.compare_meshes(a, b)
    assert m["triangles"] > 2000 and m["pos_bit_exact"]


def test_many_beams_per_voxel_follow_the_point_order(hip, oracle):
    """A dense fan of beams onto a near wall: tens of points per voxel in one scan — the sorted-record fold must apply
    them in ascending point index, like the oracle's sequential loop (D6)."""
    a, b = _pair(hip, oracle, dict(virtual_voxel_size=0.25, sdf_truncation=0.5))
    rng = np.random.default_rng(5)
    n = 60000
    yz = rng.uniform(-1.0, 1.0, (n, 2))
    pts = np.concatenate([np.full((n, 1), 6.0) + rng.normal(0, 0.03, (n, 1)), yz], axis=1).astype(np.float32)
    t, q = np.zeros(3, np.float32), np.array([0, 0, 0, 1], np.float32)
    _feed((a, b), pts, t, q)
    _feed((a, b), pts[::-1].copy(), t, q)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    d, v = a.dump_blocks()
    assert v["weight"].max() == 255  # clamped: far more than 255 updates reached some voxels


@pytest.mark.parametrize("buckets", ["1", "0"], ids=["buckets", "sorted"])
@pytest.mark.parametrize("n,spread", [(3000, 0.08), (30000, 0.08), (30000, 0.6)])
def test_runs_of_every_length_follow_the_point_order(hip, oracle, monkeypatch, buckets, n, spread):
    """Beams crowded onto a patch of a near wall: with a spread below the voxel size every beam crosses the same few voxels,
    so a voxel's run is as long as the scan (3 000: sorted inside the LDS by the bitonic network of k_scan_apply; 30 000:
    beyond the LDS, windows over the point index); with a wider patch the runs are a mix of tens to thousands.  The order
    inside a run decides the result (running mean in fp32, weight clamp at 255), so every path must agree with the oracle's
    sequential loop — twice, the second time with the points reversed."""
    monkeypatch.setenv("MRH_LIDAR_BUCKETS", buckets)
    a, b = _pair(hip, oracle, dict(virtual_voxel_size=0.25, sdf_truncation=0.5), blocks=16384)
    rng = np.random.default_rng(n + int(spread * 100))
    yz = rng.uniform(-spread, spread, (n, 2)) + 0.11
    pts = np.concatenate([np.full((n, 1), 4.0) + rng.normal(0, 0.05, (n, 1)), yz], axis=1).astype(np.float32)
    t, q = np.zeros(3, np.float32), np.array([0, 0, 0, 1], np.float32)
    _feed((a, b), pts, t, q)
    _feed((a, b), pts[::-1].copy(), t, q)
    _feed((a, b), pts[: n // 3].copy(), t, q)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] >= 1 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    assert not a.stats().error_flags


def test_far_from_origin_and_device_pointer(hip, oracle):
    from mrhash_amd import hipmem

    a, b = _pair(hip, oracle)
    scene = synth.street_canyon()
    off = np.array([900.0, -700.0, 120.0], np.float32)  # beyond the verified voxel->block shift range
    for t, q in synth.drive_poses(2, step=2.0):
        pts = synth.lidar_scan(scene, t, q, rows=16, cols=256)
        d_pts = hipmem.DeviceBuffer.from_numpy(pts.astype(np.float32))
        a.set_pose(synth.quat_to_rot(q), t + off)
        a.set_points_device(d_pts.ptr, len(pts))
        a.integrate_points()
        a.sync()
        b.set_pose(synth.quat_to_rot(q), t + off)
        b.upload_points(pts)
        b.integrate_points()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 300 and r["sdf_bit_exact"]


from test_lidar import spherical_camera  # noqa: E402


def _scan_pair(hip, oracle, params, cam=None, blocks=131072, max_depth=100.0):
    p = dict(synth.VBR_PARAMS, **params)
    k = cam or dict(fx=1.0, fy=1.0, cx=0.0, cy=0.0, rows=1, cols=1)
    out = []
    for lib in (hip, oracle):
        e = capi.Engine(lib, capi.Params(num_sdf_blocks=blocks, **p))
        e.set_camera(k["fx"], k["fy"], k["cx"], k["cy"], k["rows"], k["cols"], p["min_depth"], max_depth, model=1)
        out.append(e)
    return out


def _drive(engines, n, rows, cols, step=1.5, normals=False, noise=0.0, seed=0):
    scene = synth.street_canyon()
    rng = np.random.default_rng(seed)
    for t, q in synth.drive_poses(n, step=step):
        pts = synth.lidar_scan(scene, t, q, rows=rows, cols=cols, noise_sigma=noise, rng=rng)
        for e in engines:
            e.set_pose(synth.quat_to_rot(q), t)
            e.upload_points(pts)
            if normals:
                e.upload_normals(synth.scan_normals(pts))
            assert not e.integrate_points()


@pytest.mark.parametrize("sort", ["buckets", "own"])
def test_full_size_vbr_scans_match_oracle(hip, oracle, monkeypatch, sort):
    """BASELINE configs[4] at its stated size: 128 x 1024 = 131 072 points per scan, vbr.cfg parameters, three scans of a
    drive; occupancy, payload and mesh against the oracle — through the voxel buckets of mrh_scan.h (the default), through
    the sorted records of mrh_lidar.h with the scan-sized sort of mrh_sort.h (MRH_LIDAR_BUCKETS=0): the fold is order-dependent,
    so a path that did not keep a voxel's records in point order (or lost a record) would show up in the payload."""
    monkeypatch.setenv("MRH_LIDAR_BUCKETS", "1" if sort == "buckets" else "0")
    a, b = _scan_pair(hip, oracle, dict(min_weight_threshold=1), blocks=262144)
    _drive((a, b), 3, 128, 1024, step=2.0, noise=0.02)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 2000 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    m = pu.compare_meshes(a, b)
    assert m["triangles"] > 5000 and m["pos_bit_exact"]


def test_gc_and_starve_on_scans_match_oracle(hip, oracle):
    """garbageCollect on scans (voxel_data_structures.cpp:128-129): identify + free over every live block after each scan,
    the starve step every 3rd scan through the SPHERICAL camera (atan2 / asin of mrh_softmath.h on both sides)."""
    cam = spherical_camera(32, 512)
    a, b = _scan_pair(hip, oracle, dict(n_frames_invalidate_voxels=3, min_weight_threshold=1), cam)
    _drive((a, b), 8, 32, 512, noise=0.02)
    a.sync()
    sa, sb = a.stats(), b.stats()
    assert (sa.occupied_fine, sa.free_fine) == (sb.occupied_fine, sb.free_fine) and sa.error_flags == 0
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 500 and r["sdf_bit_exact"]
    pu.compare_meshes(a, b)
    keep = _scan_pair(hip, oracle, dict(min_weight_threshold=1), cam)[0]
    _drive((keep,), 8, 32, 512, noise=0.02)
    assert keep.stats().occupied_fine > sa.occupied_fine  # GC really freed blocks


def test_normal_direction_sdf_matches_oracle(hip, oracle):
    a, b = _scan_pair(hip, oracle, dict(projective_sdf=False, min_weight_threshold=1))
    _drive((a, b), 4, 32, 512, normals=True, noise=0.02)
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 800 and r["weighted"] > 10000 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    pu.compare_meshes(a, b)
    e = capi.Engine(hip, capi.Params(num_sdf_blocks=4096, **dict(synth.VBR_PARAMS, projective_sdf=False)))
    e.set_camera(1, 1, 0, 0, 1, 1, 0.2, 100.0, model=1)
    e.set_pose(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
    e.upload_points(np.array([[5.0, 0, 0]], np.float32))
    with pytest.raises(capi.MrhError) as ei:
        e.integrate_points()  # no normals given
    assert ei.value.code == capi.MRH_ERR_STATE
    e.close()


@pytest.mark.parametrize("gc", [0, 4])
def test_variance_adaptive_scans_match_oracle(hip, oracle, gc):
    """sdf_var_threshold > 0 on scans: coarsening over all live blocks, then the scan a second time into fine and coarse
    blocks (reintegrate3D, vds.cu:1561-1580); with and without garbage collection / starve."""
    cam = spherical_camera(32, 512)
    a, b = _scan_pair(hip, oracle, dict(sdf_var_threshold=0.05, n_frames_invalidate_voxels=gc, min_weight_threshold=1), cam)
    _drive((a, b), 6, 32, 512, step=0.5, noise=0.01)
    a.sync()
    sa, sb = a.stats(), b.stats()
    assert sb.occupied_coarse > 50
    assert (sa.occupied_fine, sa.occupied_coarse, sa.free_fine, sa.free_coarse) == (sb.occupied_fine, sb.occupied_coarse, sb.free_fine, sb.free_coarse)
    r = pu.compare_maps(a, b)
    assert r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    pu.compare_meshes(a, b)


def test_softmath_is_bit_identical_on_device_and_host(tmp_path):
    """include/mrh_softmath.h compiled for gfx950 and for the host by hipcc with the arithmetic-spec flags: sin, cos, atan2
    and asin of 2^20 random arguments each, every result the same bits on both sides (tools/micro/softmath_check.hip)."""
    import os
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(pu.ROOT, "tools", "micro", "softmath_check.hip")
    exe = str(tmp_path / "softmath_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-result",
                    src, "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mismatches sin 0 cos 0 atan2 0 asin 0" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("var_threshold", [0.0, 0.05])
def test_spherical_depth_images_match_oracle(hip, oracle, var_threshold):
    """The image path under the spherical camera model (camera.cuh:91-99, :147-164, :184-201): range images of the street
    scene through mrh_integrate — the general kernels with sin / cos / atan2 / asin from mrh_softmath.h — with GC every
    frame and starve every 3rd."""
    cam = spherical_camera(64, 512)
    p = dict(synth.VBR_PARAMS, min_weight_threshold=1, n_frames_invalidate_voxels=3, sdf_var_threshold=var_threshold)
    engines = []
    for lib in (hip, oracle):
        e = capi.Engine(lib, capi.Params(num_sdf_blocks=131072, **p))
        e.set_camera(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["rows"], cam["cols"], p["min_depth"], 60.0, model=1)
        engines.append(e)
    scene = synth.street_canyon()
    for t, q in synth.drive_poses(5, step=1.0):
        depth, rgb = synth.spherical_range_image(scene, t, q, cam)
        for e in engines:
            e.set_pose(synth.quat_to_rot(q), t)
            e.upload_depth(depth)
            e.upload_rgb(rgb)
            assert not e.integrate()
    a, b = engines
    a.sync()
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 800 and r["weighted"] > 50000 and r["sdf_bit_exact"]
    pu.compare_meshes(a, b)
    # a pinhole frame into the same context afterwards takes the fast path again (its GC summaries are rebuilt first)
    if var_threshold == 0.0:
        K = synth.CFG1
        f = synth.cfg1_sphere()
        for e in engines:
            e.set_camera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, p["min_depth"], 60.0)
            pu.feed(e, f)
        a.sync()
        pu.compare_maps(a, b)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_randomised_scans(hip, oracle, seed):
    """Random voxel size, range-dependent truncation, weight sample / max, integration distance and arbitrary sensor
    orientation (roll / pitch / yaw); projective or normal-direction SDF, garbage collection + starve through a spherical
    camera or none, single-resolution or variance-adaptive map: allocation, per-voxel update order, clamping, coarsening
    and collection against the oracle."""
    rng = np.random.default_rng(40 + seed)
    vs = float(rng.choice([0.1, 0.2, 0.35]))
    gc = int(rng.choice([0, 0, 2, 3]))
    params = dict(virtual_voxel_size=vs, sdf_truncation=float(rng.uniform(1.5, 3.0)) * vs,
                  sdf_truncation_scale=float(rng.choice([0.0, 0.005, 0.02])), integration_weight_sample=int(rng.integers(1, 5)),
                  integration_weight_max=int(rng.integers(8, 256)), min_weight_threshold=1, n_frames_invalidate_voxels=gc,
                  projective_sdf=bool(rng.random() < 0.6), sdf_var_threshold=float(rng.choice([0.0, 0.0, 0.05, 0.2])))
    rows, cols = int(rng.integers(8, 24)), int(rng.integers(128, 400))
    cam = spherical_camera(rows, cols) if gc else None
    a, b = _scan_pair(hip, oracle, params, cam, max_depth=float(rng.choice([25.0, 60.0, 100.0])))
    scene = synth.street_canyon()
    for t, _ in synth.drive_poses(5, step=float(rng.uniform(0.5, 3.0))):
        q = rng.normal(size=4)
        q = (q / np.linalg.norm(q)).astype(np.float32)
        pts = synth.lidar_scan(scene, t, q, rows=rows, cols=cols, noise_sigma=0.02, rng=rng, dropout=0.03)
        for e in (a, b):
            e.set_pose(synth.quat_to_rot(q), t)
            e.upload_points(pts)
            if not params["projective_sdf"]:
                e.upload_normals(synth.scan_normals(pts))
            assert not e.integrate_points()
    a.sync()
    sa, sb = a.stats(), b.stats()
    assert (sa.occupied_fine, sa.occupied_coarse, sa.free_fine) == (sb.occupied_fine, sb.occupied_coarse, sb.free_fine), params
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 20 and r["sdf_bit_exact"] and r["sumsq_bit_exact"], params
    pu.compare_meshes(a, b)


@pytest.mark.parametrize("pipe", ["1", "0"], ids=["pipelined", "serial"])
def test_spherical_images_on_the_two_launch_path_full_size(hip, oracle, monkeypatch, pipe):
    """Round 4: single-resolution maps under the spherical camera model take the two launches of the pinhole path (k_front / k_back
    templated on the model: rays through inverse_projection_m, getDepth(cloud) as the frame's depth, the approx-frustum predicate
    of camera.cuh:184-201 in the sweep, every voxel projected through sqrt / atan2 / asin of mrh_softmath.h) — pipelined or
    serial.  128 x 1024 range images of the street, 10 frames, GC every frame, starve on the 5th: map, statistics and mesh
    against the oracle."""
    monkeypatch.setenv("MRH_PIPE", pipe)
    cam = spherical_camera(128, 1024)
    p = dict(synth.VBR_PARAMS, min_weight_threshold=1, n_frames_invalidate_voxels=5)
    engines = []
    for lib in (hip, oracle):
        e = capi.Engine(lib, capi.Params(num_sdf_blocks=65536, **p))
        e.set_camera(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["rows"], cam["cols"], p["min_depth"], 100.0, model=1)
        engines.append(e)
    a, b = engines
    scene = synth.street_canyon()
    for i, (t, q) in enumerate(synth.drive_poses(10, step=0.5)):
        depth, rgb = synth.spherical_range_image(scene, t, q, cam)
        for e in engines:
            e.set_pose(synth.quat_to_rot(q), t)
            e.upload_depth(depth)
            e.upload_rgb(rgb)
            assert not e.integrate()
        if i == 6:
            sa, sb = a.stats(), b.stats()
            assert (sa.occupied_fine, sa.free_fine, sa.last_compact_blocks) == (sb.occupied_fine, sb.free_fine, sb.last_compact_blocks)
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 1500 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    pu.compare_meshes(a, b)
    assert a.stats().error_flags == 0
