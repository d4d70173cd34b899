"""LiDAR scans (BASELINE.json configs[4]): oracle-side checks that run without a GPU.

The reference holds no golden values for allocBlocks3D / integrate3D (its tests cover the spherical projection model
only, tests/test_projections.cu:146-227), so the oracle is pinned here through properties of the restated kernels:
hand-computable single-point cases, heap conservation, and independence of the result from everything but the
canonical per-voxel update order (oracle header, D6)."""
import numpy as np
import pytest

import parity_utils as pu
from mrhash_amd import capi, synth

K1 = synth.Intrinsics(1.0, 1.0, 0.0, 0.0, 1, 1)  # any camera: only max_depth (integration distance) is used


def _engine(lib, params=None, blocks=32768, max_depth=None):
    p = dict(synth.VBR_PARAMS, **(params or {}))
    e = capi.Engine(lib, capi.Params(num_sdf_blocks=blocks, **p))
    e.set_camera(K1.fx, K1.fy, K1.cx, K1.cy, K1.rows, K1.cols, p["min_depth"], max_depth or p["max_depth"], model=1)
    return e


def _identity(e):
    e.set_pose(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))


def test_single_point_known_answer(oracle):
    """One return 10 m ahead on the x axis, voxel 0.2 m, truncation 0.4 m: the beam crosses voxels x = 48..52 (centres
    9.6 .. 10.4 m); sdf = range - |voxel centre| clamped to +-0.4; the far end (sdf == -0.4) stops the walk."""
    e = _engine(oracle)
    _identity(e)
    e.upload_points(np.array([[10.0, 0.0, 0.0]], np.float32))
    e.integrate_points()
    descs, vox = e.dump_blocks()
    got = {}
    for d, v in zip(descs, vox):
        for li in np.nonzero(v["weight"])[0]:
            x, y, z = li % 8, (li // 8) % 8, li // 64
            got[(d["x"] * 8 + x, d["y"] * 8 + y, d["z"] * 8 + z)] = (float(v["sdf"][li]), int(v["weight"][li]))
    assert set(got) == {(48, 0, 0), (49, 0, 0), (50, 0, 0), (51, 0, 0)}
    for (vx, _, _), (sdf, w) in got.items():
        assert w == 1
        assert sdf == pytest.approx(np.float32(10.0) - np.float32(vx * 0.2), abs=2e-6)
    # blocks along the segment [9.6, 10.4] m: x block 6 (voxels 48..55) only -> one block; colour stays black
    assert len(descs) == 1 and (descs["x"][0], descs["y"][0], descs["z"][0]) == (6, 0, 0)
    assert not vox["rgb"].any()
    e.close()


def test_out_of_range_and_empty_returns_are_ignored(oracle):
    e = _engine(oracle, max_depth=50.0)
    _identity(e)
    pts = np.array([[0, 0, 0], [60.0, 0, 0], [0, 1e-7, 0]], np.float32)
    e.upload_points(pts)
    e.integrate_points()
    st = e.stats()
    # (0,0,0): no return; 60 m > integration distance 50 m: allocBlocks3D clips the segment to nothing
    # (min(50, 59.6) >= min(50, 60.4)); 1e-7 m: allocated around the origin but integrate3D skips range < 1e-6
    descs, vox = e.dump_blocks()
    assert not vox["weight"].any()
    assert st.frames_integrated == 1
    e.close()


def test_heap_conservation_and_idempotent_allocation(oracle):
    e = _engine(oracle)
    scene = synth.street_canyon()
    (t, q), = synth.drive_poses(1)
    pts = synth.lidar_scan(scene, t, q, rows=16, cols=256)
    e.set_pose(synth.quat_to_rot(q), t)
    e.upload_points(pts)
    e.integrate_points()
    s1 = e.stats()
    assert s1.occupied_fine > 100 and s1.occupied_fine + s1.free_fine == 32768
    e.upload_points(pts)
    e.integrate_points()
    s2 = e.stats()
    assert s2.occupied_fine == s1.occupied_fine  # same scan, same blocks
    descs, vox = e.dump_blocks()
    assert vox["weight"].max() >= 2
    e.close()


def _voxel(e, vx, vy, vz):
    descs, vox = e.dump_blocks()
    for d, v in zip(descs, vox):
        if (d["x"], d["y"], d["z"]) == (vx >> 3, vy >> 3, vz >> 3):
            li = (vz & 7) * 64 + (vy & 7) * 8 + (vx & 7)
            return np.float32(v["sdf"][li]), np.float32(v["sum_squared"][li]), int(v["weight"][li])
    raise AssertionError("block not allocated")


def test_update_order_is_the_point_order(oracle):
    """D6: a voxel crossed by several beams of one scan receives its updates in ascending point index.  Three beams
    through voxel (50, 0, 0): running mean and variance term follow from combineVoxel (vhu.cuh:167-181) and
    vds.cu:1352-1366 in that order, computed here by hand in float32; the reversed scan gives the other value."""
    f = np.float32
    p0, p1 = np.array([10.0, 0.0, 0.0], np.float32), np.array([10.06, 0.012, 0.0], np.float32)
    p2 = np.array([9.93, -0.01, 0.004], np.float32)
    centre = f(50) * f(0.2)

    def sdf_of(p):
        rng = np.sqrt(f(f(p[0] * p[0]) + f(p[1] * p[1])) + f(p[2] * p[2]), dtype=np.float32)
        return f(rng - np.sqrt(f(centre * centre), dtype=np.float32))

    def fold(order):
        s, w, ss = f(0), 0, f(0)
        half = f(f(0.2) / f(2))
        for p in order:
            sd = sdf_of(p)
            mean = s if w > 0 else f(0)
            delta = f(f(sd - mean) / half)
            s = f(f(f(s * f(w)) + f(sd * f(1))) / f(w + 1))
            w += 1
            ss = f(f(0) + f(delta * f(f(sd - s) / half)))
        return s, ss, w

    for order in ((p0, p1, p2), (p2, p1, p0)):
        e = _engine(oracle)
        _identity(e)
        e.upload_points(np.stack(order))
        e.integrate_points()
        got = _voxel(e, 50, 0, 0)
        want = fold(order)
        assert (got[0].tobytes(), got[1].tobytes(), got[2]) == (want[0].tobytes(), want[1].tobytes(), want[2])
        e.close()
    assert fold((p0, p1, p2))[1] != fold((p2, p1, p0))[1]  # the order is observable, so it has to be pinned


spherical_camera = synth.spherical_camera


def _scan_engine(lib, params, cam=None, blocks=65536, max_depth=100.0):
    p = dict(synth.VBR_PARAMS, **params)
    e = capi.Engine(lib, capi.Params(num_sdf_blocks=blocks, **p))
    k = cam or dict(fx=1.0, fy=1.0, cx=0.0, cy=0.0, rows=1, cols=1)
    e.set_camera(k["fx"], k["fy"], k["cx"], k["cy"], k["rows"], k["cols"], p["min_depth"], max_depth, model=1)
    return e


def drive(e, n=4, rows=16, cols=256, normals=False, step=1.5):
    scene = synth.street_canyon()
    for t, q in synth.drive_poses(n, step=step):
        pts = synth.lidar_scan(scene, t, q, rows=rows, cols=cols)
        e.set_pose(synth.quat_to_rot(q), t)
        e.upload_points(pts)
        if normals:
            e.upload_normals(synth.scan_normals(pts))
        e.integrate_points()


def test_gc_on_scans_frees_blocks_without_a_surface_and_conserves_the_heap(oracle):
    """garbageCollect after every scan (voxel_data_structures.cpp:128-129): identify + free over EVERY live block (the
    scan path compacts without a camera); every 3rd scan the starve step projects through the spherical camera."""
    cam = spherical_camera(16, 256)
    keep = _scan_engine(oracle, dict(n_frames_invalidate_voxels=0), cam)
    gc = _scan_engine(oracle, dict(n_frames_invalidate_voxels=3), cam)
    drive(keep, 7)
    drive(gc, 7)
    sk, sg = keep.stats(), gc.stats()
    assert sg.occupied_fine < sk.occupied_fine and sg.occupied_fine > 100
    assert sg.occupied_fine + sg.free_fine == sg.num_sdf_blocks
    dk, vk = keep.dump_blocks()
    dg, vg = gc.dump_blocks()
    assert np.all(np.isin(dg, dk))  # GC only removes
    # a surviving block holds a weighted voxel inside the truncation band (vds.cu:1708-1711)
    w = vg["weight"] > 0
    assert np.all((np.where(w, np.abs(vg["sdf"]), np.inf).min(axis=1) < np.float32(0.4)))
    # the starve step took weight off the front-most voxel of some pixels: total weight below the no-GC run on common blocks
    common = np.isin(dk, dg)
    assert vg["weight"].astype(np.int64).sum() < vk["weight"][common].astype(np.int64).sum()
    for e in (keep, gc):
        e.close()


def test_normal_direction_sdf_needs_and_uses_the_normals(oracle):
    e = _scan_engine(oracle, dict(projective_sdf=False))
    _identity(e)
    e.upload_points(np.array([[5.0, 0, 0]], np.float32))
    with pytest.raises(capi.MrhError) as ei:
        e.integrate_points()  # no normals given
    assert ei.value.code == capi.MRH_ERR_STATE
    e.close()
    maps = []
    for flip in (1.0, -1.0):
        e = _scan_engine(oracle, dict(projective_sdf=False, min_weight_threshold=1))
        scene = synth.street_canyon()
        for t, q in synth.drive_poses(2, step=2.0):
            pts = synth.lidar_scan(scene, t, q, rows=16, cols=256)
            e.set_pose(synth.quat_to_rot(q), t)
            e.upload_points(pts)
            e.upload_normals(flip * synth.scan_normals(pts))
            e.integrate_points()
        d, v = e.dump_blocks()
        assert len(d) > 200 and (v["weight"] > 0).sum() > 1000
        maps.append((d, v))
        e.close()
    # along the normal the SDF is dot(voxel - point, normal) (vds.cu:1322-1326): flipping the normals flips its sign
    proj = _scan_engine(oracle, dict(min_weight_threshold=1))
    drive(proj, 2, step=2.0)
    dp, vp = proj.dump_blocks()
    assert not (len(dp) == len(maps[0][0]) and np.array_equal(vp["sdf"], maps[0][1]["sdf"]))
    assert not (len(maps[0][0]) == len(maps[1][0]) and np.array_equal(maps[0][1]["sdf"], maps[1][1]["sdf"]))
    proj.close()


def test_variance_adaptive_scans_coarsen_flat_regions(oracle):
    """sdf_var_threshold > 0 on scans: checkVarSDF over every live block, coarsened blocks are re-created at 4^3 and the
    whole scan is integrated a second time (reintegrate3D launches integrate3DKernel, vds.cu:1561-1580)."""
    e = _scan_engine(oracle, dict(sdf_var_threshold=0.05, min_weight_threshold=1))
    drive(e, 4, step=0.5)
    s = e.stats()
    assert s.occupied_coarse > 20 and s.occupied_fine > 100
    d, v = e.dump_blocks()
    coarse = d["resolution"] == 1
    assert coarse.sum() == s.occupied_coarse
    assert (v["weight"][coarse][:, :64] > 0).any() and not (v["weight"][coarse][:, 64:] > 0).any()
    # fine slots + coarse units account for the whole pool: 8 coarse units per converted fine slot
    assert (s.free_coarse + s.occupied_coarse) % 8 == 0
    assert s.occupied_fine + s.free_fine + (s.free_coarse + s.occupied_coarse) // 8 == s.num_sdf_blocks
    e.close()


def test_spherical_depth_image_fuses_like_the_same_points_would_project(oracle):
    """The image path with the spherical camera model (camera.cuh:91-99, :147-164): a range image of the street scene;
    allocation walks the back-projected rays, integration projects voxels with atan2 / asin (mrh_softmath.h)."""
    cam = spherical_camera(32, 256)
    p = dict(synth.VBR_PARAMS, min_weight_threshold=1)
    e = capi.Engine(oracle, capi.Params(num_sdf_blocks=65536, **p))
    e.set_camera(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["rows"], cam["cols"], p["min_depth"], 60.0, model=1)
    scene = synth.street_canyon()
    (t, q), = synth.drive_poses(1)
    depth, rgb = synth.spherical_range_image(scene, t, q, cam)
    e.set_pose(synth.quat_to_rot(q), t)
    e.upload_depth(depth)
    e.upload_rgb(rgb)
    e.integrate()
    d, v = e.dump_blocks()
    assert len(d) > 500
    w = v["weight"] > 0
    assert w.sum() > 20000 and np.abs(v["sdf"][w]).max() <= np.float32(0.4)
    # known answer: a weighted voxel's sdf is (range image at its pixel) - |voxel in the sensor frame|, clamped
    R = synth.quat_to_rot(q).astype(np.float64)
    checked = 0
    for bi in np.argsort(-w.sum(axis=1))[:5]:
        for li in np.nonzero(w[bi])[0][:40]:
            vx = np.array([d["x"][bi] * 8 + li % 8, d["y"][bi] * 8 + (li // 8) % 8, d["z"][bi] * 8 + li // 64]) * 0.2
            pc = R.T @ (vx - t.astype(np.float64))
            rng = np.linalg.norm(pc)
            col = int(cam["fx"] * np.arctan2(pc[1], pc[0]) + cam["cx"] + 0.5)
            row = int(cam["fy"] * np.arcsin(pc[2] / rng) + cam["cy"] + 0.5)
            want = np.clip(depth[row, col] - rng, -0.4, 0.4)
            if abs(cam["fx"] * np.arctan2(pc[1], pc[0]) + cam["cx"] + 0.5 - round(cam["fx"] * np.arctan2(pc[1], pc[0]) + cam["cx"] + 0.5)) < 1e-3:
                continue  # on a pixel boundary: the ulp-level differences of the soft functions may pick the neighbour
            assert abs(float(v["sdf"][bi][li]) - want) < 2e-4
            checked += 1
    assert checked > 100
    e.close()


def test_scan_layout_detection_is_host_code_and_finds_the_row_length():
    """mrh_detect_scan_layout (what mrh_upload_points concludes about a host cloud; pure host code, no device): organised scans
    of the usual sensors are recognised through their row length — also with missing returns, also stored column by column —,
    shuffled or small clouds are not.  The answer only orders the beams of the HIP path (tests/test_lidar_gpu.py: same map)."""
    hip = capi.load_hip()
    scene = synth.street_canyon()
    (t, q), = synth.drive_poses(1)

    def detect(pts):
        a = np.ascontiguousarray(pts, dtype=np.float32)
        return hip.mrh_detect_scan_layout(a.ctypes.data, a.shape[0])

    for rows, cols in ((128, 1024), (64, 2048), (32, 512), (16, 1024)):
        pts = synth.lidar_scan(scene, t, q, rows=rows, cols=cols)
        assert detect(pts) == cols, (rows, cols)
    pts = synth.lidar_scan(scene, t, q, rows=128, cols=1024, dropout=0.3, rng=np.random.default_rng(3))
    assert detect(pts) == 1024
    col_major = np.ascontiguousarray(synth.lidar_scan(scene, t, q, rows=128, cols=1024).reshape(128, 1024, 3).transpose(1, 0, 2)).reshape(-1, 3)
    assert detect(col_major) == 128  # rows of the array = columns of the sensor: organised all the same
    rng = np.random.default_rng(5)
    assert detect(rng.permutation(synth.lidar_scan(scene, t, q, rows=128, cols=1024))) == 0
    assert detect(rng.normal(size=(131072, 3))) == 0
    assert detect(synth.lidar_scan(scene, t, q, rows=8, cols=256)) == 0  # too small to bother
    assert detect(np.zeros((16384, 3), np.float32)) == 0  # no returns at all
