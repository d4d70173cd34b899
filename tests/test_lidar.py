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


def test_unsupported_modes_say_so(oracle):
    e = _engine(oracle, dict(n_frames_invalidate_voxels=5))
    _identity(e)
    e.upload_points(np.array([[5.0, 0, 0]], np.float32))
    with pytest.raises(capi.MrhError) as ei:
        e.integrate_points()
    assert ei.value.code == capi.MRH_ERR_UNSUPPORTED
    e.close()
    e = _engine(oracle, dict(sdf_var_threshold=0.01))
    _identity(e)
    e.upload_points(np.array([[5.0, 0, 0]], np.float32))
    with pytest.raises(capi.MrhError):
        e.integrate_points()
    e.close()
