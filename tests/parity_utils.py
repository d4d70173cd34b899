"""Shared helpers for the parity tests: drive the HIP library and the CPU oracle through the same
`capi.Engine` class and compare their canonicalised outputs.

Bars (BASELINE.json north_star): occupancy list, triangle index buffer, u8 colour/weight -> bit-exact;
TSDF values, sum_squared, vertex positions -> |diff| <= 1e-5 (they are in practice bit-identical because
both sides share one arithmetic spec; the tests report the max ulp distance they saw).
"""
from __future__ import annotations

import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mrhash_amd import capi, synth  # noqa: E402

ORACLE_PATH = os.path.join(ROOT, "oracle", "_build", "libmrh_oracle.so")
FLOAT_TOL = 1e-5

_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_PATH) or os.path.getmtime(ORACLE_PATH) < os.path.getmtime(
            os.path.join(ROOT, "oracle", "mrh_oracle.c")
        ):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
        _oracle = capi.load_library(ORACLE_PATH)
    return _oracle


def make_engine(lib, K: synth.Intrinsics, params: dict, num_sdf_blocks: int = 65536, **extra) -> capi.Engine:
    p = capi.Params(num_sdf_blocks=num_sdf_blocks, **{**params, **extra})
    e = capi.Engine(lib, p)
    e.set_camera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, p.min_depth, p.max_depth)
    return e


def feed(e: capi.Engine, f: synth.Frame, n_frames_invalidate: int = -1, dist=None):
    e.set_pose(f.R, f.t)
    e.upload_depth(f.depth)
    e.upload_rgb(f.rgb)
    if dist is not None:
        from mrhash_amd import parallel

        parallel.integrate(e, dist, n_frames_invalidate)
    else:
        assert not e.integrate(n_frames_invalidate), "sharded context needs parallel.integrate"


def _max_abs_diff(x: np.ndarray, y: np.ndarray) -> float:
    """max |x - y| where NaNs must sit at the same places on both sides (a NaN depth pixel propagates a NaN
    TSDF value through the reference's formulas; both implementations must agree on where)."""
    nx, ny = np.isnan(x), np.isnan(y)
    assert np.array_equal(nx, ny), "NaN positions differ"
    if nx.all():
        return 0.0
    return float(np.max(np.abs(x[~nx] - y[~nx]), initial=0.0))


def compare_maps(a: capi.Engine, b: capi.Engine, tol: float = FLOAT_TOL) -> dict:
    """Asserts canonical occupancy + payload parity; returns summary numbers."""
    da, va = a.dump_blocks()
    db, vb = b.dump_blocks()
    assert len(da) == len(db), f"occupancy differs: {len(da)} vs {len(db)} blocks"
    assert np.array_equal(da, db), "occupancy list (x,y,z,resolution) differs"
    assert np.array_equal(va["weight"], vb["weight"]), "voxel weights differ"
    assert np.array_equal(va["rgb"], vb["rgb"]), "voxel colours differ"
    dsdf = _max_abs_diff(va["sdf"], vb["sdf"])
    dss = _max_abs_diff(va["sum_squared"], vb["sum_squared"])
    assert dsdf <= tol, f"TSDF values differ by {dsdf}"
    assert dss <= tol, f"sum_squared differs by {dss}"
    bit_sdf = bool(np.array_equal(va["sdf"].view(np.uint32), vb["sdf"].view(np.uint32)))
    bit_ss = bool(np.array_equal(va["sum_squared"].view(np.uint32), vb["sum_squared"].view(np.uint32)))
    return dict(blocks=len(da), weighted=int((va["weight"] > 0).sum()), max_dsdf=dsdf, max_dss=dss,
                sdf_bit_exact=bit_sdf, sumsq_bit_exact=bit_ss)


def compare_meshes(a: capi.Engine, b: capi.Engine, tol: float = FLOAT_TOL) -> dict:
    ta = a.extract_triangles()
    tb = b.extract_triangles()
    assert ta.shape == tb.shape, f"triangle count differs: {ta.shape[0]} vs {tb.shape[0]}"
    dp = _max_abs_diff(ta["p"], tb["p"])
    dc = _max_abs_diff(ta["c"], tb["c"])
    assert dp <= tol, f"triangle vertex positions differ by {dp}"
    assert dc <= tol * 255, f"triangle vertex colours differ by {dc}"
    Va, Fa, Ca = a.extract_mesh()
    Vb, Fb, Cb = b.extract_mesh()
    assert Fa.shape == Fb.shape and np.array_equal(Fa, Fb), "triangle index buffer differs"
    assert Va.shape == Vb.shape
    dv = _max_abs_diff(Va, Vb)
    assert dv <= tol, f"mesh vertices differ by {dv}"
    return dict(triangles=int(ta.shape[0]), vertices=int(Va.shape[0]), faces=int(Fa.shape[0]), max_dp=dp,
                pos_bit_exact=bool(np.array_equal(ta["p"].view(np.uint32), tb["p"].view(np.uint32))))


def frame_from_spec(spec: dict) -> synth.Frame:
    """Rebuilds a cfg1 frame from the tiny description stored in the golden fixture."""
    if spec["kind"] == "plane":
        return synth.cfg1_plane(z=spec["z"])
    if spec["kind"] == "sphere":
        return synth.cfg1_sphere(radius=spec.get("radius", 0.5), zc=spec["zc"])
    raise ValueError(spec)


def make_lidar_engine(lib, params: dict, max_depth: float, num_sdf_blocks: int = 32768) -> capi.Engine:
    """Engine for point-cloud integration: any camera, only max_depth (the integration distance) matters."""
    e = capi.Engine(lib, capi.Params(num_sdf_blocks=num_sdf_blocks, **params))
    e.set_camera(1.0, 1.0, 0.0, 0.0, 1, 1, params["min_depth"], max_depth, model=1)
    return e


def lidar_scans_from_spec(spec: dict):
    """(t, q, points) of the golden LiDAR cases: street-canyon scene, drive poses; points are rounded to 1 mm so that
    the fixture does not depend on the last bits of the host's trigonometric functions."""
    scene = synth.street_canyon()
    for t, q in synth.drive_poses(spec["n"], step=spec["step"]):
        pts = synth.lidar_scan(scene, t, q, rows=spec["rows"], cols=spec["cols"])
        pts = (np.round(pts.astype(np.float64) * 1000.0) / 1000.0).astype(np.float32)
        yield t, q, pts
