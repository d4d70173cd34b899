"""include/mrh_softmath.h (sin / cos / atan2 / asin in plain fp32, shared verbatim by the HIP kernels and the oracle —
deviation D8) against numpy on dense grids: close enough to the correctly rounded values that a projected pixel can only
differ from a libm / CUDA evaluation on a rounding boundary.  The device side of "shared verbatim" is
tools/micro/softmath_check.hip (run by tests/test_lidar_gpu.py: 0 differing bits between device and host)."""
import ctypes

import numpy as np

import parity_utils as pu


def _fn():
    o = pu.oracle_lib()
    o.orc_softmath.restype = ctypes.c_float
    o.orc_softmath.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_float]
    return lambda op, a, b=0.0: o.orc_softmath(op, float(a), float(b))


def test_sin_cos_on_the_cameras_range():
    f = _fn()
    xs = np.linspace(-np.pi, np.pi, 40001).astype(np.float32)
    s = np.array([f(0, x) for x in xs], np.float32)
    c = np.array([f(1, x) for x in xs], np.float32)
    assert np.max(np.abs(s - np.sin(xs.astype(np.float64)))) < 1.5e-7
    assert np.max(np.abs(c - np.cos(xs.astype(np.float64)))) < 1.5e-7
    assert f(0, 0.0) == 0.0 and f(1, 0.0) == 1.0
    # beyond the camera's range the reduction still holds (|x| <= 8192)
    big = np.linspace(-100.0, 100.0, 5001).astype(np.float32)
    assert np.max(np.abs(np.array([f(0, x) for x in big]) - np.sin(big.astype(np.float64)))) < 2e-6


def test_asin_and_atan2():
    f = _fn()
    ys = np.linspace(-1.0, 1.0, 40001).astype(np.float32)
    a = np.array([f(3, y) for y in ys], np.float32)
    assert np.max(np.abs(a - np.arcsin(ys.astype(np.float64)))) < 3e-7
    assert f(3, 1.0) == np.float32(np.pi / 2) and f(3, -1.0) == np.float32(-np.pi / 2) and f(3, 2.0) == np.float32(np.pi / 2)
    rng = np.random.default_rng(0)
    P = rng.normal(size=(40000, 2)).astype(np.float32)
    t = np.array([f(2, p[0], p[1]) for p in P], np.float32)
    assert np.max(np.abs(t - np.arctan2(P[:, 0].astype(np.float64), P[:, 1].astype(np.float64)))) < 4e-7
    # quadrants and axes, C convention
    assert f(2, 0.0, 1.0) == 0.0 and f(2, 1.0, 0.0) == np.float32(np.pi / 2) and f(2, -1.0, 0.0) == np.float32(-np.pi / 2)
    assert abs(f(2, 0.0, -1.0) - np.pi) < 1e-6 and f(2, 0.0, 0.0) == 0.0
    assert abs(f(2, -1e-3, -1.0) + np.pi) < 1.1e-3


def test_pixel_decisions_agree_with_libm_away_from_rounding_boundaries():
    """The spherical projection of 10^5 random points with the soft functions and with numpy: the integer pixel differs
    only where the real-valued coordinate is within 1e-4 px of a rounding boundary."""
    f = _fn()
    rng = np.random.default_rng(1)
    P = rng.normal(size=(100000, 3)).astype(np.float32) * np.float32(20.0)
    fx, fy, cx, cy = np.float32(512 / (2 * np.pi)), np.float32(63 / np.deg2rad(45.0)), np.float32(256.0), np.float32(31.5)
    r = np.sqrt((P.astype(np.float64) ** 2).sum(1))
    u = fx * np.arctan2(P[:, 1].astype(np.float64), P[:, 0].astype(np.float64)) + cx + 0.5
    v = fy * np.arcsin(np.clip(P[:, 2] / r, -1, 1)) + cy + 0.5
    bad = 0
    for i in range(0, len(P), 7):
        rr = np.float32(np.sqrt(np.float32(P[i, 0] * P[i, 0] + P[i, 1] * P[i, 1]) + np.float32(P[i, 2] * P[i, 2])))
        col = int(np.float32(np.float32(fx * np.float32(f(2, P[i, 1], P[i, 0]))) + cx) + np.float32(0.5))
        row = int(np.float32(np.float32(fy * np.float32(f(3, np.float32(P[i, 2] / rr)))) + cy) + np.float32(0.5))
        if col != int(u[i]) and abs(u[i] - round(u[i])) > 1e-4:
            bad += 1
        if row != int(v[i]) and abs(v[i] - round(v[i])) > 1e-4 and v[i] > 0:
            bad += 1
    assert bad == 0
