"""Seeded synthetic RGB-D streams standing in for the datasets BASELINE.json names.

No dataset is reachable from the build or GPU box, so each BASELINE config is met by an
analytic scene that copies the dataset's intrinsics, resolution, depth quantisation and the
reference's parameter file:

  cfg1  128x128 plane / sphere, identity pose            (reference test fixture shape,
                                                           tests/test_hash_utils.cu:192-241)
  cfg2  "Replica room0": 6x4x3 m box room, 200-pose orbit, Replica K rescaled to 640x480,
        depth quantised to 1/6553.5 m                     (configurations/replica.cfg:2-23)
  cfg4  "ScanNet scene0000": 8x6x3 m furnished box room, ScanNet K, 1/5000 m quantisation,
        seeded hand-held walk                             (configurations/scannet.cfg:2-23)

World frame: camera-optical convention, z forward / x right / y down, motion in the x-z plane
(tests/test_utils.cuh:20-32).  Pixel rays use the reference's pixel-centre convention
(camera.cuh:88): dir = (ifx*(col-cx-0.5), ify*(row-cy-0.5), 1), so depth == ray parameter.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class Intrinsics:
    fx: float
    fy: float
    cx: float
    cy: float
    rows: int
    cols: int


REPLICA_NATIVE = Intrinsics(600.0, 600.0, 599.5, 339.5, 680, 1200)  # replica.cfg:20-21
REPLICA_640 = Intrinsics(600.0 * 640 / 1200, 600.0 * 480 / 680, 599.5 * 640 / 1200, 339.5 * 480 / 680, 480, 640)
SCANNET = Intrinsics(577.590698, 578.729797, 318.905426, 242.683609, 480, 640)  # scannet.cfg:20-21
CFG1 = Intrinsics(128.0, 128.0, 64.0, 64.0, 128, 128)


def quat_to_rot(q: Sequence[float]) -> np.ndarray:
    """Eigen::Quaternionf(qw,qx,qy,qz).toRotationMatrix() restated in float32
    (geowrapper.cpp:88-91; Eigen 3.4.0 Quaternion.h, no normalisation)."""
    x, y, z, w = (np.float32(v) for v in q)
    two = np.float32(2)
    tx, ty, tz = two * x, two * y, two * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = np.float32(1)
    return np.array(
        [
            [one - (tyy + tzz), txy - twz, txz + twy],
            [txy + twz, one - (txx + tzz), tyz - twx],
            [txz - twy, tyz + twx, one - (txx + tyy)],
        ],
        dtype=np.float32,
    )


def yaw_quat(angle: float) -> np.ndarray:
    """Rotation about the (down-pointing) y axis as (qx,qy,qz,qw)."""
    return np.array([0.0, np.sin(angle / 2), 0.0, np.cos(angle / 2)], dtype=np.float32)


def pixel_rays(K: Intrinsics, offset: float = 0.5) -> np.ndarray:
    """[rows, cols, 3] float64 camera-frame ray directions with z == 1.  `offset` = 0.5 is the reference's back-projection
    (inverseProjection, camera.cuh:88: pixel (r, c) looks along (c - cx - 0.5, r - cy - 0.5)); its forward projection rounds
    fx x / z + cx to the nearest pixel (camera.cuh:137-138), i.e. places the same pixel half a pixel further — `offset` = 0
    renders for THAT convention (tests/test_parity_gpu.py::test_mesh_accuracy_against_the_analytic_room)."""
    c = np.arange(K.cols, dtype=np.float64)[None, :]
    r = np.arange(K.rows, dtype=np.float64)[:, None]
    x = (c - K.cx - offset) / K.fx + 0 * r
    y = (r - K.cy - offset) / K.fy + 0 * c
    return np.stack([x, y, np.ones_like(x)], axis=-1)


@dataclass
class Box:
    lo: Tuple[float, float, float]
    hi: Tuple[float, float, float]


@dataclass
class Scene:
    """Axis-aligned room seen from inside plus solid axis-aligned boxes seen from outside."""

    room: Box
    furniture: List[Box] = field(default_factory=list)
    checker: float = 0.10  # metres
    seed: int = 0

    def cast(self, K: Intrinsics, R: np.ndarray, t: np.ndarray, pixel_offset: float = 0.5) -> Tuple[np.ndarray, np.ndarray]:
        """depth [rows, cols] float64 (ray parameter == camera z) and hit points [rows, cols, 3]."""
        return self.cast_dirs(pixel_rays(K, pixel_offset), R, t)

    def cast_dirs(self, d_cam: np.ndarray, R: np.ndarray, t: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Ray parameter of the first hit along sensor-frame directions d_cam [..., 3] and the hit points (world)."""
        d = d_cam @ np.asarray(R, dtype=np.float64).T
        o = np.asarray(t, dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
            lo, hi = np.array(self.room.lo), np.array(self.room.hi)
            t_exit = np.where(d > 0, (hi - o) * inv, np.where(d < 0, (lo - o) * inv, np.inf)).min(axis=-1)
            depth = t_exit
            for b in self.furniture:
                blo, bhi = np.array(b.lo), np.array(b.hi)
                t0 = (blo - o) * inv
                t1 = (bhi - o) * inv
                tn = np.minimum(t0, t1)
                tf = np.maximum(t0, t1)
                par = d == 0
                inside = (o >= blo) & (o <= bhi)
                tn = np.where(par, np.where(inside, -np.inf, np.inf), tn)
                tf = np.where(par, np.where(inside, np.inf, -np.inf), tf)
                t_in = tn.max(axis=-1)
                t_out = tf.min(axis=-1)
                hit = (t_in <= t_out) & (t_in > 0)
                depth = np.where(hit & (t_in < depth), t_in, depth)
        pts = o + d * depth[..., None]
        return depth, pts

    def color(self, pts: np.ndarray) -> np.ndarray:
        """Seeded world-space checker: uint8 [rows, cols, 3]."""
        cell = np.floor(pts / self.checker + 1e-6).astype(np.int64)
        h = (cell[..., 0] * 73856093) ^ (cell[..., 1] * 19349669) ^ (cell[..., 2] * 83492791) ^ (self.seed * 2654435761)
        h = (h ^ (h >> 13)) * 1274126177
        h ^= h >> 16
        rgb = np.stack([(h >> 0) & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF], axis=-1)
        return rgb.astype(np.uint8)


@dataclass
class Frame:
    t: np.ndarray  # float32 [3]
    q: np.ndarray  # float32 [4] (qx,qy,qz,qw)
    R: np.ndarray  # float32 [3,3] == quat_to_rot(q)
    depth: np.ndarray  # float32 [rows, cols]
    rgb: np.ndarray  # uint8 [rows, cols, 3]


def render(scene: Scene, K: Intrinsics, t: np.ndarray, q: np.ndarray, depth_scaling: Optional[float] = None,
           noise_sigma: float = 0.0, rng: Optional[np.random.Generator] = None, max_depth: float = 30.0, pixel_offset: float = 0.5) -> Frame:
    R = quat_to_rot(q)
    depth, pts = scene.cast(K, R, t, pixel_offset)
    if noise_sigma > 0:
        depth = depth + (rng or np.random.default_rng(0)).normal(0.0, noise_sigma, size=depth.shape)
    if depth_scaling:
        # dataset PNGs store round(depth * scaling) as uint16 (replica.cfg:22, scannet.cfg:22)
        depth = np.clip(np.rint(depth * depth_scaling), 0, 65535) / depth_scaling
    depth = np.where(np.isfinite(depth), depth, 0.0)
    return Frame(np.asarray(t, np.float32), np.asarray(q, np.float32), R, depth.astype(np.float32), scene.color(pts))


# ---- the BASELINE configs ---------------------------------------------------------------------

def cfg1_plane(z: float = 1.0) -> Frame:
    """cfg1: constant-depth plane, identity pose, uniform colour (test_hash_utils.cu:192-241 shape)."""
    K = CFG1
    depth = np.full((K.rows, K.cols), z, dtype=np.float32)
    rgb = np.zeros((K.rows, K.cols, 3), dtype=np.uint8)
    rgb[..., 0] = 255
    q = np.array([0, 0, 0, 1], dtype=np.float32)
    return Frame(np.zeros(3, np.float32), q, quat_to_rot(q), depth, rgb)


def cfg1_sphere(radius: float = 0.5, zc: float = 1.5, background: float = 0.0) -> Frame:
    """cfg1: sphere of `radius` centred at (0,0,zc) in front of an identity-pose camera."""
    K = CFG1
    d = pixel_rays(K)
    a = (d * d).sum(-1)
    b = -2.0 * d[..., 2] * zc
    c = zc * zc - radius * radius
    disc = b * b - 4 * a * c
    with np.errstate(invalid="ignore"):
        tt = (-b - np.sqrt(disc)) / (2 * a)
    depth = np.where(disc > 0, tt, background).astype(np.float32)
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, size=(K.rows, K.cols, 3), dtype=np.uint8)
    q = np.array([0, 0, 0, 1], dtype=np.float32)
    return Frame(np.zeros(3, np.float32), q, quat_to_rot(q), depth, rgb)


def replica_room() -> Scene:
    return Scene(Box((-3.0, -1.5, -2.0), (3.0, 1.5, 2.0)), seed=0)


def scannet_room() -> Scene:
    rng = np.random.default_rng(0)
    furn = []
    for _ in range(8):
        cx, cz = rng.uniform(-3.2, 3.2), rng.uniform(-2.2, 2.2)
        if abs(cx) < 1.0 and abs(cz) < 1.0:
            cx += 1.6
        sx, sz, sy = rng.uniform(0.3, 0.9), rng.uniform(0.3, 0.9), rng.uniform(0.4, 1.2)
        furn.append(Box((cx - sx, 1.5 - sy, cz - sz), (cx + sx, 1.5, cz + sz)))  # standing on the floor (y down)
    return Scene(Box((-4.0, -1.5, -3.0), (4.0, 1.5, 3.0)), furn, seed=1)


def street_canyon() -> Scene:
    """BASELINE configs[4] stand-in ("VBR"): a 100 m street between box buildings, sensor frame x forward / z up."""
    rng = np.random.default_rng(4)
    furn = []
    for side in (-1.0, 1.0):
        x = -48.0
        while x < 46.0:
            w, d, h = rng.uniform(6, 14), rng.uniform(4, 9), rng.uniform(5, 18)
            y0 = side * rng.uniform(6.0, 9.0)
            furn.append(Box((x, min(y0, y0 + side * d), -1.8), (x + w, max(y0, y0 + side * d), -1.8 + h)))
            x += w + rng.uniform(1.0, 5.0)
    for _ in range(10):  # parked cars / kiosks on the road side
        cx, cy = rng.uniform(-40, 40), rng.choice([-1.0, 1.0]) * rng.uniform(3.0, 5.0)
        furn.append(Box((cx - 2.2, cy - 0.9, -1.8), (cx + 2.2, cy + 0.9, -0.3)))
    return Scene(Box((-50.0, -30.0, -1.8), (50.0, 30.0, 40.0)), furn, seed=4)


def lidar_dirs(rows: int, cols: int, el_deg: Tuple[float, float] = (-22.5, 22.5)) -> np.ndarray:
    """Unit directions of a spinning LiDAR, [rows * cols, 3], azimuth fastest (layout of test_projections.cu:146-158)."""
    az = (np.arange(cols, dtype=np.float64) + 0.5) / cols * 2.0 * np.pi - np.pi
    el = np.deg2rad(np.linspace(el_deg[0], el_deg[1], rows))
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (rows, cols))], axis=-1)
    return d.reshape(-1, 3)


def lidar_scan(scene: Scene, t: np.ndarray, q: np.ndarray, rows: int = 32, cols: int = 512, noise_sigma: float = 0.0,
               rng: Optional[np.random.Generator] = None, max_range: float = 120.0, dropout: float = 0.0) -> np.ndarray:
    """One scan as float32 [N, 3] points in the SENSOR frame (what GeoWrapper.setPointCloud receives); missing returns
    are (0, 0, 0), as the reference's matrix initialisation leaves them (vds.cu:1234)."""
    R = quat_to_rot(q)
    d = lidar_dirs(rows, cols)
    rng_len, _ = scene.cast_dirs(d, R, t)
    if noise_sigma > 0:
        rng_len = rng_len + (rng or np.random.default_rng(0)).normal(0.0, noise_sigma, size=rng_len.shape)
    pts = d * rng_len[:, None]
    bad = ~np.isfinite(rng_len) | (rng_len > max_range) | (rng_len <= 0)
    if dropout > 0:
        bad |= (rng or np.random.default_rng(0)).random(rng_len.shape) < dropout
    pts[bad] = 0.0
    return pts.astype(np.float32)


def scan_normals(pts: np.ndarray) -> np.ndarray:
    """A normal per point of a scan, oriented towards the sensor (what the reference's MAD-tree hands out,
    geowrapper.cpp:386-388): here simply the reversed beam direction tilted by a fixed, point-dependent amount — a
    deterministic stand-in, not an estimator (estimating normals is the caller's business, include/mrhash_hip.h)."""
    p = pts.astype(np.float64)
    r = np.linalg.norm(p, axis=1, keepdims=True)
    n = -p / np.maximum(r, 1e-9)
    tilt = np.stack([np.sin(0.7 * p[:, 1]), np.cos(1.3 * p[:, 0]), np.sin(0.9 * p[:, 2] + 0.4)], axis=1) * 0.35
    n = n + tilt
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-9)
    n[r[:, 0] == 0] = 0.0
    return n.astype(np.float32)


def spherical_camera(rows: int = 32, cols: int = 512, el_deg: Tuple[float, float] = (-22.5, 22.5)) -> dict:
    """Spherical intrinsics that map a rows x cols LiDAR image onto [-pi, pi) x [el0, el1] (the layout of the reference's
    test_projections.cu:146-158): col = fx * azimuth + cx, row = fy * elevation + cy."""
    fx = cols / (2.0 * np.pi)
    el0, el1 = np.deg2rad(el_deg[0]), np.deg2rad(el_deg[1])
    fy = (rows - 1) / (el1 - el0)
    return dict(fx=fx, fy=fy, cx=cols / 2.0, cy=-fy * el0, rows=rows, cols=cols)


def spherical_range_image(scene: Scene, t: np.ndarray, q: np.ndarray, cam: dict) -> Tuple[np.ndarray, np.ndarray]:
    """Range image [rows, cols] float32 + colours for the spherical camera model (camera.cuh:91-99): pixel (row, col) looks
    along azimuth (col - cx - 0.5) / fx, elevation (row - cy - 0.5) / fy; the value is the range along that ray."""
    rows, cols = cam["rows"], cam["cols"]
    az = (np.arange(cols, dtype=np.float64) - cam["cx"] - 0.5) / cam["fx"]
    el = (np.arange(rows, dtype=np.float64) - cam["cy"] - 0.5) / cam["fy"]
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (rows, cols))], axis=-1)
    rng_len, pts = scene.cast_dirs(d, quat_to_rot(q), t)
    rng_len = np.where(np.isfinite(rng_len) & (rng_len > 0), rng_len, 0.0)
    return rng_len.astype(np.float32), scene.color(pts)


VBR_PARAMS = dict(  # mrhash/configurations/vbr.cfg
    sdf_truncation=0.40, sdf_truncation_scale=0.0, integration_weight_sample=1, virtual_voxel_size=0.20,
    n_frames_invalidate_voxels=0, voxel_extents_scale=1, marching_cubes_threshold=1.5, min_weight_threshold=50,
    min_depth=0.2, max_depth=100.0, sdf_var_threshold=0.0, vertices_merging_threshold=0.0, projective_sdf=True,
)


def drive_poses(n: int, step: float = 0.5) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Sensor moving along the street (+x), slight weave and yaw, z up (identity = x forward)."""
    out = []
    for i in range(n):
        x = -30.0 + step * i
        y = 0.8 * np.sin(0.07 * i)
        yaw = 0.05 * np.sin(0.11 * i)
        q = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)], dtype=np.float32)
        out.append((np.array([x, y, 0.0], dtype=np.float32), q))
    return out


def orbit_poses(n: int, radius: float = 1.0, yaw_step_deg: float = 1.8) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Camera on a circle in the x-z plane (makeCameraCircularTrajectory idea, tests/test_utils.cuh:20-32),
    looking through the room centre at the far walls (typical depths 2.5-4.5 m, as in Replica room0)."""
    out = []
    for i in range(n):
        a = np.deg2rad(yaw_step_deg) * i
        t = np.array([radius * np.sin(a), 0.0, radius * np.cos(a)], dtype=np.float32)
        out.append((t, yaw_quat(a + np.pi)))
    return out


def walk_poses(n: int, seed: int = 0, extent: Tuple[float, float] = (2.0, 1.5)) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Seeded smooth hand-held-like walk: slow drift in x-z, slowly varying yaw, small bob in y."""
    rng = np.random.default_rng(seed)
    out = []
    pos = np.zeros(3)
    yaw = 0.0
    vel = np.zeros(3)
    wyaw = 0.0
    for _ in range(n):
        vel = 0.9 * vel + rng.normal(0, 0.004, 3) * np.array([1, 0.2, 1])
        wyaw = 0.9 * wyaw + rng.normal(0, 0.004)
        pos = pos + vel
        pos[0] = np.clip(pos[0], -extent[0], extent[0])
        pos[2] = np.clip(pos[2], -extent[1], extent[1])
        pos[1] = np.clip(pos[1], -0.3, 0.3)
        yaw += wyaw + 0.01
        out.append((pos.astype(np.float32), yaw_quat(yaw)))
    return out


def replica_stream(n: int, K: Intrinsics = REPLICA_640, noise_sigma: float = 0.0) -> Iterator[Frame]:
    scene = replica_room()
    rng = np.random.default_rng(0)
    for t, q in orbit_poses(n):
        yield render(scene, K, t, q, depth_scaling=6553.5, noise_sigma=noise_sigma, rng=rng)


def scannet_stream(n: int, K: Intrinsics = SCANNET, start: int = 0) -> Iterator[Frame]:
    scene = scannet_room()
    poses = walk_poses(start + n, seed=0)[start:]
    for t, q in poses:
        yield render(scene, K, t, q, depth_scaling=5000.0)


REPLICA_PARAMS = dict(  # configurations/replica.cfg:2-19
    sdf_truncation=0.07, sdf_truncation_scale=0.0, integration_weight_sample=1, virtual_voxel_size=0.01,
    n_frames_invalidate_voxels=100, voxel_extents_scale=1, marching_cubes_threshold=1.5, min_weight_threshold=5,
    sdf_var_threshold=0.0, vertices_merging_threshold=0.0, min_depth=0.01, max_depth=30.0,
)
SCANNET_PARAMS = dict(REPLICA_PARAMS)  # configurations/scannet.cfg:2-19 (identical map/mesh sections)
CFG1_PARAMS = dict(
    sdf_truncation=0.06, sdf_truncation_scale=0.0, integration_weight_sample=1, virtual_voxel_size=0.02,
    n_frames_invalidate_voxels=0, voxel_extents_scale=1, marching_cubes_threshold=1.5, min_weight_threshold=5,
    sdf_var_threshold=0.0, vertices_merging_threshold=0.0, min_depth=0.01, max_depth=30.0,
)


def textured_image(rows: int, cols: int, seed: int = 0) -> np.ndarray:
    """uint8 [rows, cols, 3] with flat areas, soft gradients, sharp edges and a noisy patch: a colour quad-tree over
    it has leaves on every level (used for the 3DGS splat-seed cases)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:rows, 0:cols].astype(np.float64)
    img = np.zeros((rows, cols, 3), np.float64)
    img[..., 0] = 90 + 60 * np.sin(x / max(cols, 1) * 3.1) * np.cos(y / max(rows, 1) * 2.3)
    img[..., 1] = 120 + 0.08 * x
    img[..., 2] = 60 + 0.1 * y
    for _ in range(6):  # flat rectangles with hard edges
        r0, c0 = int(rng.integers(0, rows)), int(rng.integers(0, cols))
        h, w = int(rng.integers(1, max(rows // 3, 2))), int(rng.integers(1, max(cols // 3, 2)))
        img[r0:r0 + h, c0:c0 + w] = rng.integers(0, 256, 3)
    r0, c0 = rows // 5, cols // 2
    h, w = max(rows // 4, 1), max(cols // 4, 1)
    img[r0:r0 + h, c0:c0 + w] += rng.normal(0, 25, (min(h, rows - r0), min(w, cols - c0), 3))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
