"""A SECOND restatement of the hot path, in numpy float32, written from the reference's source text and NOT from
oracle/mrh_oracle.c — so that a misreading shared by the C oracle and the HIP kernels (which were written side by side)
has a chance to show up.  Each function cites the reference lines it restates; paths are relative
to mrhash/src/sdf/ of rvp-group/mrhash.  Covered:

  allocate      allocBlocksKernel vds.cu:758-857 (every pixel ray, Amanatides-Woo over blocks) with
                worldPointToSDFBlock vhu.cuh:75-165, isSDFBlockInCameraFrustumApprox vds.cu:66-77, camera.cuh:84-203
  integrate     integrateDepthMapKernel vds.cu:1095-1181 + combineVoxel vhu.cuh:167-181 over the compact (in-frustum) blocks
  check_var / reintegrate_coarse   checkVarSDFKernel vds.cu:1857-1939 and reintegrateDepthMapKernel :1942-2018 as launched (:2085-2097)
  gc_decisions  garbageCollectIdentifyKernel vds.cu:1674-1713
  starve        starveVoxelsKernel vds.cu:1597-1649 (both passes)
  allocate3d / integrate3d   allocBlocks3DKernel vds.cu:925-1033, integrate3DKernel :1215-1379 (projective SDF; python loop per ray)
  mc_block      extractIsoSurfaceAtPosition marching_cubes.cu:72-261 + trilinearInterpolation vds.cu:260-338 + getVoxel
                vds.cu:163-205 + vertexInterp mesh_extractor.cu:6-36, voxel by voxel (pure-python loop: small cases only)

Arithmetic: IEEE binary32 throughout (numpy float32 scalars / arrays; no FMA), `normalize` as v * (1 / sqrt(dot)) — the one
place the shared arithmetic spec departs from the reference's rsqrtf.  Test infrastructure only.
"""
from __future__ import annotations

import numpy as np

F = np.float32
I = np.int32
FLT_MAX = np.finfo(np.float32).max
P0, P1, P2 = 73856093, 19349669, 83492791
BLOCK = 8


def _sign(x):  # cuda_math.cuh:62-64
    return (x > 0).astype(F) - (x < 0).astype(F)


def _trunc_int(x):  # C float -> int conversion (finite inputs)
    return np.trunc(x).astype(I)


class Camera:
    """camera.cuh:13-40 (pinhole)."""

    def __init__(self, fx, fy, cx, cy, rows, cols, min_depth, max_depth):
        self.fx, self.fy, self.cx, self.cy = F(fx), F(fy), F(cx), F(cy)
        self.ifx, self.ify = F(1.0) / F(fx), F(1.0) / F(fy)
        self.rows, self.cols = int(rows), int(cols)
        self.row_thr, self.col_thr = int(rows * 0.5), int(cols * 0.5)  # camera.cuh:27-28
        self.min_depth, self.max_depth = F(min_depth), F(max_depth)
        self.R = np.eye(3, dtype=F)
        self.t = np.zeros(3, dtype=F)

    def set_pose(self, R, t):
        self.R, self.t = np.asarray(R, F).reshape(3, 3), np.asarray(t, F).reshape(3)

    # cuda_algebra.cuh:71-75, 146-148: rotation * point (sum left to right) + translation
    @staticmethod
    def _apply(R, t, p):
        x = R[0, 0] * p[..., 0] + R[0, 1] * p[..., 1] + R[0, 2] * p[..., 2]
        y = R[1, 0] * p[..., 0] + R[1, 1] * p[..., 1] + R[1, 2] * p[..., 2]
        z = R[2, 0] * p[..., 0] + R[2, 1] * p[..., 1] + R[2, 2] * p[..., 2]
        return np.stack([x + t[0], y + t[1], z + t[2]], -1).astype(F)

    def cam_in_world(self, p):
        return self._apply(self.R, self.t, p)

    def world_in_cam(self, p):  # cuda_algebra.cuh:137-143: (R^T, -(R^T t))
        Ri = self.R.T.copy()
        ti = -(Ri[:, 0] * self.t[0] + Ri[:, 1] * self.t[1] + Ri[:, 2] * self.t[2]).astype(F)
        return self._apply(Ri, ti, p)

    def inverse_projection(self, row, col, d):  # camera.cuh:88
        x = self.ifx * (col.astype(F) - self.cx - F(0.5))
        y = self.ify * (row.astype(F) - self.cy - F(0.5))
        return np.stack([d * x, d * y, d * F(1.0)], -1).astype(F)

    def project(self, pc, approx):  # camera.cuh:131-147 / :167-182
        z = pc[..., 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            row = _trunc_int(np.nan_to_num((self.fy * pc[..., 1] / z + self.cy) + F(0.5), nan=-1e9, posinf=1e9, neginf=-1e9))
            col = _trunc_int(np.nan_to_num((self.fx * pc[..., 0] / z + self.cx) + F(0.5), nan=-1e9, posinf=1e9, neginf=-1e9))
        ok = ~((z <= self.min_depth) | (z > self.max_depth))
        if approx:
            ok &= (row >= -self.row_thr) & (col >= -self.col_thr) & (row < self.rows + self.row_thr) & (col < self.cols + self.col_thr)
        else:
            ok &= (row >= 0) & (col >= 0) & (row < self.rows) & (col < self.cols)
        return ok, row, col


def voxel_to_world(vs, v):  # vhu.cuh:66-68
    return (v.astype(F) * F(vs)).astype(F)


def world_to_voxel(vs, p):  # vhu.cuh:143-151
    q = (p / F(vs)).astype(F)
    a = q + _sign(q) * F(0.5)
    eps = F(1e-5)
    a = np.where(a >= 0, np.floor(a + eps), np.ceil(a - eps))
    return _trunc_int(a)


def voxel_to_block(v, vs):  # vhu.cuh:75-103, voxel_extents == 1
    v = v.astype(I).copy()
    v = np.where(v < 0, v - (BLOCK - 1), v)
    pw = voxel_to_world(vs, v)
    mbs = F(F(1.0) * F(BLOCK) * F(vs))
    eps = F(1e-5)
    b = np.where(pw >= 0, np.floor((pw + eps) / mbs), np.ceil((pw - eps) / mbs))
    return _trunc_int(b)


def world_to_block(vs, p):
    return voxel_to_block(world_to_voxel(vs, p), vs)


VERT_OFFSET = np.array([[0, 0, 0], [0, 0, 7], [0, 7, 0], [0, 7, 7], [7, 0, 0], [7, 0, 7], [7, 7, 0], [7, 7, 7]], I)  # params.h:41-49


def block_in_frustum(cam: Camera, vs, blocks):  # vds.cu:66-77
    blocks = np.asarray(blocks, I).reshape(-1, 3)
    ok = np.zeros(len(blocks), bool)
    for off in VERT_OFFSET:
        pw = voxel_to_world(vs, blocks * BLOCK + off)
        good, _, _ = cam.project(cam.world_in_cam(pw), approx=True)
        ok |= good
    return ok


def cloud_z(cam: Camera, depth):  # camera.cu:13-18: the cloud stays 0 outside (min_depth, max_depth]
    d = np.asarray(depth, F)
    return np.where((d <= cam.min_depth) | (d > cam.max_depth), F(0), d).astype(F)


def allocate(cam: Camera, params: dict, depth) -> set:
    """Block positions allocBlocksKernel inserts for this frame (vds.cu:758-857), as a set of (x, y, z)."""
    vs, trunc, scale = F(params["virtual_voxel_size"]), F(params["sdf_truncation"]), F(params["sdf_truncation_scale"])
    dmax_int = cam.max_depth  # geowrapper.cpp:111
    d = cloud_z(cam, depth)
    rows, cols = np.nonzero(d != 0)
    d = d[rows, cols]
    t = trunc + scale * d
    lo, hi = np.minimum(dmax_int, d - t), np.minimum(dmax_int, d + t)
    keep = ~(lo >= hi)
    rows, cols, lo, hi = rows[keep], cols[keep], lo[keep], hi[keep]
    pw_min = cam.cam_in_world(cam.inverse_projection(rows, cols, lo))
    pw_max = cam.cam_in_world(cam.inverse_projection(rows, cols, hi))
    dd = (pw_max - pw_min).astype(F)
    inv = F(1.0) / np.sqrt(dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1] + dd[:, 2] * dd[:, 2])
    direc = (dd * inv[:, None]).astype(F)
    cur = world_to_block(vs, pw_min)
    end = world_to_block(vs, pw_max)
    step = _sign(direc)
    nb = cur + _trunc_int(np.clip(step, F(0.0), F(1.0)))
    boundary = voxel_to_world(vs, nb * BLOCK) - F(0.5) * vs
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        t_max = ((boundary - pw_min) / direc).astype(F)
        t_delta = ((step * F(BLOCK) * vs) / direc).astype(F)
    bound = _trunc_int(end.astype(F) + step)
    guard = (np.abs(direc) < F(1e-6)) | (np.abs(boundary - direc) < F(1e-6))  # vds.cu:801-827, literally
    t_max = np.where(guard, FLT_MAX, t_max).astype(F)
    t_delta = np.where(guard, FLT_MAX, t_delta).astype(F)
    alive = np.ones(len(cur), bool)
    visited = set()
    for _ in range(1024):
        if not alive.any():
            break
        idx = np.nonzero(alive)[0]
        uniq = np.unique(cur[idx], axis=0)
        for b in uniq[block_in_frustum(cam, vs, uniq)]:
            visited.add((int(b[0]), int(b[1]), int(b[2])))
        tm = t_max[idx]
        ax = (tm[:, 0] < tm[:, 1]) & (tm[:, 0] < tm[:, 2])
        az = ~ax & (tm[:, 2] < tm[:, 1])
        axis = np.where(ax, 0, np.where(az, 2, 1))
        cur[idx, axis] = _trunc_int(cur[idx, axis].astype(F) + step[idx, axis])
        done = cur[idx, axis] == bound[idx, axis]
        with np.errstate(over="ignore"):
            t_max[idx, axis] = np.where(done, t_max[idx, axis], (t_max[idx, axis] + t_delta[idx, axis]).astype(F))
        alive[idx[done]] = False
    return visited


def integrate(cam: Camera, params: dict, depth, rgb, blocks: dict) -> dict:
    """integrateDepthMapKernel (vds.cu:1095-1181) + combineVoxel (vhu.cuh:167-181) over every block of `blocks`
    ({(x, y, z): structured voxel array [512]}) that passes the compaction predicate (vds.cu:406-434).  Fine blocks only."""
    from mrhash_amd import capi

    vs, trunc, scale = F(params["virtual_voxel_size"]), F(params["sdf_truncation"]), F(params["sdf_truncation_scale"])
    w1 = np.uint8(params["integration_weight_sample"])
    wmax = int(params.get("integration_weight_max", 255)) & 0xFF
    keys = sorted(blocks)
    pos = np.array(keys, I).reshape(-1, 3)
    vis = block_in_frustum(cam, vs, pos)
    d_img = cloud_z(cam, depth)
    out = {k: blocks[k].copy() for k in keys}
    if not vis.any():
        return out
    pos = pos[vis]
    lin = np.arange(512)
    local = np.stack([lin % 8, (lin % 64) // 8, lin // 64], -1).astype(I)  # delinearizeVoxelPos, vhu.cuh:130-136
    pi = pos[:, None, :] * BLOCK + local[None, :, :]
    pf = voxel_to_world(vs, pi)
    pc = cam.world_in_cam(pf)
    ok, row, col = cam.project(pc, approx=False)
    row, col = np.where(ok, row, 0), np.where(ok, col, 0)
    d = d_img[row, col]
    ok &= ~((d == 0) | (d > cam.max_depth))
    sdf = (d - pc[..., 2]).astype(F)
    t = (trunc + scale * d).astype(F)
    ok &= ~(sdf <= -t)
    sdf = np.where(sdf >= 0, np.minimum(t, sdf), np.maximum(-t, sdf)).astype(F)
    old = np.stack([blocks[tuple(int(c) for c in p)] for p in pos]).view(capi.VOXEL_DTYPE).reshape(len(pos), 512)
    s0, w0, c0 = old["sdf"], old["weight"], old["rgb"]
    c1 = np.asarray(rgb, np.uint8)[row, col]
    curr_mean = np.where(w0 > 0, s0, sdf).astype(F)
    half = F(vs / F(2))
    delta = ((sdf - curr_mean) / half).astype(F)
    c0 = np.where((w0 == 0)[..., None], c1, c0)
    res = (F(0.5) * c0.astype(F) + F(0.5) * c1.astype(F)).astype(F)
    rgb_new = _trunc_int(res + F(0.5)).astype(np.uint8)
    wsum = w0.astype(np.int32) + int(w1)  # uchar + uchar promotes to int
    with np.errstate(invalid="ignore", divide="ignore"):
        s_new = ((s0 * w0.astype(F) + sdf * F(w1)) / wsum.astype(F)).astype(F)
    w_new = np.minimum(wmax, wsum).astype(np.uint8)
    delta2 = ((sdf - s_new) / half).astype(F)
    ss_new = (F(0) + delta * delta2).astype(F)  # whole-voxel store of a default Voxel, then atomicAdd (vds.cu:1178-1180)
    new = old.copy()
    new["sdf"] = np.where(ok, s_new, s0)
    new["sum_squared"] = np.where(ok, ss_new, old["sum_squared"])
    new["weight"] = np.where(ok, w_new, w0)
    new["rgb"] = np.where(ok[..., None], rgb_new, old["rgb"])
    for i, p in enumerate(pos):
        out[tuple(int(c) for c in p)] = new[i]
    return out


# ---- variance-adaptive resolution, garbage collection, starve (numpy; blocks = {(x, y, z): voxel array [512]}) -------------

def compact_positions(cam: Camera, params: dict, blocks: dict):
    """flatAndReduceHashTable(camera) (vds.cu:406-434): the blocks whose approx-frustum test passes, here in ascending
    (x, y, z) order — the reference's compact order is a race (atomicAdd winners), the build's canonical order C1."""
    vs = F(params["virtual_voxel_size"])
    keys = sorted(blocks)
    if not keys:
        return []
    vis = block_in_frustum(cam, vs, np.array(keys, I).reshape(-1, 3))
    return [k for k, v in zip(keys, vis) if v]


def check_var(cam: Camera, params: dict, blocks: dict, all_blocks: bool = False):
    """checkVarSDFKernel (vds.cu:1857-1939) over the compact FINE blocks: the positions the kernel hands to reallocBlocks.
    64 threads per block; thread t sums the 2x2x2 cell at (2 (t % 4), 2 ((t / 4) % 4), 2 (t / 16)) in dz, dy, dx order over the
    voxels with weight > 0 (sum_squared and weight, both as float), then the shared-memory tree `s[t] += s[t + stride]` for
    stride = 32 .. 1; thread 0: weight sum >= 2, var = sum / (weight - 1), coarsen if weight - 1 > 1e-6, var > 0 and
    var < sdf_var_threshold."""
    thr = F(params["sdf_var_threshold"])
    out = []
    t = np.arange(64)
    gx, gy, gz = (t % 4) * 2, ((t // 4) % 4) * 2, (t // 16) * 2
    for key in (sorted(blocks) if all_blocks else compact_positions(cam, params, blocks)):  # flatAndReduceHashTable() without a camera lists every block
        vox = blocks[key]
        ss = np.zeros(64, F)
        ww = np.zeros(64, F)
        for dz in range(2):
            for dy in range(2):
                for dx in range(2):
                    li = (gz + dz) * 64 + (gy + dy) * 8 + (gx + dx)
                    w = vox["weight"][li]
                    on = w > 0
                    ss = np.where(on, (ss + vox["sum_squared"][li]).astype(F), ss)
                    ww = np.where(on, (ww + w.astype(F)).astype(F), ww)
        stride = 32
        while stride > 0:
            ss[:stride] = (ss[:stride] + ss[stride:2 * stride]).astype(F)
            ww[:stride] = (ww[:stride] + ww[stride:2 * stride]).astype(F)
            stride //= 2
        if ww[0] < 2:
            continue
        with np.errstate(invalid="ignore", divide="ignore"):
            var = float(F(ss[0] / F(ww[0] - F(1))))  # float division, then widened to double (vds.cu:1919)
        if F(ww[0] - F(1)) > F(1e-6) and var > 0.0 and var < float(thr):
            out.append(key)
    return out


def reintegrate_coarse(cam: Camera, params: dict, depth, rgb, key):
    """The 64-voxel payload of a block that checkVarSDF just re-allocated at resolution 1 (zeroed by reallocBlock), after
    reintegrateDepthMapKernel (vds.cu:1942-2018) as LAUNCHED (:2085-2097): the kernel is started with `n_threads` where a dim3
    was meant, so blockDim.y is 1, voxel_idx == blockIdx.y and only voxel indices 0 .. 512 / n_threads - 1 = 0 .. 31 run
    (n_threads = 16, params.h:15).  Voxel i of a coarse block sits at pos * 8 + 2 * delinearize(i, 4)."""
    from mrhash_amd import capi

    vs, trunc, scale = F(params["virtual_voxel_size"]), F(params["sdf_truncation"]), F(params["sdf_truncation_scale"])
    w1 = int(params["integration_weight_sample"]) & 0xFF
    wmax = int(params.get("integration_weight_max", 255)) & 0xFF
    out = np.zeros(64, capi.VOXEL_DTYPE)
    idx = np.arange(32)
    local = 2 * np.stack([idx % 4, (idx % 16) // 4, idx // 16], -1).astype(I)
    pi = np.array(key, I)[None, :] * BLOCK + local
    pc = cam.world_in_cam(voxel_to_world(vs, pi))
    ok, row, col = cam.project(pc, approx=False)
    row, col = np.where(ok, row, 0), np.where(ok, col, 0)
    d = cloud_z(cam, depth)[row, col]
    ok &= ~((d == 0) | (d > cam.max_depth))
    sdf = (d - pc[:, 2]).astype(F)
    t = (trunc + scale * d).astype(F)
    ok &= ~(sdf <= -t)
    sdf = np.where(sdf >= 0, np.minimum(t, sdf), np.maximum(-t, sdf)).astype(F)
    c1 = np.asarray(rgb, np.uint8)[row, col]
    # the voxel is empty (weight 0): colour := new colour, then combineVoxel (vhu.cuh:167-181) with w0 = 0, s0 = 0
    col_new = _trunc_int(F(0.5) * c1.astype(F) + F(0.5) * c1.astype(F) + F(0.5)).astype(np.uint8)
    with np.errstate(invalid="ignore", divide="ignore"):
        s_new = ((F(0) * F(0) + sdf * F(w1)) / F(w1)).astype(F)
    out["sdf"][:32] = np.where(ok, s_new, F(0))
    out["weight"][:32] = np.where(ok, min(wmax, w1), 0)
    out["rgb"][:32] = np.where(ok[:, None], col_new, 0)
    return out  # sum_squared stays 0: the merged voxel is default-constructed and combineVoxel never sets it


def gc_decisions(cam: Camera, params: dict, blocks: dict):
    """garbageCollectIdentifyKernel (vds.cu:1674-1713) over the compact fine blocks: positions whose decision is 1 —
    min over the weighted voxels of |sdf| >= truncation(max depth), or no weighted voxel at all."""
    thr = F(F(params["sdf_truncation"]) + F(params["sdf_truncation_scale"]) * cam.max_depth)
    out = []
    for key in compact_positions(cam, params, blocks):
        vox = blocks[key]
        a = np.where(vox["weight"] == 0, FLT_MAX, np.abs(vox["sdf"])).astype(F)
        if a.min() >= thr or vox["weight"].max() == 0:
            out.append(key)
    return out


def starve(cam: Camera, params: dict, blocks: dict):
    """starveVoxelsKernel twice (vds.cu:1597-1649, host :1652-1671) over the compact fine blocks: per pixel the voxel with
    the smallest (depth bits, thread id) among ALL voxels that project into it (weight is not looked at; depth >= min depth)
    loses one unit of weight.  Thread id = compact index * 512 + voxel index with the canonical compact order (C1)."""
    vs = F(params["virtual_voxel_size"])
    keys = compact_positions(cam, params, blocks)
    out = {k: v.copy() for k, v in blocks.items()}
    if not keys:
        return out
    pos = np.array(keys, I).reshape(-1, 3)
    lin = np.arange(512)
    local = np.stack([lin % 8, (lin % 64) // 8, lin // 64], -1).astype(I)
    pc = cam.world_in_cam(voxel_to_world(vs, pos[:, None, :] * BLOCK + local[None, :, :]))
    z = pc[..., 2]
    ok, row, col = cam.project(pc, approx=False)
    ok &= ~(z < cam.min_depth)
    tid = (np.arange(len(keys))[:, None] * 512 + lin[None, :]).astype(np.uint64)
    packed = (z.view(np.uint32).astype(np.uint64) << np.uint64(32)) + tid
    pix = row.astype(np.int64) * cam.cols + col
    sel = np.nonzero(ok)
    order = np.lexsort((packed[sel], pix[sel]))
    first = np.ones(len(order), bool)
    first[1:] = pix[sel][order][1:] != pix[sel][order][:-1]
    win_b, win_v = sel[0][order][first], sel[1][order][first]
    for b, v in zip(win_b, win_v):
        w = out[keys[b]]["weight"]
        w[v] = max(0, int(w[v]) - 1)
    return out


# ---- LiDAR scans: allocBlocks3DKernel vds.cu:925-1033, integrate3DKernel vds.cu:1215-1379 (projective SDF, fine blocks) -----

def _norm3(p):  # norm3df, restated as sqrtf((x^2 + y^2) + z^2)
    return np.sqrt((p[..., 0] * p[..., 0] + p[..., 1] * p[..., 1]) + p[..., 2] * p[..., 2]).astype(F)


def _normalize(p):  # cuda_math.cuh:1075-1078 with rsqrtf as 1 / sqrtf
    inv = F(1.0) / np.sqrt(p[..., 0] * p[..., 0] + p[..., 1] * p[..., 1] + p[..., 2] * p[..., 2])
    return (p * inv[..., None]).astype(F)


def _dda_setup(vs, pw_min, pw_max, cell: int):
    """Common head of the two DDA kernels: first / bound cell, step, t_max, t_delta for cells of `cell` voxels."""
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        direc = _normalize((pw_max - pw_min).astype(F))
        if cell == BLOCK:
            cur, end = world_to_block(vs, pw_min), world_to_block(vs, pw_max)
        else:
            cur, end = world_to_voxel(vs, pw_min), world_to_voxel(vs, pw_max)
        step = _sign(direc)
        nb = cur + _trunc_int(np.clip(step, F(0.0), F(1.0)))
        boundary = voxel_to_world(vs, nb * cell) - F(0.5) * vs
        t_max = ((boundary - pw_min) / direc).astype(F)
        t_delta = ((step * F(cell) * vs) / direc).astype(F) if cell == BLOCK else ((step * vs) / direc).astype(F)
        bound = _trunc_int(end.astype(F) + step)
        guard = (np.abs(direc) < F(1e-6)) | (np.abs(boundary - direc) < F(1e-6))
    return cur, bound, step, np.where(guard, FLT_MAX, t_max).astype(F), np.where(guard, FLT_MAX, t_delta).astype(F)


def _dda_walk(cur, bound, step, t_max, t_delta, visit):
    """The traversal loop both kernels share (vds.cu:1009-1031 / :1356-1377) for ONE ray; visit(cell) returns False to stop."""
    cur, t_max = [int(c) for c in cur], [F(v) for v in t_max]
    for _ in range(1024):
        if visit(tuple(cur)) is False:
            return
        if t_max[0] < t_max[1] and t_max[0] < t_max[2]:
            a = 0
        elif t_max[2] < t_max[1]:
            a = 2
        else:
            a = 1
        cur[a] = int(np.trunc(F(F(cur[a]) + step[a])))
        if cur[a] == bound[a]:
            return
        with np.errstate(over="ignore"):
            t_max[a] = F(t_max[a] + t_delta[a])


def allocate3d(cam: Camera, params: dict, points, normals=None) -> set:
    """Blocks allocBlocks3DKernel inserts for one scan (points in the sensor frame, (0, 0, 0) = no return).  No frustum
    test on this path.  `normals` (one per point): the segment runs along the normal instead of the beam (vds.cu:957-962)."""
    vs, trunc, scale = F(params["virtual_voxel_size"]), F(params["sdf_truncation"]), F(params["sdf_truncation_scale"])
    p = np.asarray(points, F).reshape(-1, 3)
    nrm = None if normals is None else np.asarray(normals, F).reshape(-1, 3)
    rng = _norm3(p)
    keep = rng != 0
    p, rng = p[keep], rng[keep]
    nrm = None if nrm is None else nrm[keep]
    t = (trunc + scale * rng).astype(F)
    lo, hi = np.minimum(cam.max_depth, rng - t), np.minimum(cam.max_depth, rng + t)
    keep = ~(lo >= hi)
    p, rng, lo, hi = p[keep], rng[keep], lo[keep], hi[keep]
    with np.errstate(divide="ignore", invalid="ignore"):
        cdir = _normalize(p) if nrm is None else _normalize(nrm[keep])
    pw_min = cam.cam_in_world((p + cdir * (lo - rng)[:, None]).astype(F))
    pw_max = cam.cam_in_world((p + cdir * (hi - rng)[:, None]).astype(F))
    cur, bound, step, t_max, t_delta = _dda_setup(vs, pw_min, pw_max, BLOCK)
    visited = set()
    for i in range(len(p)):
        _dda_walk(cur[i], bound[i], step[i], t_max[i], t_delta[i], lambda c: visited.add(c))
    return visited


def integrate3d(cam: Camera, params: dict, points, blocks: dict, normals=None) -> dict:
    """integrate3DKernel for one scan, the points taken in ascending index (the build's canonical order D6 for the
    reference's racing read-modify-writes).  Fine blocks; projective SDF, or with `normals` the normal-direction one
    (segment pcam + n (min_depth - range) .. pcam + n (max_depth - range), sdf = dot(voxel - point, n): vds.cu:1248-1251, :1322-1326)."""
    vs, trunc, scale = F(params["virtual_voxel_size"]), F(params["sdf_truncation"]), F(params["sdf_truncation_scale"])
    w1 = int(params["integration_weight_sample"]) & 0xFF
    wmax = int(params.get("integration_weight_max", 255)) & 0xFF
    half = F(vs / F(2))
    out = {k: v.copy() for k, v in blocks.items()}
    p = np.asarray(points, F).reshape(-1, 3)
    nrm = None if normals is None else np.asarray(normals, F).reshape(-1, 3)
    rng = _norm3(p)
    keep = ~((rng.astype(np.float64) < 1e-6) | (rng > cam.max_depth))  # `range < 1e-6` compares in double (vds.cu:1233)
    p, rng = p[keep], rng[keep]
    nrm = None if nrm is None else nrm[keep]
    t = (trunc + scale * rng).astype(F)
    lo, hi = np.minimum(cam.max_depth, rng - t), np.minimum(cam.max_depth, rng + t)
    keep = ~(lo >= hi)
    p, rng, t, lo, hi = p[keep], rng[keep], t[keep], lo[keep], hi[keep]
    if nrm is None:
        cdir = _normalize(p)
        pw_min = cam.cam_in_world((p - cdir * t[:, None]).astype(F))
        pw_max = cam.cam_in_world((p + cdir * t[:, None]).astype(F))
        ndir = np.zeros_like(p)
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            ndir = _normalize(nrm[keep])
        pw_min = cam.cam_in_world((p + ndir * (lo - rng)[:, None]).astype(F))
        pw_max = cam.cam_in_world((p + ndir * (hi - rng)[:, None]).astype(F))
    cur, bound, step, t_max, t_delta = _dda_setup(vs, pw_min, pw_max, 1)
    for i in range(len(p)):
        r_i, t_i, p_i, n_i = rng[i], t[i], p[i], ndir[i]

        def visit(v, r_i=r_i, t_i=t_i, p_i=p_i, n_i=n_i):
            b = tuple(int(c) for c in voxel_to_block(np.array(v, I), vs))
            blk = out.get(b)
            if blk is None:
                return True
            pc = cam.world_in_cam(voxel_to_world(vs, np.array(v, I)))
            if nrm is None:
                sdf = F(r_i - _norm3(pc))
            else:  # dot(voxel_pos_camera - pcam, norm_dir), cuda_math dot: x*x' + y*y' + z*z' left to right
                dv = (pc - p_i).astype(F)
                sdf = F(F(F(dv[0] * n_i[0]) + F(dv[1] * n_i[1])) + F(dv[2] * n_i[2]))
            if sdf <= -t_i:
                return False  # `break`: the ray is done
            sdf = min(t_i, sdf) if sdf >= 0 else max(F(-t_i), sdf)
            li = (v[2] % 8) * 64 + (v[1] % 8) * 8 + (v[0] % 8)
            s0, w0 = F(blk["sdf"][li]), int(blk["weight"][li])
            mean = s0 if w0 > 0 else F(0)  # vds.cu:1343-1348: 0, not the sample, for an empty voxel
            delta = F(F(sdf - mean) / half)
            c0 = blk["rgb"][li].astype(F)
            blk["rgb"][li] = _trunc_int(F(0.5) * c0 + F(0.5) * F(0) + F(0.5)).astype(np.uint8)  # the sample carries colour 0
            s_new = F(F(F(s0 * F(w0)) + F(sdf * F(w1))) / F(w0 + w1))
            blk["sdf"][li] = s_new
            blk["weight"][li] = min(wmax, w0 + w1)
            blk["sum_squared"][li] = F(F(0) + F(delta * F(F(sdf - s_new) / half)))
            return True

        _dda_walk(cur[i], bound[i], step[i], t_max[i], t_delta[i], visit)
    return out


def integrate3d_multires(cam: Camera, params: dict, points, blocks: dict) -> dict:
    """integrate3DKernel (projective SDF) on a map with fine AND coarse blocks, blocks = {(x, y, z): (resolution, voxels)}:
    on a coarse block the SDF is measured at the fine voxel coordinate divided by 2 with C's truncation toward zero, times the
    coarse voxel size (vds.cu:1303-1309), and the voxel updated is the one the coarse local coordinate names (dense index:
    deviation D1).  Points in ascending index (D6)."""
    vs, trunc, scale = F(params["virtual_voxel_size"]), F(params["sdf_truncation"]), F(params["sdf_truncation_scale"])
    w1 = int(params["integration_weight_sample"]) & 0xFF
    wmax = int(params.get("integration_weight_max", 255)) & 0xFF
    half = F(vs / F(2))
    out = {k: (r, v.copy()) for k, (r, v) in blocks.items()}
    p = np.asarray(points, F).reshape(-1, 3)
    rng = _norm3(p)
    keep = ~((rng.astype(np.float64) < 1e-6) | (rng > cam.max_depth))
    p, rng = p[keep], rng[keep]
    t = (trunc + scale * rng).astype(F)
    lo, hi = np.minimum(cam.max_depth, rng - t), np.minimum(cam.max_depth, rng + t)
    keep = ~(lo >= hi)
    p, rng, t = p[keep], rng[keep], t[keep]
    cdir = _normalize(p)
    pw_min = cam.cam_in_world((p - cdir * t[:, None]).astype(F))
    pw_max = cam.cam_in_world((p + cdir * t[:, None]).astype(F))
    cur, bound, step, t_max, t_delta = _dda_setup(vs, pw_min, pw_max, 1)
    for i in range(len(p)):
        r_i, t_i = rng[i], t[i]

        def visit(v, r_i=r_i, t_i=t_i):
            b = tuple(int(c) for c in voxel_to_block(np.array(v, I), vs))
            ent = out.get(b)
            if ent is None:
                return True
            res, blk = ent
            sc = 1 << res
            aprox = np.array([int(v[0] / sc), int(v[1] / sc), int(v[2] / sc)], I)  # C division: toward zero
            pc = cam.world_in_cam(voxel_to_world(F(vs * F(sc)), aprox))
            sdf = F(r_i - _norm3(pc))
            if sdf <= -t_i:
                return False
            sdf = min(t_i, sdf) if sdf >= 0 else max(F(-t_i), sdf)
            lx, ly, lz = v[0] % 8, v[1] % 8, v[2] % 8
            li = lz * 64 + ly * 8 + lx if res == 0 else (lz // 2) * 16 + (ly // 2) * 4 + (lx // 2)
            s0, w0 = F(blk["sdf"][li]), int(blk["weight"][li])
            mean = s0 if w0 > 0 else F(0)
            delta = F(F(sdf - mean) / half)
            c0 = blk["rgb"][li].astype(F)
            blk["rgb"][li] = _trunc_int(F(0.5) * c0 + F(0.5) * F(0) + F(0.5)).astype(np.uint8)
            s_new = F(F(F(s0 * F(w0)) + F(sdf * F(w1))) / F(w0 + w1))
            blk["sdf"][li] = s_new
            blk["weight"][li] = min(wmax, w0 + w1)
            blk["sum_squared"][li] = F(F(0) + F(delta * F(F(sdf - s_new) / half)))
            return True

        _dda_walk(cur[i], bound[i], step[i], t_max[i], t_delta[i], visit)
    return out


# ---- marching cubes, one voxel at a time (pure python over numpy float32 scalars) ------------------------------------

class Map:
    def __init__(self, params: dict, blocks: dict):
        self.vs = F(params["virtual_voxel_size"])
        self.min_w = int(params["min_weight_threshold"])
        self.thr = F(params["marching_cubes_threshold"])
        self.blocks = blocks

    def _w2v(self, p):
        return tuple(int(c) for c in world_to_voxel(self.vs, np.array(p, F)))

    def get_voxel(self, p):  # vds.cu:163-205 via worldPointToVirtualVoxelPos; single resolution
        v = self._w2v(p)
        b = tuple(int(c) for c in voxel_to_block(np.array(v, I), self.vs))
        blk = self.blocks.get(b)
        if blk is None:
            return F(0), 0, (0, 0, 0)
        li = (v[2] % 8) * 64 + (v[1] % 8) * 8 + (v[0] % 8)  # python % is non-negative: vhu.cuh:110-128 with block_size 8
        e = blk[li]
        return F(e["sdf"]), int(e["weight"]), tuple(int(c) for c in e["rgb"])

    def trilinear(self, pos):  # vds.cu:260-338, single resolution (voxel size == vs everywhere)
        h = self.vs
        dual = [F(pos[a] - h * F(0.5)) for a in range(3)]
        x0, y0, z0 = dual
        x1, y1, z1 = x0, y0, z0
        sdf = [F(0)] * 8
        for i in range(8):
            dx, dy, dz = i & 1, (i >> 1) & 1, (i >> 2) & 1
            vp = (F(dual[0] + F(dx) * h), F(dual[1] + F(dy) * h), F(dual[2] + F(dz) * h))
            s, w, _ = self.get_voxel(vp)
            if not w:
                return False, F(0)
            sdf[i] = s
            x1, y1, z1 = max(x1, vp[0]), max(y1, vp[1]), max(z1, vp[2])
        ddx = F((pos[0] - x0) / (x1 - x0)) if (x1 - x0) > F(1e-6) else F(0.5)
        ddy = F((pos[1] - y0) / (y1 - y0)) if (y1 - y0) > F(1e-6) else F(0.5)
        ddz = F((pos[2] - z0) / (z1 - z0)) if (z1 - z0) > F(1e-6) else F(0.5)
        c = [sdf[0], F(sdf[1] - sdf[0]), F(sdf[2] - sdf[0]), F(sdf[4] - sdf[0]),
             F(F(F(sdf[3] - sdf[2]) - sdf[1]) + sdf[0]), F(F(F(sdf[6] - sdf[4]) - sdf[2]) + sdf[0]), F(F(F(sdf[5] - sdf[4]) - sdf[1]) + sdf[0]),
             F(F(F(F(F(F(F(sdf[7] - sdf[6]) - sdf[5]) - sdf[3]) + sdf[1]) + sdf[4]) + sdf[2]) - sdf[0])]
        dist = F(c[0] + F(c[1] * ddx))
        dist = F(dist + F(c[2] * ddy))
        dist = F(dist + F(c[3] * ddz))
        dist = F(dist + F(F(c[4] * ddx) * ddy))
        dist = F(dist + F(F(c[5] * ddy) * ddz))
        dist = F(dist + F(F(c[6] * ddx) * ddz))
        dist = F(dist + F(F(F(c[7] * ddx) * ddy) * ddz))
        return True, dist


class MultiMap(Map):
    """A map with fine AND coarse blocks: blocks = {(x, y, z): (resolution, voxels)} with 512 voxels at resolution 0 and 64
    (dense, z * 16 + y * 4 + x of the half coordinates) at resolution 1 — coarse blocks are read with the index their
    writers use, the build's deviation D1 from the reference's stride-8 read (vhu.cuh:110-128)."""

    jumps = 0   # trilinear samples that took the coarser re-sample (vds.cu:296-309)
    shrunk = 0  # corner offsets shrunk by checkVertexVoxels

    def _entry(self, b):  # getHashEntry: a miss answers resolution 0 (vds.cu:124-126)
        e = self.blocks.get(b)
        return (0, None) if e is None else e

    def voxel_size_at(self, p):  # getVoxelSize(float3) vds.cu:236-240
        b = tuple(int(c) for c in world_to_block(self.vs, np.array(p, F)))
        return F(self.vs * F(1 << self._entry(b)[0]))

    def get_voxel_res(self, p):  # getVoxel(float3, block_res) vds.cu:176-205
        v = self._w2v(p)
        b = tuple(int(c) for c in voxel_to_block(np.array(v, I), self.vs))
        res, vox = self._entry(b)
        if vox is None:
            return F(0), 0, (0, 0, 0), 0  # block_res is left untouched (0 after the caller's reset)
        lx, ly, lz = v[0] % 8, v[1] % 8, v[2] % 8
        li = lz * 64 + ly * 8 + lx if res == 0 else (lz // 2) * 16 + (ly // 2) * 4 + (lx // 2)
        e = vox[li]
        return F(e["sdf"]), int(e["weight"]), tuple(int(c) for c in e["rgb"]), res

    def get_voxel(self, p):
        return self.get_voxel_res(p)[:3]

    def trilinear(self, pos):  # vds.cu:260-338
        h = self.voxel_size_at(pos)
        dual = [F(pos[a] - F(h * F(0.5))) for a in range(3)]
        # the base resolution is looked up with the LOCAL voxel size in place of the finest one (vds.cu:264), literally
        bb = tuple(int(c) for c in voxel_to_block(world_to_voxel(h, np.array(pos, F)), h))
        base_res = self._entry(bb)[0]
        pos_sdf = self.get_voxel(dual)[0]
        x0, y0, z0 = dual
        x1, y1, z1 = x0, y0, z0
        sdf = [F(0)] * 8
        for i in range(8):
            dx, dy, dz = i & 1, (i >> 1) & 1, (i >> 2) & 1
            vp = (F(dual[0] + F(F(dx) * h)), F(dual[1] + F(F(dy) * h)), F(dual[2] + F(F(dz) * h)))
            s, w, _, res = self.get_voxel_res(vp)
            if not w:
                return False, F(0)
            if res > base_res:
                self.jumps += 1
                nh = F(h * F(2))
                npos = tuple(F(F(pos[a] - F(nh * F(0.5))) + F(F((dx, dy, dz)[a]) * nh)) for a in range(3))
                ns = self.get_voxel(npos)[0]
                sdf[i] = F(F(F(0.5) * pos_sdf) + F(F(0.5) * ns))
            else:
                sdf[i] = s
            x1, y1, z1 = max(x1, vp[0]), max(y1, vp[1]), max(z1, vp[2])
        ddx = F((pos[0] - x0) / (x1 - x0)) if (x1 - x0) > F(1e-6) else F(0.5)
        ddy = F((pos[1] - y0) / (y1 - y0)) if (y1 - y0) > F(1e-6) else F(0.5)
        ddz = F((pos[2] - z0) / (z1 - z0)) if (z1 - z0) > F(1e-6) else F(0.5)
        c = [sdf[0], F(sdf[1] - sdf[0]), F(sdf[2] - sdf[0]), F(sdf[4] - sdf[0]),
             F(F(F(sdf[3] - sdf[2]) - sdf[1]) + sdf[0]), F(F(F(sdf[6] - sdf[4]) - sdf[2]) + sdf[0]), F(F(F(sdf[5] - sdf[4]) - sdf[1]) + sdf[0]),
             F(F(F(F(F(F(F(sdf[7] - sdf[6]) - sdf[5]) - sdf[3]) + sdf[1]) + sdf[4]) + sdf[2]) - sdf[0])]
        dist = F(c[0] + F(c[1] * ddx))
        dist = F(dist + F(c[2] * ddy))
        dist = F(dist + F(c[3] * ddz))
        dist = F(dist + F(F(c[4] * ddx) * ddy))
        dist = F(dist + F(F(c[5] * ddy) * ddz))
        dist = F(dist + F(F(c[6] * ddx) * ddz))
        dist = F(dist + F(F(F(c[7] * ddx) * ddy) * ddz))
        return True, dist

    def corner_offsets(self, pf):
        """extractIsoSurfaceAtPosition's P / M (marching_cubes.cu:77-82) after checkVertexVoxels (:7-69): the half-voxel offsets
        per axis and side, shrunk by 0.499 where the neighbouring position answers another voxel size."""
        h = self.voxel_size_at(pf)
        Pp = F(h * F(0.5))
        sP, sM = [F(Pp * F(1)) for _ in range(3)], [F(F(-Pp) * F(1)) for _ in range(3)]
        for a in range(3):
            for side in (sP, sM):
                q = [F(pf[0] + F(0)), F(pf[1] + F(0)), F(pf[2] + F(0))]
                q[a] = F(pf[a] + side[a])
                vs_q = self.voxel_size_at(q)
                if vs_q > 0 and vs_q < 1 and vs_q != h:
                    side[a] = F(side[a] * F(0.499))
                    self.shrunk += 1
        return sP, sM


def vertex_interp(p1, p2, d1, d2, c1, c2):  # mesh_extractor.cu:6-36
    iso = F(0)
    if abs(F(iso - d1)) < F(0.00001):
        return p1, tuple(F(F(c) / F(255.0)) for c in c1)
    if abs(F(iso - d2)) < F(0.00001):
        return p2, tuple(F(F(c) / F(255.0)) for c in c2)
    if abs(F(d1 - d2)) < F(0.00001):
        return p1, tuple(F(F(c) / F(255.0)) for c in c1)
    mu = F(F(iso - d1) / F(d2 - d1))
    p = tuple(F(p1[a] + F(mu * F(p2[a] - p1[a]))) for a in range(3))
    col = tuple(F(F(c1[a]) + F(F(mu * F(c2[a] - c1[a])) / F(255.0))) for a in range(3))  # int difference, then float (vhu Vertex colour)
    return p, col


def mc_voxel(m: Map, tri_table, pf):
    """Triangles ([(p, c)] x 3 each) of one voxel, marching_cubes.cu:72-261 (a MultiMap adds checkVertexVoxels and the local
    voxel size)."""
    if isinstance(m, MultiMap):
        sP, sM = m.corner_offsets(pf)
    else:
        Pp = F(m.vs * F(0.5))
        sP, sM = [Pp] * 3, [F(-Pp)] * 3
    ps, dist, cols = [], [], []
    for k in range(8):  # corner k: bit 0 = +x, bit 1 = +y, bit 2 = +z (p000, p001 = +x, p010 = +y, ...)
        p = (F(pf[0] + (sP[0] if k & 1 else sM[0])), F(pf[1] + (sP[1] if k & 2 else sM[1])), F(pf[2] + (sP[2] if k & 4 else sM[2])))
        valid, dk = m.trilinear(p)
        s, w, c = m.get_voxel(p)
        if not valid:
            if w < m.min_w:
                return []
            dk = s
        ps.append(p); dist.append(F(dk)); cols.append(c)
    cube = sum(1 << k for k in range(8) if dist[k] < F(0))
    for a in dist:
        for b in dist:
            if F(a * b) < F(0):
                if F(abs(a) + abs(b)) > m.thr:
                    return []
            elif abs(F(a - b)) > m.thr:
                return []
    if any(abs(a) > m.thr for a in dist):
        return []
    row = tri_table[cube]
    tris = []
    for j in range(row[0]):
        tri = []
        for k in range(3):
            code = row[1 + 3 * j + k]
            a, b = code >> 4, code & 0xF
            tri.append(vertex_interp(ps[a], ps[b], dist[a], dist[b], cols[a], cols[b]))
        tris.append(tri)
    return tris


def reference_tri_table(reference_root: str):
    """[256][16] = {triangle count, 15 edge codes}, built from the reference's own Transvoxel tables (params.h:138-435):
    regularCellClass, regularCellData (vertex / triangle counts + vertex indices), regularVertexData (low byte = the two
    corner numbers of the edge).  Parsed from the source text at test time; falls back to include/mrh_mc_tables.h."""
    import os
    import re

    path = os.path.join(reference_root, "mrhash", "src", "sdf", "params.h")
    if not os.path.exists(path):
        return None
    txt = open(path).read()

    def array_body(name):
        i = txt.index(name)
        i = txt.index("{", txt.index("=", i))
        depth, j = 0, i
        while True:
            if txt[j] == "{":
                depth += 1
            elif txt[j] == "}":
                depth -= 1
                if depth == 0:
                    return txt[i + 1: j]
            j += 1

    nums = lambda s: [int(x, 0) for x in re.findall(r"0x[0-9A-Fa-f]+|\d+", re.sub(r"//[^\n]*", "", s))]  # noqa: E731
    cls = nums(array_body("regularCellClass"))
    assert len(cls) == 256
    data_rows = re.findall(r"\{\s*(0x[0-9A-Fa-f]+)\s*,\s*\{([^}]*)\}\s*\}", array_body("regularCellData"))
    assert len(data_rows) == 16
    cell = [(int(g, 16), nums(v)) for g, v in data_rows]
    vrows = re.findall(r"\{([^{}]*)\}", array_body("regularVertexData"))
    assert len(vrows) == 256
    vdata = [nums(r) for r in vrows]
    table = []
    for cube in range(256):
        geo, vidx = cell[cls[cube]]
        ntri = geo & 0x0F  # RegularCellData::getTriangleCount
        row = [ntri] + [0] * 15
        for s in range(3 * ntri):
            row[1 + s] = vdata[cube][vidx[s]] & 0xFF
        table.append(row)
    return table
