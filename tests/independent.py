"""A SECOND restatement of the hot path, in numpy float32, written from the reference's source text and NOT from
oracle/mrh_oracle.c — so that a misreading shared by the C oracle and the HIP kernels (which were written side by side)
has a chance to show up.  Each function cites the reference lines it restates; paths are relative
to mrhash/src/sdf/ of rvp-group/mrhash.  Covered:

  allocate      allocBlocksKernel vds.cu:758-857 (every pixel ray, Amanatides-Woo over blocks) with
                worldPointToSDFBlock vhu.cuh:75-165, isSDFBlockInCameraFrustumApprox vds.cu:66-77, camera.cuh:84-203
  integrate     integrateDepthMapKernel vds.cu:1095-1181 + combineVoxel vhu.cuh:167-181 over the compact (in-frustum) blocks
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
    """Triangles ([(p, c)] x 3 each) of one voxel, marching_cubes.cu:72-261 on a single-resolution map."""
    Pp = F(m.vs * F(0.5))
    Mm = F(-Pp)
    ps, dist, cols = [], [], []
    for k in range(8):  # corner k: bit 0 = +x, bit 1 = +y, bit 2 = +z (p000, p001 = +x, p010 = +y, ...)
        p = (F(pf[0] + (Pp if k & 1 else Mm)), F(pf[1] + (Pp if k & 2 else Mm)), F(pf[2] + (Pp if k & 4 else Mm)))
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
