"""3DGS splat seeds: oracle-side checks that run without a GPU.

The reference holds no test for CUDAQTree / processNodesKernel, so this part of the oracle is "parity unpinned"
against reference output; it is pinned here against an independent numpy restatement of computeError's summation
order (quad_tree.cu:6-90) and of the subdivision rule (:102-167), and through properties: the leaves tile the image,
a uniform image is one leaf, every seed sits in a voxel of weight 1 and carries the centre pixel's colour."""
import numpy as np
import pytest

import parity_utils as pu
from mrhash_amd import capi, synth

F32 = np.float32


def np_node_error(rgb: np.ndarray, x0: int, y0: int, w: int, h: int) -> np.float32:
    """computeError with numpy float32: thread t owns pixels t, t + 256, ... (row-major inside the node) and adds them
    sequentially (np.cumsum accumulates left to right in float32); the 256 partial sums fold by halving."""
    rows, cols, _ = rgb.shape
    node = rgb[y0:y0 + h, x0:x0 + w].reshape(-1, 3).astype(F32)
    count = w * h

    def fold(vals):  # vals: [count] float32 per-pixel terms
        s = np.zeros(256, F32)
        for t in range(min(256, count)):
            s[t] = np.cumsum(vals[t::256], dtype=F32)[-1]
        stride = 128
        while stride > 0:
            s[:stride] = s[:stride] + s[stride:2 * stride]
            stride //= 2
        return s[0]

    fin = []
    for k in range(3):
        mean = F32(fold(node[:, k]) / F32(count))
        d = node[:, k] - mean
        fin.append(F32(fold((d * d).astype(F32)) / F32(count)))
    err = F32(F32(F32(fin[0] * F32(0.2989)) + F32(fin[1] * F32(0.5870))) + F32(fin[2] * F32(0.1140)))
    return F32(F32(err * F32(cols * rows)) / F32(90000000.0))


def np_quadtree(rgb: np.ndarray, thr: float, min_px: int):
    rows, cols, _ = rgb.shape
    level, leaves = [(0, 0, cols, rows)], []
    while level:
        nxt = []
        for (x0, y0, w, h) in level:
            w1, h1 = w // 2, h // 2
            if np_node_error(rgb, x0, y0, w, h) <= F32(thr) or w1 <= min_px or h1 <= min_px:
                leaves.append((x0, y0, w, h))
                continue
            nxt += [(x0, y0, w1, h1), (x0, y0 + h1, w1, h - h1), (x0 + w1, y0, w - w1, h1), (x0 + w1, y0 + h1, w - w1, h - h1)]
        level = nxt
    return np.array(leaves, np.int32).reshape(-1, 4)


def _engine(lib, rows, cols, params=None, blocks=8192):
    K = synth.Intrinsics(0.9 * cols, 0.9 * cols, cols / 2.0, rows / 2.0, rows, cols)
    return pu.make_engine(lib, K, dict(synth.CFG1_PARAMS, **(params or {})), num_sdf_blocks=blocks), K


def _plane_frame(e, rows, cols, rgb, z=1.0, integrate=True):
    e.set_pose(np.eye(3, dtype=F32), np.zeros(3, F32))
    depth = np.full((rows, cols), z, F32)
    e.upload_depth(depth)
    e.upload_rgb(rgb)
    if integrate:
        assert not e.integrate()
    return depth


def _leaves_as_array(lv):
    return np.stack([lv["x0"], lv["y0"], lv["width"], lv["height"]], axis=1)


@pytest.mark.parametrize("rows,cols,thr,min_px", [(48, 64, 0.002, 1), (61, 97, 0.0005, 0), (33, 20, 0.001, 2), (8, 8, 0.0, 0)])
def test_quadtree_matches_numpy_restatement(oracle, rows, cols, thr, min_px):
    rgb = synth.textured_image(rows, cols, seed=rows)
    e, _ = _engine(oracle, rows, cols)
    _plane_frame(e, rows, cols, rgb)
    e.splat_seeds(thr, min_px)
    got = _leaves_as_array(e.qtree_leaves())
    want = np_quadtree(rgb, thr, min_px)
    assert np.array_equal(got, want)
    # the leaves tile the image: every pixel in exactly one leaf
    cover = np.zeros((rows, cols), np.int32)
    for x0, y0, w, h in got:
        cover[y0:y0 + h, x0:x0 + w] += 1
    assert (cover == 1).all()
    e.close()


def test_uniform_image_is_one_leaf_and_one_seed(oracle):
    rows, cols = 48, 64
    rgb = np.full((rows, cols, 3), 77, np.uint8)
    e, K = _engine(oracle, rows, cols)
    _plane_frame(e, rows, cols, rgb, z=1.0)
    seeds = e.splat_seeds(0.1, 1)
    lv = e.qtree_leaves()
    assert len(lv) == 1 and tuple(lv[0]) == (0, 0, cols, rows)
    # centre pixel (cols/2 + 0.5 -> 32, 24); back-projection of a plane at z = 1 under the identity pose
    assert len(seeds) == 1
    s = seeds[0]
    px, py = 32, 24
    want = np.array([(px - K.cx - 0.5) / K.fx, (py - K.cy - 0.5) / K.fy, 1.0])
    assert np.allclose(s["p"], want, atol=1e-6)
    assert s["scale"] == pytest.approx(np.sqrt((cols / 2) ** 2 + (rows / 2) ** 2) / K.fx, rel=1e-6)
    assert tuple(s["rgb"]) == (77, 77, 77)
    e.close()


def test_seeds_need_weight_one_and_valid_depth(oracle):
    rows, cols = 48, 64
    rgb = synth.textured_image(rows, cols, seed=3)
    e, _ = _engine(oracle, rows, cols)
    depth = _plane_frame(e, rows, cols, rgb)
    first = e.splat_seeds(0.002, 1)
    lv = e.qtree_leaves()
    assert 0 < len(first) <= len(lv)
    for s in first[:50]:  # every seed sits in a voxel that has been observed exactly once
        v, found = e.get_voxel(*[int(c) for c in _world_to_voxel(s["p"], 0.02)])
        assert found and v["weight"] == 1
    # the same frame again: weights become 2 -> no voxel qualifies
    assert not e.integrate()
    assert len(e.splat_seeds(0.002, 1)) == 0
    assert len(e.qtree_leaves()) == len(lv)
    # a hole in the depth image removes exactly the seeds whose centre pixel falls into it
    e2, _ = _engine(oracle, rows, cols)
    d2 = depth.copy()
    d2[:, : cols // 2] = 0.0
    e2.set_pose(np.eye(3, dtype=F32), np.zeros(3, F32))
    e2.upload_depth(d2)
    e2.upload_rgb(rgb)
    assert not e2.integrate()
    half = e2.splat_seeds(0.002, 1)
    lv2 = e2.qtree_leaves()
    centre_x = np.floor(lv2["x0"] + 0.5 * lv2["width"] + 0.5).astype(int)
    assert len(half) == int((centre_x >= cols // 2).sum() - (centre_x >= cols).sum())
    e.close()
    e2.close()


def _world_to_voxel(p, vs):
    # worldPointToVirtualVoxelPos (vhu.cuh:74-82): p / vs, then + sign(p) * 0.5, truncated
    q = np.asarray(p, F32) / F32(vs)
    return (q + np.sign(q).astype(F32) * F32(0.5)).astype(np.int32)


def test_argument_checks(oracle):
    rows, cols = 16, 16
    e, _ = _engine(oracle, rows, cols)
    with pytest.raises(capi.MrhError):
        e.splat_seeds(0.1, 1)  # no images yet
    _plane_frame(e, rows, cols, np.zeros((rows, cols, 3), np.uint8))
    with pytest.raises(capi.MrhError):
        e.splat_seeds(0.1, -1)
    with pytest.raises(capi.MrhError):
        e.splat_seeds(float("nan"), 1)
    e.upload_rgb(np.zeros((8, 8, 3), np.uint8))
    with pytest.raises(capi.MrhError):
        e.splat_seeds(0.1, 1)  # colour image shape differs from the camera
    e.close()
