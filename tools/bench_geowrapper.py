"""Frames/s through the drop-in surface itself: GeoWrapper.setCurrPose / setDepthImage / setRGBImage / compute() with
host numpy inputs (what apps/rgbd_runner.py does), 640x480 Replica stand-in."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
os.environ.setdefault("MRHASH_NUM_SDF_BLOCKS", "262144")
from mrhash.src.pygeowrapper import GeoWrapper  # noqa: E402
from mrhash_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
K = synth.REPLICA_640
frames = list(synth.replica_stream(n))
g = GeoWrapper(sdf_truncation=0.07, sdf_truncation_scale=0.0, integration_weight_sample=1, virtual_voxel_size=0.01,
               n_frames_invalidate_voxels=100, voxel_extents_scale=1, viewer_active=False, marching_cubes_threshold=1.5,
               min_weight_threshold=5, min_depth=0.01, max_depth=30.0)
g.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 30.0, 0)


def run(lo, hi):
    parts = np.zeros(4)
    for f in frames[lo:hi]:
        t0 = time.perf_counter(); g.setCurrPose(f.t, f.q)
        t1 = time.perf_counter(); g.setDepthImage(f.depth)
        t2 = time.perf_counter(); g.setRGBImage(f.rgb)
        t3 = time.perf_counter(); g.compute()
        t4 = time.perf_counter()
        parts += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
    return parts


run(0, 20)
g.streamAllOut()
t0 = time.perf_counter()
parts = run(20, n)
g.streamAllOut()
dt = time.perf_counter() - t0
m = n - 20
print(f"GeoWrapper: {m / dt:.0f} frames/s ({1e6 * dt / m:.1f} us/frame); setCurrPose {1e6 * parts[0] / m:.1f}  setDepthImage {1e6 * parts[1] / m:.1f}"
      f"  setRGBImage {1e6 * parts[2] / m:.1f}  compute {1e6 * parts[3] / m:.1f} us")
t0 = time.perf_counter()
g.extractMesh("/tmp/mrh_bench_mesh.ply")
dt = time.perf_counter() - t0
print(f"GeoWrapper.extractMesh: {1e3 * dt:.1f} ms for {len(g.getFaces())} faces ({os.path.getsize('/tmp/mrh_bench_mesh.ply') / 1e6:.1f} MB ASCII PLY)")
