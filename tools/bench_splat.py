"""Splat-seed timing on the replica stand-in stream (640x480): per-call wall time of mrh_splat_seeds after each frame."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mrhash_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hip = capi.load_hip()
K = synth.REPLICA_640
_p = capi.Params(num_sdf_blocks=262144, **synth.REPLICA_PARAMS)
e = capi.Engine(hip, _p)
e.set_camera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, _p.min_depth, _p.max_depth)
frames = list(synth.replica_stream(n))
ts, counts = [], []
for f in frames:
    e.set_pose(f.R, f.t)
    e.upload_depth(f.depth)
    e.upload_rgb(f.rgb)
    e.integrate()
    e.sync()
    t0 = time.perf_counter()
    s = e.splat_seeds(0.1, 1)
    ts.append(time.perf_counter() - t0)
    counts.append((len(e.qtree_leaves()), len(s)))
ts = np.array(ts[2:]) * 1e6
print(f"splat_seeds: median {np.median(ts):.0f} us  min {ts.min():.0f} us  (blocking call incl. read-back of leaves and seeds)")
print("leaves/seeds per frame:", counts[:6])
