"""Splat-seed timing on the replica stand-in stream (640x480): per-call wall time of mrh_splat_seeds after each frame."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mrhash_amd import capi, synth  # noqa: E402
from tests import parity_utils as pu  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hip = capi.load_hip()
K = synth.REPLICA_640
e = pu.make_engine(hip, K, synth.REPLICA_PARAMS, num_sdf_blocks=262144)
frames = list(synth.replica_stream(n))
ts, counts = [], []
for f in frames:
    pu.feed(e, f)
    e.sync()
    t0 = time.perf_counter()
    s = e.splat_seeds(0.1, 1)
    ts.append(time.perf_counter() - t0)
    counts.append((len(e.qtree_leaves()), len(s)))
ts = np.array(ts[2:]) * 1e6
print(f"splat_seeds: median {np.median(ts):.0f} us  min {ts.min():.0f} us  (blocking call incl. read-back of leaves and seeds)")
print("leaves/seeds per frame:", counts[:6])
