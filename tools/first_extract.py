import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from mrhash_amd import capi, synth
hip = capi.load_hip()
n = 25
res = bench.Resident(bench.render_stream("replica", n), synth.REPLICA_640)
for rep in range(2):
    p = capi.Params(num_sdf_blocks=262144, device_id=0, **dict(synth.REPLICA_PARAMS, sdf_var_threshold=0.005))
    e = bench.make_engine(hip, p, synth.REPLICA_640)
    res.run(e, 0, n); e.sync()
    for i in range(3):
        t0 = time.perf_counter(); nt = e.extract_triangles(soup=False); print(f"context {rep} extraction {i}: {1e3*(time.perf_counter()-t0):.2f} ms, {nt} triangles", flush=True)
    e.close()
