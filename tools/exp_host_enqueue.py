#!/usr/bin/env python3
"""Host enqueue time per frame vs wall time per frame, pipelined and serial (MRH_PIPE set by the caller)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mrhash_amd import capi, hipmem, synth
n, W = int(sys.argv[1]) if len(sys.argv) > 1 else 40, 8
hip = capi.load_hip(); hipmem.set_device(0)
frames = bench.render_stream("replica", W + n)
res = bench.Resident(frames, synth.REPLICA_640)
for rep in range(3):
    e = bench.make_engine(hip, capi.Params(num_sdf_blocks=262144, device_id=0, **synth.REPLICA_PARAMS), synth.REPLICA_640)
    res.run(e, 0, W); e.sync(); hipmem.synchronize()
    t0 = time.perf_counter()
    res.run(e, W, W + n)
    t1 = time.perf_counter()
    e.sync()
    t2 = time.perf_counter()
    print(f"MRH_PIPE={os.environ.get('MRH_PIPE','1')} period={os.environ.get('MRH_PIPE_PERIOD','-')}: enqueue {(t1 - t0) / n * 1e6:6.1f} us/frame, wall {(t2 - t0) / n * 1e6:6.1f} us/frame", flush=True)
    e.close()
