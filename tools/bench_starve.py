#!/usr/bin/env python3
"""Starve frames of the 640x480 stream on their own: `period` given on the command line (every period-th frame starves), resident
frames, no synchronisation inside.  usage: tools/bench_starve.py [frames] [period]   (under tools/rocprof_cmd.sh for the kernel times)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mrhash_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
period = int(sys.argv[2]) if len(sys.argv) > 2 else 10
hip = capi.load_hip()
res = bench.Resident(bench.render_stream("replica", n), synth.REPLICA_640)
for per in (1 << 30, period):
    e = bench.make_engine(hip, capi.Params(num_sdf_blocks=262144, device_id=0, **synth.REPLICA_PARAMS), synth.REPLICA_640)
    res.run(e, 0, 10, integrate=lambda x: x.integrate(1 << 30))
    e.sync()
    t0 = time.perf_counter()
    res.run(e, 10, n, integrate=lambda x: x.integrate(per))
    e.sync()
    dt = time.perf_counter() - t0
    print(f"period {per}: {(n - 10) / dt:.0f} frames/s, {dt / (n - 10) * 1e6:.1f} us per frame, blocks {e.stats().occupied_fine}")
    e.close()
