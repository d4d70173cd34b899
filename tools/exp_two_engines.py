#!/usr/bin/env python3
"""How well does the chip co-schedule the two launches of a frame when they come from DIFFERENT streams?  N independent
contexts (own stream each) fuse the same resident frames, fed round-robin from one host thread: aggregate frames/s of N
engines against one.  No kernel changes — an upper-bound probe for overlapping k_front(f + 1) with k_back(f).
usage: python tools/exp_two_engines.py [frames=60]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mrhash_amd import capi, hipmem, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W = 10
hip = capi.load_hip()
hipmem.set_device(0)
frames = bench.render_stream("replica", W + n)
res = bench.Resident(frames, synth.REPLICA_640)
for N in (1, 2, 3, 1):
    engs = [bench.make_engine(hip, capi.Params(num_sdf_blocks=131072, device_id=0, **synth.REPLICA_PARAMS), synth.REPLICA_640) for _ in range(N)]
    for e in engs:
        res.run(e, 0, W)
    for e in engs:
        e.sync()
    hipmem.synchronize()
    t0 = time.perf_counter()
    for i in range(W, W + n):
        for e in engs:
            res.run(e, i, i + 1)
    for e in engs:
        e.sync()
    dt = time.perf_counter() - t0
    # host-only cost of the same loop (enqueue without waiting is what it already is; report per-call time)
    print(f"{N} engine(s): {N * n / dt:9.0f} frames/s aggregate, {dt / n * 1e6:7.1f} us per round of {N} frame(s)", flush=True)
    for e in engs:
        e.close()
