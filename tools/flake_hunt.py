#!/usr/bin/env python3
"""Repeats the upload-fed pipelined 640x480 run of tests/test_parity_gpu.py::test_replica_640x480_stream many times against ONE oracle
map and reports every run whose occupancy differs (which blocks, and the library's counters).  usage: tools/flake_hunt.py [runs] [frames]
MRH_FH_JITTER=<us>: a random pause of up to that many microseconds between the calls of a frame (a host that is sometimes ahead of the
device and sometimes behind it: the event queries of the pipeline then take both branches); MRH_FH_SECONDS=<s>: stop after that long."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MRH_PIPE_UPLOADS", "1")
import numpy as np
import parity_utils as pu
from mrhash_amd import capi, synth

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hip, orc = capi.load_hip(), pu.oracle_lib()
frames = list(synth.replica_stream(nf))
b = pu.make_engine(orc, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
for f in frames:
    pu.feed(b, f)
db, vb = b.dump_blocks()
want = set(map(tuple, np.stack([db["x"], db["y"], db["z"]], 1).tolist()))
bad = 0
jit = float(os.environ.get("MRH_FH_JITTER", "0")) * 1e-6
limit = float(os.environ.get("MRH_FH_SECONDS", "0"))
rng = random.Random(1)
t_start = time.time()


def pause():
    if jit > 0:
        t_end = time.perf_counter() + rng.uniform(0, jit)
        while time.perf_counter() < t_end:
            pass


done = 0
for r in range(runs):
    if limit and time.time() - t_start > limit:
        break
    a = pu.make_engine(hip, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
    for f in frames:
        if jit > 0:
            a.set_pose(f.R, f.t); pause()
            a.upload_depth(f.depth); pause()
            a.upload_rgb(f.rgb); pause()
            assert not a.integrate(-1); pause()
        else:
            pu.feed(a, f)
    a.sync()
    done += 1
    da, va = a.dump_blocks()
    got = set(map(tuple, np.stack([da["x"], da["y"], da["z"]], 1).tolist()))
    if got != want or not np.array_equal(va["weight"], vb["weight"]):
        bad += 1
        st = a.stats()
        print(f"run {r}: {len(got)} blocks against {len(want)}; missing {sorted(want - got)[:6]} extra {sorted(got - want)[:6]}; flags {st.error_flags} free {st.free_fine}", flush=True)
    a.close()
print(f"flake hunt: {bad} of {done} runs differ ({time.time() - t_start:.0f} s, jitter {jit * 1e6:.0f} us)")
