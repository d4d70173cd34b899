#!/usr/bin/env python3
"""Repeats the upload-fed pipelined 640x480 run of tests/test_parity_gpu.py::test_replica_640x480_stream many times against ONE oracle
map and reports every run whose occupancy differs (which blocks, and the library's counters).  usage: tools/flake_hunt.py [runs] [frames]"""
import os, sys
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
for r in range(runs):
    a = pu.make_engine(hip, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
    for f in frames:
        pu.feed(a, f)
    a.sync()
    da, va = a.dump_blocks()
    got = set(map(tuple, np.stack([da["x"], da["y"], da["z"]], 1).tolist()))
    if got != want or not np.array_equal(va["weight"], vb["weight"]):
        bad += 1
        st = a.stats()
        print(f"run {r}: {len(got)} blocks against {len(want)}; missing {sorted(want - got)[:6]} extra {sorted(got - want)[:6]}; flags {st.error_flags} free {st.free_fine}", flush=True)
    a.close()
print(f"flake hunt: {bad} of {runs} runs differ")
