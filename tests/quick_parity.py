#!/usr/bin/env python3
"""GPU-box smoke: HIP vs oracle on cfg1 and a few 640x480 frames. Prints a summary."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mrhash_amd import capi, synth
import parity_utils as pu

if os.environ.get("MRH_LIB"):  # a tuning build of the library (mrhash_amd/build.py variant ...)
    capi.HIP_LIB_PATH = os.path.abspath(os.environ["MRH_LIB"])
hip = capi.load_hip(); orc = pu.oracle_lib()
print(hip.mrh_version(), orc.mrh_version())

def run(name, K, params, frames, nb=65536, mesh=True, **extra):
    a = pu.make_engine(hip, K, params, nb, **extra); b = pu.make_engine(orc, K, params, nb, **extra)
    t0 = time.time()
    try:
        for f in frames:
            pu.feed(a, f); pu.feed(b, f)
        a.sync()
        r = pu.compare_maps(a, b)
        print(name, "map OK", r, "t=%.2fs" % (time.time() - t0), flush=True)
        if mesh:
            t0 = time.time(); r = pu.compare_meshes(a, b); print(name, "mesh OK", r, "t=%.2fs" % (time.time() - t0), flush=True)
    except AssertionError as ex:
        print(name, "PARITY FAIL:", ex, flush=True)
    sa, sb = a.stats(), b.stats()
    print(name, "stats hip", sa.occupied_fine, sa.occupied_coarse, sa.free_fine, sa.free_coarse, "orc", sb.occupied_fine, sb.occupied_coarse, sb.free_fine, sb.free_coarse, flush=True)
    a.close(); b.close()

which = sys.argv[1:] or ["plane", "sphere", "replica", "starve", "multires"]
if "plane" in which: run("cfg1-plane", synth.CFG1, synth.CFG1_PARAMS, [synth.cfg1_plane()])
if "sphere" in which: run("cfg1-sphere", synth.CFG1, synth.CFG1_PARAMS, [synth.cfg1_sphere()] * 3)
if "replica" in which:
    fr = list(synth.replica_stream(int(os.environ.get("MRH_QP_FRAMES", "3"))))
    run("replica-640", synth.REPLICA_640, synth.REPLICA_PARAMS, fr, nb=131072, mesh=False)
if "starve" in which:
    p = dict(synth.CFG1_PARAMS); p["n_frames_invalidate_voxels"] = 2
    run("cfg1-starve", synth.CFG1, p, [synth.cfg1_sphere()] * 5)
if "multires" in which:
    p = dict(synth.CFG1_PARAMS); p["sdf_var_threshold"] = 0.5; p["n_frames_invalidate_voxels"] = 3
    run("cfg1-multires", synth.CFG1, p, [synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.52), synth.cfg1_sphere(zc=1.49), synth.cfg1_sphere()], nb=16384)
print("QUICK PARITY DONE")
