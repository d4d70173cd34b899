#!/usr/bin/env python3
"""BASELINE.json configs[4] stand-in ("VBR"): 128 x 1024 LiDAR scans (131 072 points) along a 100 m street, vbr.cfg
parameters (voxel 0.20 m, truncation 0.40 m, integration distance 100 m).  Prints scans/s and points/s of
mrh_integrate_points with the scan resident in HBM, and the CPU restatement's time for the same scans."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mrhash_amd import capi, hipmem, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rows, cols = 128, 1024
hip = capi.load_hip()
scene = synth.street_canyon()
poses = synth.drive_poses(n, step=0.5)
scans = [synth.lidar_scan(scene, t, q, rows=rows, cols=cols) for t, q in poses]
d_scans = [hipmem.DeviceBuffer.from_numpy(np.ascontiguousarray(s, dtype=np.float32)) for s in scans]
e = capi.Engine(hip, capi.Params(num_sdf_blocks=262144, **synth.VBR_PARAMS))
e.set_camera(1, 1, 0, 0, 1, 1, 0.2, 100.0, model=1)
def run(lo, hi):
    for i in range(lo, hi):
        t, q = poses[i]
        e.set_pose(synth.quat_to_rot(q), t)
        e.set_points_device(d_scans[i].ptr, len(scans[i]))
        e.integrate_points()
run(0, 5); e.sync()
t0 = time.perf_counter(); run(5, n); e.sync(); dt = time.perf_counter() - t0
st = e.stats()
print(f"HIP: {1e6 * dt / (n - 5):9.1f} us/scan  {(n - 5) / dt:9.1f} scans/s  {(n - 5) * rows * cols / dt / 1e6:8.1f} Mpoints/s   blocks {st.occupied_fine}")
if "--no-cpu" not in sys.argv:
    import parity_utils as pu
    o = capi.Engine(pu.oracle_lib(), capi.Params(num_sdf_blocks=262144, **synth.VBR_PARAMS))
    o.set_camera(1, 1, 0, 0, 1, 1, 0.2, 100.0, model=1)
    tc = 0.0
    for i in range(3):
        t, q = poses[i]
        o.set_pose(synth.quat_to_rot(q), t); o.upload_points(scans[i])
        c0 = time.perf_counter(); o.integrate_points(); tc += time.perf_counter() - c0
    print(f"CPU restatement (sequential by definition, D6): {1e3 * tc / 3:8.1f} ms/scan  {3 / tc:6.2f} scans/s")
