#!/usr/bin/env python3
"""BASELINE.json configs[2]: Replica stand-in 640x480, variance-adaptive multi-resolution fusion + marching-cubes extract.
Prints per-frame time of the multi-resolution path and the extraction time (triangle soup + CPU post-process)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrhash_amd import capi, hipmem, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
var = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
hip = capi.load_hip()
Kc = synth.REPLICA_640
scene = synth.replica_room()
frames = [synth.render(scene, Kc, t, q, depth_scaling=6553.5) for t, q in synth.orbit_poses(n)]
dd = hipmem.DeviceBuffer.from_numpy(np.stack([f.depth for f in frames]))
rr = hipmem.DeviceBuffer.from_numpy(np.stack([f.rgb for f in frames]))
for label, v in (("single-res", 0.0), ("multi-res", var)):
    params = capi.Params(num_sdf_blocks=262144, **dict(synth.REPLICA_PARAMS, sdf_var_threshold=v))
    e = capi.Engine(hip, params)
    e.set_camera(Kc.fx, Kc.fy, Kc.cx, Kc.cy, Kc.rows, Kc.cols, params.min_depth, params.max_depth)
    def run(lo, hi):
        for i in range(lo, hi):
            f = frames[i]
            e.set_pose(f.R, f.t)
            e.set_depth_device(dd.ptr + i * Kc.rows * Kc.cols * 4, Kc.rows, Kc.cols)
            e.set_rgb_device(rr.ptr + i * Kc.rows * Kc.cols * 3, Kc.rows, Kc.cols)
            e.integrate()
    run(0, 10); e.sync()
    t0 = time.perf_counter(); run(10, n); e.sync(); dt = time.perf_counter() - t0
    st = e.stats()
    print(f"{label}: {1e6 * dt / (n - 10):8.1f} us/frame ({(n - 10) / dt:8.1f} frames/s)  fine {st.occupied_fine} coarse {st.occupied_coarse}")
    t0 = time.perf_counter(); tris = e.extract_triangles(); t1 = time.perf_counter()
    V, F, C = e.extract_mesh(); t2 = time.perf_counter()
    print(f"{label}: extract_triangles {1e3 * (t1 - t0):7.2f} ms ({tris.shape[0]} triangles)   get V,F,C {1e3 * (t2 - t1):7.2f} ms  V {V.shape[0]} F {F.shape[0]}")
    t0 = time.perf_counter(); tris = e.extract_triangles(); t1 = time.perf_counter()
    print(f"{label}: extract_triangles again {1e3 * (t1 - t0):7.2f} ms")
    t0 = time.perf_counter(); nt = e.extract_triangles(soup=False); t1 = time.perf_counter()
    print(f"{label}: extract without the soup read-back (GeoWrapper.extractMesh) {1e3 * (t1 - t0):7.2f} ms ({nt} triangles)")
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e.extract_triangles(soup=False); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{label}: five more: " + " ".join(f"{t:.2f}" for t in ts) + " ms")
    e.set_profile(True); e.extract_triangles(soup=False)  # kernel times: event-carrying launches, kept out of the timed calls
    st = e.stats(); e.set_profile(False)
    mc_ms = st.last_mc_count_ms + st.last_mc_emit_ms
    alg = 6144.0 * st.occupied_fine + 768.0 * st.occupied_coarse + 72.0 * nt
    print(f"{label}: k_mc count {st.last_mc_count_ms:.3f} ms + emit {st.last_mc_emit_ms:.3f} ms over {st.last_mc_blocks} blocks; algorithmic bytes "
          f"{alg / 1e6:.1f} MB -> {alg / (mc_ms * 1e-3) / 1e9 if mc_ms > 0 else 0:.1f} GB/s")
    e.close()
