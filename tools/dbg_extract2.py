import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from mrhash_amd import capi, hipmem, synth
mode = sys.argv[1]
hip = capi.load_hip()
Kc = synth.REPLICA_640
scene = synth.replica_room()
n = 25
frames = [synth.render(scene, Kc, t, q, depth_scaling=6553.5) for t, q in synth.orbit_poses(n)]
dd = hipmem.DeviceBuffer.from_numpy(np.stack([f.depth for f in frames]))
rr = hipmem.DeviceBuffer.from_numpy(np.stack([f.rgb for f in frames]))
def build(var, upload=False, nblocks=262144):
    params = capi.Params(num_sdf_blocks=nblocks, **dict(synth.REPLICA_PARAMS, sdf_var_threshold=var))
    e = capi.Engine(hip, params)
    e.set_camera(Kc.fx, Kc.fy, Kc.cx, Kc.cy, Kc.rows, Kc.cols, params.min_depth, params.max_depth)
    for i, f in enumerate(frames):
        e.set_pose(f.R, f.t)
        if upload:
            e.upload_depth(f.depth); e.upload_rgb(f.rgb)
        else:
            e.set_depth_device(dd.ptr + i * Kc.rows * Kc.cols * 4, Kc.rows, Kc.cols)
            e.set_rgb_device(rr.ptr + i * Kc.rows * Kc.cols * 3, Kc.rows, Kc.cols)
        e.integrate()
    e.sync()
    return e
if mode == "dummy_first":
    d = build(0.0); d.close()
if mode == "dummy_extract_first":
    d = build(0.0); d.extract_triangles(soup=False); d.extract_triangles(soup=False); d.close()
if mode == "upload_first":
    d = build(0.0, upload=True); d.close()
if mode == "profile_first":
    d = build(0.0); d.set_profile(True); d.extract_triangles(soup=False); d.set_profile(False); d.close()
e = build(0.005)
for k in range(6):
    if k == 5: os.environ["MRH_DEBUG"] = "1"
    t0 = time.perf_counter(); nt = e.extract_triangles(soup=False); print(f"{mode} call {k}: {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr)
