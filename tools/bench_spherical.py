"""The spherical-image leg of bench.py on its own (128 x 1024 range images of the street scene, vbr.cfg parameters), for
rocprofv3 --kernel-trace --stats and A/B runs.  usage: python tools/bench_spherical.py [frames] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrhash_amd import capi, synth, hipmem

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w_img = 3
hip = capi.load_hip()
cam = synth.spherical_camera(128, 1024)
sposes = synth.drive_poses(n_img, step=0.5)
scene = synth.street_canyon()
imgs = [synth.spherical_range_image(scene, t, q, cam) for t, q in sposes]
dd = hipmem.DeviceBuffer.from_numpy(np.stack([a for a, _ in imgs]).astype(np.float32))
dc = hipmem.DeviceBuffer.from_numpy(np.stack([b for _, b in imgs]).astype(np.uint8))
sp = dict(synth.VBR_PARAMS, n_frames_invalidate_voxels=100)
e = capi.Engine(hip, capi.Params(num_sdf_blocks=262144, device_id=0, **sp))
e.set_camera(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["rows"], cam["cols"], sp["min_depth"], 100.0, model=1)
npx = cam["rows"] * cam["cols"]


def run(lo, hi):
    for i in range(lo, hi):
        t, q = sposes[i]
        e.set_pose(synth.quat_to_rot(q), t)
        e.set_depth_device(dd.ptr + i * npx * 4, cam["rows"], cam["cols"])
        e.set_rgb_device(dc.ptr + i * npx * 3, cam["rows"], cam["cols"])
        e.integrate()


out = []
for r in range(reps):
    e.reset()
    run(0, w_img); e.sync()
    c0 = time.perf_counter()
    run(w_img, n_img); e.sync()
    out.append((time.perf_counter() - c0) / (n_img - w_img) * 1e6)
print("spherical us_per_frame", " ".join(f"{t:.1f}" for t in out), "blocks", int(e.stats().occupied_fine))
