#!/usr/bin/env python3
"""Tuning aid: where a k_mc workgroup's time goes, per block class (build with -DMRH_MC_TRACE, see tools/trace_mc.sh).
Shader-clock cycles of thread 0, summed over the workgroups of one extraction of the configs[2] map at the driver's workload."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrhash_amd import capi, hipmem, synth
capi.HIP_LIB_PATH = os.path.join(ROOT, "mrhash_amd", "csrc", "libmrhash_trace.so")
hip = capi.load_hip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
Kc = synth.REPLICA_640
scene = synth.replica_room()
frames = [synth.render(scene, Kc, t, q, depth_scaling=6553.5) for t, q in synth.orbit_poses(n)]
dd = hipmem.DeviceBuffer.from_numpy(np.stack([f.depth for f in frames]))
rr = hipmem.DeviceBuffer.from_numpy(np.stack([f.rgb for f in frames]))
for label, v in (("single-res", 0.0), ("multi-res", 0.005)):
    params = capi.Params(num_sdf_blocks=262144, **dict(synth.REPLICA_PARAMS, sdf_var_threshold=v))
    e = capi.Engine(hip, params)
    e.set_camera(Kc.fx, Kc.fy, Kc.cx, Kc.cy, Kc.rows, Kc.cols, params.min_depth, params.max_depth)
    for i, f in enumerate(frames):
        e.set_pose(f.R, f.t)
        e.set_depth_device(dd.ptr + i * Kc.rows * Kc.cols * 4, Kc.rows, Kc.cols)
        e.set_rgb_device(rr.ptr + i * Kc.rows * Kc.cols * 3, Kc.rows, Kc.cols)
        e.integrate()
    e.sync()
    e.extract_triangles(soup=False)
    buf = (C.c_uint32 * (2 * 65536 * 8))()
    hip.mrh_debug_mc_trace(None, 1)
    nt = e.extract_triangles(soup=False)
    hip.mrh_debug_mc_trace(buf, 1)
    a = np.frombuffer(buf, dtype=np.uint32).astype(np.int64).reshape(2, 65536, 8)
    st = e.stats()
    print(f"{label}: fine {st.occupied_fine} coarse {st.occupied_coarse} triangles {nt}")
    for p, pn in enumerate(("count", "emit")):
        for c, cn in enumerate(("fine, no coarse neighbour", "fine next to coarse", "coarse")):
            r = a[p][a[p][:, 0] == c + 1]
            if not len(r):
                continue
            stage, pre, known, lit, whole, nk, nl = (r[:, k].mean() for k in (1, 2, 3, 4, 5, 6, 7))
            print(f"  {pn:5s} {cn:26s} blocks {len(r):6d}  cand known {nk:6.1f} literal {nl:6.1f} | cycles per block: staging {stage:8.0f} prescreen {pre:8.0f} "
                  f"known {known:8.0f} literal {lit:8.0f} scan+tail {whole - stage - pre - known - lit:8.0f} | whole {whole:8.0f} median {np.median(r[:, 5]):8.0f}  (sum over blocks {r[:, 5].sum() / 1e6:8.1f} M)")
    e.close()
