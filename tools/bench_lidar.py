"""LiDAR scans alone (bench.py's configs[4] leg): us per scan over a few repetitions and a digest of the final map, for A/B runs
of library switches (MRH_LIDAR_*).  usage: python tools/bench_lidar.py [reps]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mrhash_amd import capi  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if os.environ.get("BENCH_LIB"):  # another build of the library (A/B on one box)
    capi.HIP_LIB_PATH = os.path.abspath(os.environ["BENCH_LIB"])
hip = capi.load_hip()
le, scans, d_scans, run_scans = bench.lidar_setup(hip, 262144)
n, w = bench.LIDAR_SCANS, bench.LIDAR_WARMUP
if os.environ.get("HOST_CLOUDS"):  # the drop-in hand-over: host arrays through mrh_upload_points (looked at: mrh_detect_scan_layout); HOST_CLOUDS=-1: not looked at
    from mrhash_amd import synth
    le.set_scan_layout(0 if os.environ["HOST_CLOUDS"] != "-1" else -1)
    poses = synth.drive_poses(n, step=0.5)

    def run_scans(lo, hi):  # noqa: F811
        for i in range(lo, hi):
            t, q = poses[i]
            le.set_pose(synth.quat_to_rot(q), t)
            le.upload_points(scans[i])
            le.integrate_points()
times = []
digest = None
for r in range(reps):
    le.reset()
    run_scans(0, w)
    le.sync()
    t0 = time.perf_counter()
    run_scans(w, n)
    le.sync()
    times.append((time.perf_counter() - t0) / (n - w) * 1e6)
    if r == 0:
        d, v = le.dump_blocks()
        digest = hashlib.sha256(d.tobytes() + v.tobytes()).hexdigest()[:16]
print("lidar us_per_scan", " ".join(f"{t:.1f}" for t in times), "blocks", len(d), "map", digest, "env", {k: v for k, v in os.environ.items() if k.startswith("MRH_")})
le.close()
