"""Where the host-input path (numpy -> mrh_upload_depth / mrh_upload_rgb -> mrh_integrate per frame) spends its time on the
host: each call of the frame loop timed separately over a few hundred frames.  Run with MRH_COPY_THREADS=<n> to vary the
staging pool.

    python tools/host_path_breakdown.py [--frames 300]
"""
import argparse
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    args = ap.parse_args()
    from mrhash_amd import capi, hipmem, synth

    hipmem.set_device(0)
    hip = capi.load_hip()
    K, P = synth.REPLICA_640, synth.REPLICA_PARAMS
    frames = bench.render_stream("replica", 40)
    params = capi.Params(num_sdf_blocks=262144, device_id=0, **P)
    e = bench.make_engine(hip, params, K)
    gc.collect()
    gc.disable()
    n = args.frames
    acc = {"set_pose": 0.0, "upload_depth": 0.0, "upload_rgb": 0.0, "integrate": 0.0}
    for i in range(20):
        f = frames[i % 40]
        e.set_pose(f.R, f.t); e.upload_depth(f.depth); e.upload_rgb(f.rgb); e.integrate()
    e.sync()
    t_all = time.perf_counter()
    for i in range(n):
        f = frames[i % 40]
        t0 = time.perf_counter(); e.set_pose(f.R, f.t)
        t1 = time.perf_counter(); e.upload_depth(f.depth)
        t2 = time.perf_counter(); e.upload_rgb(f.rgb)
        t3 = time.perf_counter(); e.integrate()
        t4 = time.perf_counter()
        acc["set_pose"] += t1 - t0; acc["upload_depth"] += t2 - t1; acc["upload_rgb"] += t3 - t2; acc["integrate"] += t4 - t3
    e.sync()
    t_all = time.perf_counter() - t_all
    print(f"MRH_COPY_THREADS={os.environ.get('MRH_COPY_THREADS', 'default')}: {t_all / n * 1e6:.1f} us per frame; "
          + ", ".join(f"{k} {v / n * 1e6:.1f}" for k, v in acc.items()))
    # uploads alone (no kernels to share the device with): staging copy + transfer
    t0 = time.perf_counter()
    for i in range(n):
        f = frames[i % 40]
        e.upload_depth(f.depth); e.upload_rgb(f.rgb)
    t_host = time.perf_counter() - t0
    e.sync()
    hipmem.synchronize()
    print(f"uploads alone: {t_host / n * 1e6:.1f} us per frame")
    # the same loop with inputs already in HBM: the host cost of mrh_integrate's own enqueues
    res = bench.Resident(frames, K)
    e2 = bench.make_engine(hip, params, K)
    res.run(e2, 0, 20); e2.sync()
    t0 = time.perf_counter()
    for r in range(n // 20):
        res.run(e2, 20, 40)
    t_enq = time.perf_counter() - t0
    e2.sync()
    t_tot = time.perf_counter() - t0
    e.close()  # MRH_DEBUG=1: the frame worker's own account of the host-fed loop
    print(f"resident inputs: enqueue {t_enq / (n // 20 * 20) * 1e6:.1f} us per frame of host time, {t_tot / (n // 20 * 20) * 1e6:.1f} us per frame to completion")


if __name__ == "__main__":
    main()
