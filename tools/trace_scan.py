#!/usr/bin/env python3
"""Tuning aid: phase stamps of the LiDAR scan kernels (build with -DMRH_SCAN_TRACE, see tools/trace_scan.sh): for the last scan of
the bench's drive, per kernel: when its workgroups start and end relative to the first one, and how long each phase takes."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mrhash_amd import capi
capi.HIP_LIB_PATH = os.path.join(ROOT, "mrhash_amd", "csrc", "libmrhash_trace.so")
hip = capi.load_hip()
le, scans, d_scans, run_scans = bench.lidar_setup(hip, 262144)
run_scans(0, bench.LIDAR_SCANS - 1)
le.sync()
W = 4096
buf = (C.c_uint64 * (4 * W * 8))()
hip.mrh_debug_scan_trace(None, 1)
run_scans(bench.LIDAR_SCANS - 1, bench.LIDAR_SCANS)
le.sync()
hip.mrh_debug_scan_trace(buf, 0)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(4, W, 8)  # (apply's slots 3..6 carry 20 bits of info above bit 44: read back as uint64 below)
tick = 0.01  # us per tick (100 MHz)
names = {0: ("walk", ["start", "beams walked", "set ready", "grouped (A)", "atomics + groups (B)", "records stored (C)"]),
         1: ("offsets", ["start", "end"]), 2: ("place", ["start", "end"]), 3: ("apply", ["start", "chunks done", "end"])}
t0_all = a[:, :, 0][a[:, :, 0] > 0].min()
for k, (nm, slots) in names.items():
    r = a[k][a[k][:, 0] > 0]
    if not len(r):
        continue
    k0 = r[:, 0].min()
    line = [f"{nm}: {len(r)} workgroups, first starts {(k0 - t0_all) * tick:.1f} us after the scan's first stamp"]
    for s, sn in enumerate(slots):
        v = (r[:, s] - k0) * tick
        v = v[r[:, s] > 0]
        line.append(f"  {sn:24s} min {v.min():6.1f}  median {np.median(v):6.1f}  p90 {np.percentile(v, 90):6.1f}  max {v.max():6.1f}  (us after the kernel's first workgroup)")
    if len(slots) > 2:
        for s in range(1, len(slots)):
            ok = (r[:, s] > 0) & (r[:, s - 1] > 0)
            d = (r[ok, s] - r[ok, s - 1]) * tick
            line.append(f"  phase {slots[s - 1]} -> {slots[s]}: median {np.median(d):.1f} us, p90 {np.percentile(d, 90):.1f}, max {d.max():.1f}")
    print("\n".join(line))
    if k == 3:  # apply: its waves one by one (slots 3..6: end stamp | records << 44 | voxels << 54 | long run << 63 of the wave's LAST chunk)
        raw = a[k].view(np.uint64) if a[k].dtype != np.uint64 else a[k]
        w = raw[:, 3:7].reshape(-1)
        w = w[w > 0]
        st = (w & np.uint64((1 << 44) - 1)).astype(np.int64)
        info = (w >> np.uint64(44)).astype(np.int64)
        nr, nv, lng = info & 1023, (info >> 10) & 511, (info >> 19) & 1
        end = (st - (k0 & ((1 << 44) - 1))) * tick
        print(f"  waves with a chunk: {len(w)} (long runs {int(lng.sum())}); wave end median {np.median(end):.1f} p90 {np.percentile(end, 90):.1f} max {end.max():.1f}")
        for name, sel in (("long runs", lng == 1), ("chunks", lng == 0)):
            if sel.any():
                print(f"    {name}: {int(sel.sum())} waves, end median {np.median(end[sel]):.1f} p90 {np.percentile(end[sel], 90):.1f} max {end[sel].max():.1f}; records median {np.median(nr[sel]):.0f} max {nr[sel].max()}")
        for i in np.argsort(-end)[:12]:
            print(f"    slow wave: end {end[i]:5.1f} us  records {nr[i]:4d} voxels {nv[i]:3d} long {lng[i]}")
        order = np.argsort(-end)
        top = order[: max(1, len(order) // 20)]
        print(f"    slowest 5 % of waves: records median {np.median(nr[top]):.0f}, voxels median {np.median(nv[top]):.0f}, long {int(lng[top].sum())} of {len(top)}")
    if k == 0:  # the walk's slowest workgroups, phase by phase (us): who the launch waits for
        idx = np.nonzero(a[k][:, 0] > 0)[0]
        order = idx[np.argsort(-a[k][idx, 5])][:12]
        for w in order:
            ph = np.diff(a[k][w, :6]) * tick
            print(f"  slow wg {w:4d}: start {(a[k][w, 0] - k0) * tick:5.1f} end {(a[k][w, 5] - k0) * tick:5.1f} | walk {ph[0]:5.1f} set {ph[1]:4.1f} A {ph[2]:5.1f} B {ph[3]:5.1f} C {ph[4]:4.1f}")
le.close()
