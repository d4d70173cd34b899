#!/usr/bin/env python3
"""Tuning aid: where the kernels of consecutive LiDAR scans lie in time (rocprofv3 --kernel-trace of tools/bench_lidar.py):
per scan, the start and end of each launch relative to the scan's k_scan_walk start, and how far k_alloc3d of the NEXT scan
has come when this scan's k_scan_apply ends.  usage: python tools/trace_scan_streams.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for k in ("k_alloc3d", "k_scan_walk", "k_scan_offsets", "k_scan_place", "k_scan_apply"):
        if k in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
walks = [i for i, r in enumerate(rows) if r[2] == "k_scan_walk"]
for wi in walks[-6:]:
    t0 = rows[wi][0]
    near = [r for r in rows if -40000 < r[0] - t0 < 90000]
    print(" | ".join(f"{k[2:] if k.startswith('k_') else k} {(a - t0) / 1e3:6.1f}..{(b - t0) / 1e3:6.1f}" for a, b, k in near))
