#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 PMC passes written by tools/profile_bench.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_tcc"):
    for fn in glob.glob(os.path.join(root, d, "*counter_collection.csv")):
        for row in csv.DictReader(open(fn)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for k in sorted(acc):
    if not k.startswith("mrh::"):
        continue
    parts = [f"{c}={v[0] / max(v[1], 1):.4g}" for c, v in sorted(acc[k].items())]
    n = max(v[1] for v in acc[k].values())
    print(f"{k}  (dispatches {n}; per-dispatch averages)\n    " + "  ".join(parts))
print("notes: FETCH_SIZE/WRITE_SIZE are in KiB-like units of the tool (x1024 B); on gfx950 FETCH_SIZE under-reports wide "
      "coalesced reads by 2x (MI355X_MICROARCH.md HBM section); SQ_* cycle counters are in quad-cycles.")
