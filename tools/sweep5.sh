#!/bin/bash
for sw in 64 128 256; do echo -n "MERGED 1 SWEEP_WGS $sw : "; MRH_SWEEP_WGS=$sw python bench.py --steps 200 --warmup 20 --no-cpu 2>&1 | grep '^{"metric' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms_avg'], round(d['roofline']['frac'],3))"; done
echo -n "MERGED 0 : "; MRH_MERGED=0 python bench.py --steps 200 --warmup 20 --no-cpu 2>&1 | grep '^{"metric' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
bash tools/rocprof_stats.sh mg
