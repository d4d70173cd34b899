#!/bin/bash
for ov in 1 0; do echo -n "OVERLAP $ov : "; MRH_OVERLAP=$ov python bench.py --steps 200 --warmup 20 --no-cpu 2>&1 | grep '^{"metric' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms_avg'], round(d['roofline']['frac'],3))"; done
bash tools/rocprof_stats.sh ov
