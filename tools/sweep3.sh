#!/bin/bash
for tile in 16 8; do for gi in 1 0; do echo -n "ALLOC_TILE $tile GC_INLINE $gi : "; MRH_ALLOC_TILE=$tile MRH_GC_INLINE=$gi python bench.py --steps 100 --warmup 10 --no-cpu 2>&1 | grep '^{"metric' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms_avg'], round(d['roofline']['frac'],3))"; done; done
MRH_ALLOC_TILE=8 bash tools/rocprof_stats.sh t8
