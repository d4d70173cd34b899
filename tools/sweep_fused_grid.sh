#!/bin/bash
# tuning sweep for the fused integrate kernel: variant x waves per block (MRH_FUSED_NB) x grid size (MRH_FUSED_GRID)
for pipe in 0 1; do for nb in 2 1; do for g in 512 1024 2048 4096; do echo -n "PIPE $pipe NB $nb GRID $g : "; MRH_FUSED_PIPE=$pipe MRH_FUSED_NB=$nb MRH_FUSED_GRID=$g python bench.py --steps 100 --warmup 10 --no-cpu 2>&1 | grep '^{"metric' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel_ms_avg'], round(d['roofline']['frac'],3))"; done; done; done
