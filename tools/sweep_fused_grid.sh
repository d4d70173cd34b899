#!/bin/bash
# tuning sweep for the fused integrate kernel: waves per block (MRH_FUSED_NB) x grid size (MRH_FUSED_GRID)
for nb in 2 1; do for g in 1024 2048 4096 8192; do echo "NB $nb GRID $g"; MRH_FUSED_NB=$nb MRH_FUSED_GRID=$g python bench.py --steps 100 --warmup 10 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"; done; done
