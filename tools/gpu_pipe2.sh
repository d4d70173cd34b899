#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-pmc --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 value', round(d['value']), 'us', round(d['ms_per_step']*1e3,2), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,2), 'k_insert', round(d['roofline']['k_front_ms_avg']*1e3,2))"; }
run base
MRH_DBG_INSERT=1 run tiles_only
MRH_DBG_INSERT=2 run sweep_only
for w in 32 64 256 512; do MRH_SWEEP_WGS=$w run sweep_wgs_$w; done
MRH_PIPE=0 run pipe0
