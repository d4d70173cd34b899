#!/bin/bash
cd $GRAFT_REPO_ROOT; T=${1:-p8}; mkdir -p gpurun_out/$T
run() { timeout 300 python3 bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-extras --no-pmc --no-cpu 2>gpurun_out/$T/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 value', round(d['value']), 'us', round(d['ms_per_step']*1e3,2), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,2), 'k_front', round(d['roofline']['k_front_ms_avg']*1e3,2), 'serial k_back', round(d['roofline'].get('serial',{}).get('kernel_ms_avg',0)*1e3,2))" || tail -3 gpurun_out/$T/err_$1.txt; }
for g in 1024 2048 3072 4096 8192; do MRH_FUSED_GRID=$g run grid$g; done
for g in 2048 4096; do STEPS=100 MRH_FUSED_GRID=$g run grid${g}_100; done
