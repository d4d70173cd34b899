#!/bin/bash
cd $GRAFT_REPO_ROOT; T=${1:-p7}; mkdir -p gpurun_out/$T
run() { timeout 300 python3 bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-extras --no-pmc --no-cpu 2>gpurun_out/$T/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 value', round(d['value']), 'us', round(d['ms_per_step']*1e3,2), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,2), 'k_front', round(d['roofline']['k_front_ms_avg']*1e3,2))" || tail -3 gpurun_out/$T/err_$1.txt; }
for i in 1 2; do
MRH_PIPE_PRIO=1 run prio1
MRH_PIPE_PRIO=0 run prio0
done
STEPS=100 MRH_PIPE_PRIO=1 run prio1_100
STEPS=100 MRH_PIPE_PRIO=0 run prio0_100
MRH_DEBUG=1 python tools/exp_host_enqueue.py 40 2>&1 | grep -E "pipelined frames|enqueue" | tail -2
