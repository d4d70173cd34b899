#!/bin/bash
cd $GRAFT_REPO_ROOT; T=${1:-p5}; mkdir -p gpurun_out/$T
run() { timeout 300 python3 bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-extras --no-pmc --no-cpu 2>gpurun_out/$T/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 value', round(d['value']), 'us', round(d['ms_per_step']*1e3,2), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,2), 'k_front', round(d['roofline']['k_front_ms_avg']*1e3,2))" || tail -3 gpurun_out/$T/err_$1.txt; }
MRH_PIPE_DEFER=1 run defer1
MRH_PIPE_DEFER=2 run defer2
MRH_PIPE_DEFER=1 run defer1
MRH_PIPE_DEFER=2 run defer2
STEPS=100 MRH_PIPE_DEFER=1 run defer1_100
STEPS=100 MRH_PIPE_DEFER=2 run defer2_100
MRH_DEBUG=1 python tools/exp_host_enqueue.py 40 2>&1 | grep -E "pipelined frames|enqueue" | tail -2
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_bench_gpu.py::test_the_timed_entry_point_of_bench_py_matches_the_oracle -m gpu -x -q 2>&1 | tail -5
