#!/bin/bash
cd $GRAFT_REPO_ROOT; T=${1:-p4}; mkdir -p gpurun_out/$T
run() { timeout 300 python3 bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-extras --no-pmc --no-cpu 2>gpurun_out/$T/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 value', round(d['value']), 'us', round(d['ms_per_step']*1e3,2), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,2), 'k_front', round(d['roofline']['k_front_ms_avg']*1e3,2))" || tail -3 gpurun_out/$T/err_$1.txt; }
run lazy
MRH_PIPE=0 run serial_pipe0
MRH_PIPE_SERIAL=1 run serial_env
STEPS=100 run lazy100
STEPS=100 MRH_PIPE=0 run serial100
MRH_PIPE_PERIOD=8 run period8
MRH_PIPE_PERIOD=1000 run period1000
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_bench_gpu.py::test_the_timed_entry_point_of_bench_py_matches_the_oracle -m gpu -x -q 2>&1 | tail -25 > gpurun_out/$T/tests.txt
tail -15 gpurun_out/$T/tests.txt
