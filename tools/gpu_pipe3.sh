#!/bin/bash
cd $GRAFT_REPO_ROOT; T=${1:-p3}; mkdir -p gpurun_out/$T
run() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-pmc --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 value', round(d['value']), 'us', round(d['ms_per_step']*1e3,2), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,2), 'k_front', round(d['roofline']['k_front_ms_avg']*1e3,2))"; }
run lazy
MRH_PIPE=0 run strict
MRH_PIPE_PERIOD=16 run period16
MRH_PIPE_PERIOD=1000 run period1000
MRH_PIPE_BACK_WGS=1024 run back1024
MRH_PIPE_BACK_WGS=1536 run back1536
MRH_PIPE_BACK_WGS=3072 run back3072
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_bench_gpu.py::test_the_timed_entry_point_of_bench_py_matches_the_oracle -m gpu -x -q 2>&1 | tail -25 > gpurun_out/$T/tests.txt
tail -15 gpurun_out/$T/tests.txt
