#!/bin/bash
cd $GRAFT_REPO_ROOT; T=${1:-p10}; mkdir -p gpurun_out/$T
exec > gpurun_out/$T/out.txt 2>&1
run() { timeout 300 python3 bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-extras --no-pmc --no-cpu 2>gpurun_out/$T/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 value', round(d['value']), 'us', round(d['ms_per_step']*1e3,2), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,2), 'k_front', round(d['roofline']['k_front_ms_avg']*1e3,2))" || tail -3 gpurun_out/$T/err_$1.txt; }
for i in 1 2 3; do
run base
MRH_PIPE_GRID=512 run grid512
MRH_PIPE_GRID=768 run grid768
MRH_PIPE_GRID=896 run grid896
MRH_PIPE_GRID=1280 run grid1280
MRH_PIPE_GRID=2048 run grid2048
MRH_PIPE_DEFER=2 run defer2
MRH_SWEEP_WGS=64 run sweep64
MRH_SWEEP_WGS=256 run sweep256
done
STEPS=100 run base100
STEPS=100 MRH_PIPE_GRID=2048 run grid2048_100
STEPS=100 MRH_PIPE_DEFER=2 run defer2_100
STEPS=100 MRH_PIPE_PERIOD=64 run period64_100
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pipelined or replica_640x480_stream or fuzz" 2>&1 | tail -5
