#!/bin/bash
# the round's evidence on the GPU box (through gpurun): bench lines, per-leg kernel statistics, pipeline trace, the N > 1 code
# paths a one-GPU box allows.  Outputs under gpurun_out/<tag>/, copied into profiles/<tag>/ by hand.
set -x
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_line_driver_command.json 2> gpurun_out/$TAG/bench_driver.err
head -c 300 gpurun_out/$TAG/bench_line_driver_command.json
timeout 900 python bench.py > gpurun_out/$TAG/bench_line_default_run.json 2> gpurun_out/$TAG/bench_default.err
head -c 300 gpurun_out/$TAG/bench_line_default_run.json
MRH_PIPE=0 timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu --no-extras > gpurun_out/$TAG/bench_line_driver_command_serial.json 2>/dev/null
tools/profile_r04.sh $TAG
STEPS=100 WARM=10 tools/trace_pipe2.sh > gpurun_out/$TAG/pipeline_trace_110_frames.txt 2>&1
MRH_PIPE=0 STEPS=100 WARM=10 tools/trace_pipe2.sh > gpurun_out/$TAG/serial_trace_110_frames.txt 2>&1
python tools/exp_two_engines.py 60 > gpurun_out/$TAG/two_engines.txt 2>&1
tools/micro/two_stream_events > gpurun_out/$TAG/two_stream_events.txt 2>&1
python tools/host_path_breakdown.py --frames 200 > gpurun_out/$TAG/host_path.txt 2>&1
MRH_PIPE_UPLOADS=1 python tools/host_path_breakdown.py --frames 200 >> gpurun_out/$TAG/host_path.txt 2>&1
MRH_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 8 --steps 30 --warmup 5 --blocks 65536 > gpurun_out/$TAG/bench_8ranks_one_device_gloo.json 2> gpurun_out/$TAG/bench_8ranks.err
head -c 200 gpurun_out/$TAG/bench_8ranks_one_device_gloo.json
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python -c "
import os, sys
sys.argv = ['bench.py', '--gpus', '1', '--steps', '120', '--warmup', '10']
sys.path.insert(0, os.getcwd())
import bench
a = bench.parse_args()
sys.stdout.flush(); bench._RESULT_FD = os.dup(1); os.dup2(2, 1)
bench.bench_multi(a)" > gpurun_out/$TAG/bench_1rank_rccl.json 2> gpurun_out/$TAG/bench_1rank_rccl.err
head -c 200 gpurun_out/$TAG/bench_1rank_rccl.json
gcc -std=c11 -O1 -Iinclude examples/comm_smoke.c -o /tmp/comm_smoke -Lmrhash_amd/csrc -lmrhash_hip -Wl,-rpath,$PWD/mrhash_amd/csrc -lm
MRH_COMM_SELF_LOOP=1 /tmp/comm_smoke 1 > gpurun_out/$TAG/comm_smoke_1rank_self_loop.txt 2>&1
tail -3 gpurun_out/$TAG/comm_smoke_1rank_self_loop.txt
