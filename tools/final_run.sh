#!/bin/bash
# the round's closing measurements on the GPU box (through gpurun): complete GPU suite, the bench lines, the N > 1 code paths a
# one-GPU box allows, LiDAR kernel statistics.  Outputs under gpurun_out/<tag>/, copied into profiles/<tag>/ by hand.
set -x
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -12 > gpurun_out/$TAG/gpu_suite_final.txt
cat gpurun_out/$TAG/gpu_suite_final.txt
timeout 900 python bench.py > gpurun_out/$TAG/bench_default_final.json 2> gpurun_out/$TAG/bench_default_final.err
tail -c 300 gpurun_out/$TAG/bench_default_final.json
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_driver_cmd_final.json 2> gpurun_out/$TAG/bench_driver_cmd_final.err
head -c 300 gpurun_out/$TAG/bench_driver_cmd_final.json
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
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof_lidar_final -o t -- python tests/bench_lidar.py 40 --no-cpu > gpurun_out/$TAG/prof_lidar_final.log 2>&1
rm -f gpurun_out/$TAG/prof_lidar_final/t_kernel_trace.csv
tail -1 gpurun_out/$TAG/prof_lidar_final.log
