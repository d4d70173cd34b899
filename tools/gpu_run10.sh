set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14 > gpurun_out/r03/gpu_suite_full_2.txt
cat gpurun_out/r03/gpu_suite_full_2.txt
timeout 900 python bench.py > gpurun_out/r03/bench_default_3.json 2> gpurun_out/r03/bench_default_3.err
tail -c 400 gpurun_out/r03/bench_default_3.json; tail -3 gpurun_out/r03/bench_default_3.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_cmd_2.json 2> gpurun_out/r03/bench_driver_cmd_2.err
head -c 300 gpurun_out/r03/bench_driver_cmd_2.json
MRH_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 8 --steps 30 --warmup 5 --blocks 65536 > gpurun_out/r03/bench_8ranks_one_device_gloo.json 2> gpurun_out/r03/bench_8ranks.err
head -c 300 gpurun_out/r03/bench_8ranks_one_device_gloo.json
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python -c "
import os, sys
sys.argv = ['bench.py', '--gpus', '1', '--steps', '120', '--warmup', '10']
sys.path.insert(0, os.getcwd())
import bench
a = bench.parse_args()
sys.stdout.flush(); bench._RESULT_FD = os.dup(1); os.dup2(2, 1)
bench.bench_multi(a)" > gpurun_out/r03/bench_1rank_rccl.json 2> gpurun_out/r03/bench_1rank_rccl.err
head -c 300 gpurun_out/r03/bench_1rank_rccl.json
