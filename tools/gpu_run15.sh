set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -12 > gpurun_out/r03/gpu_suite_full_3.txt
cat gpurun_out/r03/gpu_suite_full_3.txt
timeout 900 python bench.py > gpurun_out/r03/bench_default_4.json 2> gpurun_out/r03/bench_default_4.err
tail -c 300 gpurun_out/r03/bench_default_4.json; tail -3 gpurun_out/r03/bench_default_4.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_cmd_3.json 2> gpurun_out/r03/bench_driver_cmd_3.err
head -c 300 gpurun_out/r03/bench_driver_cmd_3.json
