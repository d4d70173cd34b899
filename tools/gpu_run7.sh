set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -20 > gpurun_out/r03/gpu_suite_full_1.txt
cat gpurun_out/r03/gpu_suite_full_1.txt
timeout 900 python bench.py > gpurun_out/r03/bench_default_2.json 2> gpurun_out/r03/bench_default_2.err
tail -c 1500 gpurun_out/r03/bench_default_2.json; tail -3 gpurun_out/r03/bench_default_2.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_cmd_1.json 2> gpurun_out/r03/bench_driver_cmd_1.err
head -c 600 gpurun_out/r03/bench_driver_cmd_1.json
