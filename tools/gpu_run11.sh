set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03/gpu_suite_11.txt
cat gpurun_out/r03/gpu_suite_11.txt
timeout 1500 bash tools/profile_bench.sh r03 > gpurun_out/r03/profile_bench.log 2>&1
tail -40 gpurun_out/r03/profile_bench.log | cut -c1-220
