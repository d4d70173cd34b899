set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03/gpu_suite_1.txt
cat gpurun_out/r03/gpu_suite_1.txt
timeout 600 python bench.py > gpurun_out/r03/bench_default_1.json 2> gpurun_out/r03/bench_default_1.err
tail -c 3000 gpurun_out/r03/bench_default_1.json
tail -5 gpurun_out/r03/bench_default_1.err
