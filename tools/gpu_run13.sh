set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_geowrapper_gpu.py tests/test_sharding_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03/gpu_suite_13.txt
cat gpurun_out/r03/gpu_suite_13.txt
timeout 600 python tools/trace_mc.py 25 > gpurun_out/r03/trace_mc_25b.txt 2>&1
cat gpurun_out/r03/trace_mc_25b.txt
timeout 600 python tools/bench_cfg3.py 25 > gpurun_out/r03/cfg3_25g.txt 2>&1
cat gpurun_out/r03/cfg3_25g.txt
