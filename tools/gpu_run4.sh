set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "multires or mesh or 640 or variance or fuzz" 2>&1 | tail -8 > gpurun_out/r03/gpu_suite_4.txt
cat gpurun_out/r03/gpu_suite_4.txt
timeout 300 python tools/bench_cfg3.py 25 2>&1 | tail -12 > gpurun_out/r03/cfg3_25c.txt
cat gpurun_out/r03/cfg3_25c.txt
timeout 300 python tools/bench_cfg3.py 110 2>&1 | tail -12 > gpurun_out/r03/cfg3_110c.txt
cat gpurun_out/r03/cfg3_110c.txt
