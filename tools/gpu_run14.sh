set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_geowrapper_gpu.py tests/test_sharding_gpu.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r03/gpu_suite_14.txt
cat gpurun_out/r03/gpu_suite_14.txt
timeout 600 python tools/bench_cfg3.py 25 > gpurun_out/r03/cfg3_25h.txt 2>&1
cat gpurun_out/r03/cfg3_25h.txt
timeout 300 python tools/dbg_extract.py 2>&1 | grep -E "extract:|call" | tail -4
