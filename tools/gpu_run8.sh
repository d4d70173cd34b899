set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_lidar_gpu.py tests/test_parity_gpu.py tests/test_geowrapper_gpu.py tests/test_sharding_gpu.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r03/gpu_suite_8.txt
cat gpurun_out/r03/gpu_suite_8.txt
timeout 300 python tests/bench_lidar.py 40 --no-cpu 2>&1 | tail -3 > gpurun_out/r03/lidar_40c.txt
cat gpurun_out/r03/lidar_40c.txt
MRH_LIDAR_SORT_ROCPRIM=1 timeout 300 python tests/bench_lidar.py 40 --no-cpu 2>&1 | tail -2
timeout 300 python tools/bench_cfg3.py 25 2>&1 | tail -12 > gpurun_out/r03/cfg3_25f.txt
cat gpurun_out/r03/cfg3_25f.txt
timeout 300 python tools/bench_cfg3.py 110 2>&1 | tail -6
