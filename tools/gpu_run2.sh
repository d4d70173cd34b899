set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_geowrapper_gpu.py tests/test_splat_gpu.py tests/test_sharding_gpu.py tests/test_lidar_gpu.py -m gpu -x -q --durations=12 2>&1 | tail -40 > gpurun_out/r03/gpu_suite_2.txt
cat gpurun_out/r03/gpu_suite_2.txt
MRH_DEBUG=1 timeout 300 python tools/bench_cfg3.py 25 2>&1 | grep -v "voxel->block\|division by" | tail -30 > gpurun_out/r03/cfg3_25.txt
cat gpurun_out/r03/cfg3_25.txt
timeout 300 python tools/bench_cfg3.py 110 2>&1 | tail -12 > gpurun_out/r03/cfg3_110.txt
cat gpurun_out/r03/cfg3_110.txt
