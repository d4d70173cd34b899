set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_lidar_gpu.py tests/test_parity_gpu.py -m gpu -x -q -k "lidar or scan or multires or mesh or 640 or interleaved or far_end" 2>&1 | tail -8 > gpurun_out/r03/gpu_suite_5.txt
cat gpurun_out/r03/gpu_suite_5.txt
timeout 300 python tests/bench_lidar.py 40 --no-cpu 2>&1 | tail -3 > gpurun_out/r03/lidar_40.txt
cat gpurun_out/r03/lidar_40.txt
timeout 300 python tools/bench_cfg3.py 25 2>&1 | tail -12 > gpurun_out/r03/cfg3_25d.txt
cat gpurun_out/r03/cfg3_25d.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_lidar -o t -- python tests/bench_lidar.py 40 --no-cpu > gpurun_out/r03/prof_lidar.log 2>&1
head -24 gpurun_out/r03/prof_lidar/t_kernel_stats.csv | cut -c1-200
rm -f gpurun_out/r03/prof_lidar/t_kernel_trace.csv
