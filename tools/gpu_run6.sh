set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_lidar_gpu.py tests/test_parity_gpu.py tests/test_geowrapper_gpu.py -m gpu -x -q -k "lidar or scan or multires or mesh or 640 or interleaved or far_end or comm_entry or stream" 2>&1 | tail -8 > gpurun_out/r03/gpu_suite_6.txt
cat gpurun_out/r03/gpu_suite_6.txt
timeout 300 python tests/bench_lidar.py 40 --no-cpu 2>&1 | tail -3 > gpurun_out/r03/lidar_40b.txt
cat gpurun_out/r03/lidar_40b.txt
timeout 300 python tools/bench_cfg3.py 25 2>&1 | tail -12 > gpurun_out/r03/cfg3_25e.txt
cat gpurun_out/r03/cfg3_25e.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_lidar -o t -- python tests/bench_lidar.py 40 --no-cpu > gpurun_out/r03/prof_lidar.log 2>&1
head -12 gpurun_out/r03/prof_lidar/t_kernel_stats.csv | cut -c1-150,300-420
rm -f gpurun_out/r03/prof_lidar/t_kernel_trace.csv
