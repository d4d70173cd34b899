set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_geowrapper_gpu.py tests/test_sharding_gpu.py tests/test_mesh_postprocess.py -m gpu -x -q --durations=5 -k "mesh or Mesh or 640 or rccl or sharded or two_rank or submaps or golden or plain_c or postprocess or runner" 2>&1 | tail -25 > gpurun_out/r03/gpu_suite_3.txt
cat gpurun_out/r03/gpu_suite_3.txt
MRH_DEBUG=1 timeout 300 python tools/bench_cfg3.py 25 2>&1 | grep -v "voxel->block\|division by" | tail -14 > gpurun_out/r03/cfg3_25b.txt
cat gpurun_out/r03/cfg3_25b.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_cfg3 -o t -- python tools/bench_cfg3.py 25 > gpurun_out/r03/prof_cfg3.log 2>&1
head -40 gpurun_out/r03/prof_cfg3/t_kernel_stats.csv
rm -f gpurun_out/r03/prof_cfg3/t_kernel_trace.csv
