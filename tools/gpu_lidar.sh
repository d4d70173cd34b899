#!/bin/bash
# LiDAR scans on the GPU box: the scan tests, A/B of the two record paths, kernel stats of the bucket path.  usage: tools/gpu_lidar.sh [tests|notests]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
if [ "${1:-tests}" = tests ]; then
  timeout 900 python -m pytest tests/test_lidar_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/lidar_tests.log
  cat gpurun_out/lidar_tests.log
fi
for b in 1 0 1 0; do MRH_LIDAR_BUCKETS=$b timeout 300 python tools/bench_lidar.py 3 2>&1 | tail -1; done | tee gpurun_out/lidar_ab.log
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_lidar -o t -- python tools/bench_lidar.py 2 > gpurun_out/st_lidar.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/st_lidar/t_kernel_stats.csv')))
for r in rows[:14]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), '%8.1f us avg' % (float(r['AverageNs'])/1e3), 'min %7.1f max %7.1f' % (float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
rm -f gpurun_out/st_lidar/t_kernel_trace.csv
