set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_lidar_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r03/gpu_suite_9.txt
cat gpurun_out/r03/gpu_suite_9.txt
timeout 300 python tests/bench_lidar.py 40 --no-cpu 2>&1 | tail -1
MRH_LIDAR_SORT_ROCPRIM=1 timeout 300 python tests/bench_lidar.py 40 --no-cpu 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_lidar2 -o t -- python tests/bench_lidar.py 40 --no-cpu > gpurun_out/r03/prof_lidar2.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03/prof_lidar2/t_kernel_stats.csv')))
for r in rows[:14]:
    n=r['Name']; print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us  {float(r['TotalDurationNs'])/1e3/40:8.1f} us/scan  {n[:90]}")
PY
rm -f gpurun_out/r03/prof_lidar2/t_kernel_trace.csv
