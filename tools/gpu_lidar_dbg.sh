#!/bin/bash
# timing experiments on the bucket scans: phases left out one at a time (MRH_SCAN_DBG bits; results are wrong then)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for d in ${DBGS:-0 1 2 3 4 8 16 24}; do
  MRH_DEBUG=1 MRH_SCAN_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_dbg$d -o t -- python tools/bench_lidar.py 1 > gpurun_out/st_dbg$d.log 2>&1
  echo "dbg=$d $(grep "last scan" gpurun_out/st_dbg$d.log | tail -1)"
  python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/st_dbg$d/t_kernel_stats.csv')))
print('   ', ' | '.join('%s %.1f' % (r['Name'].split('(')[0].replace('mrh::',''), float(r['AverageNs'])/1e3) for r in rows if 'k_scan' in r['Name'] or 'alloc3d' in r['Name']))
PY
  rm -rf gpurun_out/st_dbg$d
done
