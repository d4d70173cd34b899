#!/bin/bash
# quick per-kernel timing of the bench (kernel-trace only). usage: tools/rocprof_stats.sh <tag> [env...]
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$TAG -o t -- python bench.py --steps 100 --warmup 10 --no-cpu --no-pmc > gpurun_out/st_$TAG.log 2>&1
grep "^{\"metric" gpurun_out/st_$TAG.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fps', d['value'], 'kernel_ms', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'])"
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/st_$TAG/t_kernel_stats.csv')))
for r in rows[:9]: print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), '%8.1f us avg' % (float(r['AverageNs'])/1e3), r['Percentage'])
PY
rm -f gpurun_out/st_$TAG/t_kernel_trace.csv
