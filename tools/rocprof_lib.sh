#!/bin/bash
# per-kernel timing of the bench for a given library build. usage: tools/rocprof_lib.sh <tag> <lib.so>
TAG=${1:-x}; LIB=${2:-mrhash_amd/csrc/libmrhash_hip.so}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$TAG -o t -- python tools/bench_with_lib.py $LIB --steps 100 --warmup 10 --no-cpu > gpurun_out/st_$TAG.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/st_$TAG/t_kernel_stats.csv')))
for r in rows[:6]: print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), '%8.1f us avg' % (float(r['AverageNs'])/1e3), r['Percentage'])
PY
rm -f gpurun_out/st_$TAG/t_kernel_trace.csv
