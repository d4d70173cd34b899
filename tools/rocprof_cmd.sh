#!/bin/bash
# per-kernel timing of an arbitrary python command. usage: tools/rocprof_cmd.sh <tag> <script> [args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$TAG -o t -- python "$@" > gpurun_out/st_$TAG.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/st_$TAG/t_kernel_stats.csv')))
for r in rows[:24]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), '%9.1f us avg' % (float(r['AverageNs'])/1e3), r['Percentage'])
PY
rm -f gpurun_out/st_$TAG/t_kernel_trace.csv
