#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python tools/bench_spherical.py 12 4
python tools/bench_spherical.py 40 3
MRH_PIPE=0 python tools/bench_spherical.py 40 3
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
MRH_PIPE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_sph -o t -- python tools/bench_spherical.py 40 2 > gpurun_out/st_sph.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/st_sph/t_kernel_stats.csv')))
for r in rows[:14]: print(r['Name'][:110].ljust(110), r['Calls'].rjust(5), '%8.1f us avg' % (float(r['AverageNs'])/1e3))
PY
cp gpurun_out/st_sph/t_kernel_stats.csv gpurun_out/sph_kernel_stats.csv; rm -rf gpurun_out/st_sph
