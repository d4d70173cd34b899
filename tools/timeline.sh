#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python bench.py --steps 60 --warmup 10 --no-cpu > gpurun_out/tl.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/tl/t_kernel_trace.csv')))
rows=[r for r in rows if 'mrh::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
# print a window in steady state of the first (unprofiled) pass
sel=rows[200:236]
for r in sel:
    n=r['Kernel_Name'].split('(')[0].replace('void mrh::','').replace('mrh::','')[:28]
    print('%-28s q=%s start=%9.1f end=%9.1f dur=%6.1f' % (n, r.get('Queue_Id','?'), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
rm -rf gpurun_out/tl
