#!/bin/bash
# timeline of the last extraction of bench.py's mc leg: every kernel with its start relative to the first, its duration and the gap before it
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_mc -o t -- python bench.py --pmc-inner-mc --steps 20 --warmup 5 > gpurun_out/tr_mc.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/tr_mc/**/t_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last extraction: from the last k_compact before the last k_mc<false> to the last k_copy_out
names = [r['Kernel_Name'] for r in rows]
last_mc = max(i for i, n in enumerate(names) if 'k_mc<false>' in n)
start = max(i for i in range(last_mc) if 'k_compact' in names[i])
end = max(i for i, n in enumerate(names) if 'k_copy_out' in n)
t0 = int(rows[start]['Start_Timestamp']); prev_end = t0
for r in rows[start:end + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f us  dur %8.1f  gap %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:70]))
    prev_end = max(prev_end, e)
print('total %.1f us' % ((prev_end - t0) / 1e3))
PY
rm -rf gpurun_out/tr_mc
