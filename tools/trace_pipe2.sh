#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_pipe; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --pmc-inner --steps ${STEPS:-100} --warmup ${WARM:-10} > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
fn = glob.glob('$OUT/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(fn))]
backs = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'k_back' in r['Kernel_Name'])
fronts = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'k_front' in r['Kernel_Name'])
cad = [(backs[i+1][0]-backs[i][0])/1e3 for i in range(len(backs)-1)]
dur = [(b[1]-b[0])/1e3 for b in backs]
import statistics as st
print('backs', len(backs), 'cadence median', round(st.median(cad),1), 'mean', round(st.mean(cad),1), 'dur median', round(st.median(dur),1), 'front dur median', round(st.median([(f[1]-f[0])/1e3 for f in fronts]),1))
big = [(i, round(c,1)) for i, c in enumerate(cad) if c > 45]
print('cadence > 45 us at', big)
for lo in range(0, len(cad), 10): print(lo, [round(c) for c in cad[lo:lo+10]], 'dur', [round(d) for d in dur[lo:lo+10]])
PY
