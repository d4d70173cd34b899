#!/bin/bash
# kernel timeline of the pipelined frames: gaps between consecutive integrations, where the front halves run
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_pipe; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --pmc-inner --steps ${STEPS:-40} --warmup ${WARM:-8} > $OUT/log.txt 2>&1
python - <<PY
import csv, glob
fn = glob.glob('$OUT/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(fn))]
ev = []
for r in rows:
    n = r['Kernel_Name']
    k = 'back' if 'k_back' in n else 'front' if 'k_front' in n else 'reclaim' if 'k_reclaim' in n else None
    if k: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k, r.get('Queue_Id', '?')))
ev.sort()
t0 = ev[0][0]
backs = [e for e in ev if e[2] == 'back']
fronts = [e for e in ev if e[2] == 'front']
print('queues:', sorted(set((e[2], e[3]) for e in ev)))
for i in range(int("${FROM:-20}"), min(int("${TO:-36}"), len(backs) - 1)):
    b = backs[i]; nb = backs[i + 1]
    f = [x for x in fronts if x[0] >= b[0] - 40000 and x[0] < nb[0]]
    print(f"back {i}: start {(b[0]-t0)/1e3:9.1f} dur {(b[1]-b[0])/1e3:5.1f} gap_to_next {(nb[0]-b[1])/1e3:5.1f} | fronts:", ' '.join(f"[{(x[0]-t0)/1e3:.1f} +{(x[1]-x[0])/1e3:.1f}]" for x in f))
PY
