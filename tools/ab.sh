#!/bin/bash
# A/B timing of two library builds on the same box: interleaved bench runs.  usage: ab.sh libA.so libB.so [reps]
cd "$GRAFT_REPO_ROOT"
for i in $(seq 1 ${3:-3}); do
  for L in $1 $2; do
    echo -n "$L : "; python tools/bench_with_lib.py $L --steps 200 --warmup 20 --no-cpu 2>&1 | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'fps  k_ms', round(d['roofline']['kernel_ms_avg']*1000,2), 'us frac', round(d['roofline']['frac'],4))"
  done
done
