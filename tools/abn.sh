#!/bin/bash
# interleaved timing of several library builds on one box.  usage: abn.sh reps lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT"
R=$1; shift
for i in $(seq 1 $R); do
  for L in "$@"; do
    echo -n "$L : "; python tools/bench_with_lib.py $L --steps 200 --warmup 20 --no-cpu 2>&1 | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'fps  k_ms', round(d['roofline']['kernel_ms_avg']*1000,2), 'us frac', round(d['roofline']['frac'],4))"
  done
done
