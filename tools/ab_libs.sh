#!/bin/bash
# interleaved A/B of several library builds on one box at the driver's command (and at STEPS2 steps; SERIAL=1: also with the two
# launches of a frame one after the other, k_back alone on the chip), headline pass only, after a parity check of each build
# against the oracle.
#   usage (through gpurun): tools/ab_libs.sh reps name1 name2@VAR=val,VAR2=val ...   (libmrhash_<name>.so, optional environment)
cd "$GRAFT_REPO_ROOT"
R=$1; shift
SPECS=("$@")
envof() { local e="${1#*@}"; [[ "$1" == *@* ]] && echo "${e//,/ }"; }
for sp in "${SPECS[@]}"; do
  n="${sp%%@*}"
  echo "== parity $sp: $(env $(envof $sp) MRH_LIB=mrhash_amd/csrc/libmrhash_$n.so MRH_QP_FRAMES=${QP_FRAMES:-6} python tests/quick_parity.py replica sphere 2>&1 | grep -c 'map OK') of 2 maps OK"
done
one() {  # spec steps warmup [extra env]
  local n="${1%%@*}"
  env $(envof $1) $4 python tools/bench_with_lib.py mrhash_amd/csrc/libmrhash_$n.so --steps $2 --warmup $3 --no-cpu --no-pmc --no-extras 2>/dev/null | grep '^{"metric' | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s %-10s steps %3d: %6d fps  k_back %.2f us frac %.4f  k_front %.2f us' % ('$1', '$4', d['steps'], round(d['value']), r['kernel_ms_avg']*1000, r['frac'], r['k_front_ms_avg']*1000))"
}
for i in $(seq 1 $R); do
  for sp in "${SPECS[@]}"; do
    one $sp 20 5
    one $sp ${STEPS2:-100} 10
    [ -n "$SERIAL" ] && one $sp 60 10 MRH_PIPE=0
  done
done
