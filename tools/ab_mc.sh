#!/bin/bash
# A/B of the marching-cubes kernels (tools/bench_cfg3.py) between two library builds on one box.  usage: ab_mc.sh libA.so libB.so [reps]
cd "$GRAFT_REPO_ROOT"
for i in $(seq 1 ${3:-2}); do
  for L in $1 $2; do
    echo "$L :"; python - "$L" <<'PY' 2>&1 | grep "k_mc"
import os, sys
sys.path.insert(0, os.getcwd())
from mrhash_amd import capi
capi.HIP_LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench_cfg3.py", "110"]
exec(open("tools/bench_cfg3.py").read())
PY
  done
done
