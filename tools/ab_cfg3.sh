#!/bin/bash
# A/B of the multi-resolution frame time between two library builds on one box.  usage: ab_cfg3.sh libA.so libB.so [reps]
cd "$GRAFT_REPO_ROOT"
for i in $(seq 1 ${3:-3}); do
  for L in $1 $2; do
    echo -n "$L : "; python - "$L" <<'PY' 2>&1 | grep "multi-res:" | head -1
import os, sys
sys.path.insert(0, os.getcwd())
from mrhash_amd import capi
capi.HIP_LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench_cfg3.py", "110"]
exec(open("tools/bench_cfg3.py").read())
PY
  done
done
