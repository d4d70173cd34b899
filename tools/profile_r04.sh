#!/bin/bash
# Per-leg kernel statistics at the DRIVER's step count: one bench leg per rocprofv3 run, so the
# averages of k_front / k_back / k_mc are read off the CSV instead of being back-solved from a mixed run.
#   usage (GPU box, through gpurun): tools/profile_r04.sh <tag>      -> gpurun_out/<tag>/*.csv
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$name -o t -- python bench.py "$@" > $OUT/tr_$name.log 2>&1
  cp $OUT/tr_$name/t_kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/tr_$name
  head -6 $OUT/${name}_kernel_stats.csv | cut -c1-200
}
run driver_cmd --pmc-inner --steps 20 --warmup 5
run driver_cmd_mc --pmc-inner-mc --steps 20 --warmup 5
run lidar --pmc-inner-lidar
