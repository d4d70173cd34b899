#!/bin/bash
# A/B of environment switches on one box: the driver's bench command once per variant, interleaved, twice.
#   usage (through gpurun): tools/ab_env.sh <tag> "VAR=1" "OTHER=1 MORE=2" ...    (the empty variant "" = defaults is always run)
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  i=0
  for v in "" "$@"; do
    env $v timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 ${AB_ARGS:-} > $OUT/line_${i}_$rep.json 2> $OUT/err_${i}_$rep.txt
    python - <<PY
import json
d=json.load(open('$OUT/line_${i}_$rep.json'))
mc=d.get('mc') or {}; li=d.get('lidar') or {}; r=d['roofline']
print('[%s] rep $rep' % '$v', 'value', round(d['value']), 'k_back ms', round(r.get('kernel_ms_avg') or 0, 5), 'frac', round(r['frac'], 4), 'serial', (d.get('roofline_serial') or r.get('serial') or {}).get('frac') if isinstance(r.get('serial'), dict) else r.get('serial'),
      '| parity', d.get('parity_checked'), '| mc', round(mc.get('extract_ms_in_library') or 0, 3), 'k_mc', mc.get('k_mc_count_ms'), mc.get('k_mc_emit_ms'), 'traffic', (mc.get('roofline') or {}).get('traffic'),
      '| lidar', round(li.get('us_per_scan') or 0, 1), '| pcie', round(d.get('pcie_inclusive_frames_per_s') or 0), 'link', round(d.get('h2d_link_gbs') or 0, 1), round(d.get('pcie_inclusive_frac_of_link') or 0, 3), 'loaded', d.get('h2d_link_gbs_under_load'), d.get('pcie_inclusive_frac_of_link_under_load'))
PY
    i=$((i+1))
  done
done
