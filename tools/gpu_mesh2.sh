#!/bin/bash
# mesh tests + extraction numbers + kernel stats of the extraction leg
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_geowrapper_gpu.py tests/test_sharding_gpu.py -x -q -m gpu -k "mesh or records or counting or extract or gather or known or multires" 2>&1 | tail -8
for v in "" ""; do
  env $v timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-pmc > gpurun_out/mc_line.json 2> gpurun_out/mc_line.err
  python - "$v" <<PY
import json, sys
d=json.load(open('gpurun_out/mc_line.json'))
m=d['mc']
print(sys.argv[1].ljust(22), 'value', round(d['value']), 'extract_ms', round(m['extract_ms_in_library'], 4), [round(x, 3) for x in m.get('extract_ms_runs')], 'k_mc', round(m['k_mc_count_ms'], 4))
PY
done
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_mc -o t -- python bench.py --pmc-inner-mc --steps 20 --warmup 5 > gpurun_out/st_mc.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/st_mc/t_kernel_stats.csv')))
for r in rows:
    if any(k in r['Name'] for k in ('k_mc','k_mesh','k_copy_out','k_stage','k_block','rocprim','k_list_keys','fillBuffer')): print(r['Name'][:80].ljust(80), r['Calls'].rjust(5), '%8.1f us avg' % (float(r['AverageNs'])/1e3))
PY
cp gpurun_out/st_mc/t_kernel_stats.csv gpurun_out/mc_kernel_stats.csv; rm -rf gpurun_out/st_mc
