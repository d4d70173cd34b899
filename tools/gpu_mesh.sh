#!/bin/bash
# mesh read-back on the GPU box: post-process tests, then the extraction of the bench line's mc leg with the fp32 link and with
# the round-3 doubles (MRH_MESH_F64_LINK=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_geowrapper_gpu.py -x -q -m gpu -k "mesh or records or counting or extract" 2>&1 | tail -8
for v in "" "MRH_MESH_F64_LINK=1" "MRH_STAGE_WGS=32" "MRH_STAGE_WGS=128" "MRH_COPY_THREADS=7" ""; do
  env $v timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-pmc > gpurun_out/mc_line.json 2> gpurun_out/mc_line.err
  python - "$v" <<PY
import json, sys
d=json.load(open('gpurun_out/mc_line.json'))
m=d['mc']
print(sys.argv[1].ljust(22), 'value', round(d['value']), 'extract_ms', round(m['extract_ms_in_library'], 4), [round(x, 3) for x in m.get('extract_ms_runs')], 'k_mc', round(m['k_mc_count_ms'], 4))
PY
done
MRH_DEBUG=1 timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-pmc 2>&1 >/dev/null | grep "mrhash_hip\] \(extract\|mesh\)" | tail -6
