#!/bin/bash
cd $GRAFT_REPO_ROOT; T=${1:-b1}; mkdir -p gpurun_out/$T
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$T/bench_driver_cmd.json 2> gpurun_out/$T/bench_driver_cmd.err
python - <<PY
import json
d=json.load(open('gpurun_out/$T/bench_driver_cmd.json'))
r=d['roofline']
print('value', round(d['value']), 'us', round(d['ms_per_step']*1e3,2), 'parity', d.get('parity_checked'))
print('k_back', round(r['kernel_ms_avg']*1e3,2), 'frac', round(r['frac'],3), 'serial', r.get('serial',{}).get('frac'), r.get('serial',{}).get('kernel_ms_avg'), 'frame', r['frame']['frac'], 'traffic', r['traffic'])
print('hbm', d['roofline_hbm']['frac'], d['roofline_hbm']['frames_per_s'])
print('mc', d['mc']['multires_frames_per_s'], d['mc']['extract_ms_in_library'], d['mc']['k_mc_count_ms'], d['mc']['k_mc_emit_ms'], d['mc']['roofline']['frac'], d['mc']['roofline']['traffic'])
print('lidar', d['lidar']['us_per_scan'], 'splat', d['splat']['frames_per_s_with_seeding'], 'pcie', d['pcie_inclusive_frames_per_s'], 'sph', d['spherical_images']['ms_per_frame'], 'periodic', d['periodic_frames'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
