#!/bin/bash
# mesh tests, the single-GPU bench-line test, two driver-command lines (legs time before any profiled pass), the soak
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04d
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_bench_gpu.py -x -q -m gpu -k "mesh or records or counting or known or single_gpu_line" 2>&1 | tail -4
for i in 1 2; do
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04d/bench_line_driver_command_$i.json 2> gpurun_out/r04d/bench_driver_$i.err
python - $i <<PY
import json, sys
d=json.load(open('gpurun_out/r04d/bench_line_driver_command_%s.json' % sys.argv[1]))
m=d['mc']; l=d['lidar']; s=d['spherical_images']
print('value', round(d['value']), 'extract', round(m['extract_ms_in_library'],4), 'k_mc', m['k_mc_count_ms'], m['k_mc_emit_ms'], 'mc frac', m['roofline']['frac'], 'traffic', m['roofline']['traffic'],
      'lidar', round(l['us_per_scan'],1), l['roofline']['frac'], 'sph', s['ms_per_frame'], s['passes_ms_per_frame'], s['roofline']['frac'], 'splat', d['splat']['frames_per_s_with_seeding'], 'pcie', d['pcie_inclusive_frames_per_s'], d.get('parity_checked'))
PY
done
timeout 1500 python tests/soak.py 500 ${SOAK:-120} 2>&1 | tail -3 | tee gpurun_out/r04d/soak.txt
