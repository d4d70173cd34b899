#!/bin/bash
# GPU run r04a: the new evidence tests, the driver's bench command, per-leg kernel statistics
cd $GRAFT_REPO_ROOT; T=${1:-r04a}; mkdir -p gpurun_out/$T
timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_sharding_gpu.py tests/test_reference_dump.py -m gpu -x -q --durations=8 2>&1 | tail -25 > gpurun_out/$T/tests.txt
cat gpurun_out/$T/tests.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$T/bench_driver_cmd.json 2> gpurun_out/$T/bench_driver_cmd.err
python - <<PY
import json
d=json.load(open('gpurun_out/$T/bench_driver_cmd.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'parity_checked', d.get('parity_checked'), d['cpu_baseline'] and d['cpu_baseline']['parity'])
print('k_back', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'], 'k_front', d['roofline']['k_front_ms_avg'])
print('lidar', d['lidar']['us_per_scan'], d['lidar']['roofline']['traffic'], d['lidar']['roofline'].get('traffic_note'))
print('mc', d['mc']['extract_ms_in_library'], d['mc']['k_mc_count_ms'], d['mc']['k_mc_emit_ms'], 'pcie', d['pcie_inclusive_frames_per_s'], 'sph', d['spherical_images']['ms_per_frame'])
PY
tools/profile_r04.sh $T
