#!/bin/bash
# full GPU suite + the driver's bench command.  usage: tools/gpu_suite_bench.sh <tag>
cd $GRAFT_REPO_ROOT; T=${1:-full}; mkdir -p gpurun_out/$T
timeout 2700 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -14 > gpurun_out/$T/gpu_suite.txt
tail -14 gpurun_out/$T/gpu_suite.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$T/bench_line_driver_command.json 2> gpurun_out/$T/bench_driver.err
python - <<PY
import json
d=json.load(open('gpurun_out/$T/bench_line_driver_command.json'))
print('value', d['value'], 'frac', d['roofline']['frac'], 'lidar us', d['lidar']['us_per_scan'], 'traffic', d['lidar']['roofline'].get('traffic'), 'mc', d.get('mc',{}).get('extract_ms_in_library'), 'pcie', d.get('pcie_inclusive_frames_per_s'))
PY
