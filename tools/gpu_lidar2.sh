#!/bin/bash
# scan tests, then the phase experiments
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lidar_gpu.py -x -q -m gpu 2>&1 | tail -5
for b in 1 0; do MRH_LIDAR_BUCKETS=$b timeout 300 python tools/bench_lidar.py 3 2>&1 | tail -1; done
tools/gpu_lidar_dbg.sh
