#!/bin/bash
# quick parity check of what changed this session: scans + extraction
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_lidar_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "lidar or scan or runs or mesh or records or counting" 2>&1 | tail -4
