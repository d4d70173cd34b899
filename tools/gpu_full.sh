#!/bin/bash
cd $GRAFT_REPO_ROOT; T=${1:-full}; mkdir -p gpurun_out/$T
timeout 2700 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -30 > gpurun_out/$T/gpu_suite.txt
tail -30 gpurun_out/$T/gpu_suite.txt
