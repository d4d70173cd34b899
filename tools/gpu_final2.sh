#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_geowrapper_gpu.py -x -q -m gpu -k "mesh or records or known or extract" 2>&1 | tail -3
S=$SECONDS
bash tools/final_run_r04.sh r04f > gpurun_out/final_r04f.log 2>&1
echo "final_run seconds $((SECONDS-S))"
S=$SECONDS; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04f/bench_line_driver_command_2.json 2>/dev/null; echo "driver command wall seconds $((SECONDS-S))"
tail -3 gpurun_out/final_r04f.log
