#!/bin/bash
# the headline alone, fast and slow Python call paths interleaved
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
for v in "" "MRH_BENCH_SLOW_CALLS=1"; do
env $v python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$v'.ljust(24), round(d['value']), round(d['ms_per_step']*1e3,2), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,1), 'k_front', round(d['roofline']['k_front_ms_avg']*1e3,1))"
done; done
env python3 bench.py --gpus 1 --steps 100 --warmup 10 --no-pmc --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('100 steps', round(d['value']), round(d['ms_per_step']*1e3,2))"
MRH_BENCH_SLOW_CALLS=1 python3 bench.py --gpus 1 --steps 100 --warmup 10 --no-pmc --no-cpu --no-extras 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('100 steps slow', round(d['value']), round(d['ms_per_step']*1e3,2))"
timeout 600 python -m pytest tests/test_bench_gpu.py -x -q -m gpu -k "timed_entry or single_gpu" 2>&1 | tail -3
