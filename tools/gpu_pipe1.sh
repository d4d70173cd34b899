#!/bin/bash
cd $GRAFT_REPO_ROOT; T=${1:-p1}; mkdir -p gpurun_out/$T
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_bench_gpu.py::test_the_timed_entry_point_of_bench_py_matches_the_oracle tests/test_geowrapper_gpu.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/$T/tests.txt
tail -12 gpurun_out/$T/tests.txt
for pipe in 1 0 1 0; do
MRH_PIPE=$pipe timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-pmc --no-cpu > gpurun_out/$T/bench_pipe$pipe.json 2> gpurun_out/$T/bench_pipe$pipe.err
python - <<PY
import json
d=json.load(open('gpurun_out/$T/bench_pipe$pipe.json'))
print('pipe=$pipe value', round(d['value']), 'ms', round(d['ms_per_step']*1e3,2), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,2), 'k_front/insert', round(d['roofline']['k_front_ms_avg']*1e3,2))
PY
done
for w in 1024 1536 3072 4096; do
MRH_PIPE_BACK_WGS=$w timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-pmc --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('back_wgs=$w value', round(d['value']), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,2))"
done
python bench.py --steps 100 --warmup 10 --no-extras --no-pmc --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('100 steps value', round(d['value']), 'k_back', round(d['roofline']['kernel_ms_avg']*1e3,2), 'frac', d['roofline']['frac'])"
