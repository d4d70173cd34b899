#!/bin/bash
# The round's evidence from ONE build, on the GPU box (through gpurun).  Every file lands under gpurun_out/<tag>/ with the commit
# hash of the build in MANIFEST.txt; the files to be judged are copied into profiles/<tag>/ afterwards (tools/README.md).
#   usage: tools/round.sh <tag> <commit> [sections...]     sections: suite bench default stats pmc framepmc stress soak multi trace tracebuilds churn bigmap starve
#   default sections: bench stats pmc framepmc
TAG=${1:-r05}; COMMIT=${2:-unknown}; shift 2
SECTIONS=${*:-bench stats pmc framepmc}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
echo "commit $COMMIT  sections: $SECTIONS  date $(date -u +%FT%TZ)  lib sha256 $(sha256sum mrhash_amd/csrc/libmrhash_hip.so | cut -c1-16)" >> $OUT/MANIFEST.txt
has() { [[ " $SECTIONS " == *" $1 "* ]]; }

stats_leg() {  # name, command...: kernel statistics of one leg (its own rocprofv3 run)
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$name -o t -- "$@" > $OUT/tr_$name.log 2>&1
  cp $OUT/tr_$name/t_kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/tr_$name
  echo "== $name"; head -8 $OUT/${name}_kernel_stats.csv | cut -c1-160
}
pmc_leg() {  # name, counters (quoted), command...: raw counter CSV of one leg -> per-kernel averages
  local name=$1 ctrs=$2; shift 2
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name/pmc_sq -o p -- "$@" > $OUT/pmc_$name.log 2>&1
  python tools/summarize_pmc.py $OUT/pmc_$name > $OUT/pmc_${name}.txt 2>&1
  rm -rf $OUT/pmc_$name
}

if has suite; then
  timeout 2400 python -m pytest tests -m gpu -q -x --durations=25 2>&1 | tail -40 > $OUT/gpu_suite.txt
  tail -32 $OUT/gpu_suite.txt
fi
if has bench; then
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_driver_command.json 2> $OUT/bench_driver.err
  python - <<PY
import json
d=json.load(open('$OUT/bench_line_driver_command.json'))
mc=d.get('mc') or {}; li=d.get('lidar') or {}
print('value', round(d['value']), 'frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'))
print('mc extract', mc.get('extract_ms_in_library'), mc.get('extract_ms_runs'), 'k_mc', mc.get('k_mc_count_ms'), mc.get('k_mc_emit_ms'), 'traffic', (mc.get('roofline') or {}).get('traffic'))
print('lidar us', li.get('us_per_scan'), 'traffic', (li.get('roofline') or {}).get('traffic'), 'pcie', d.get('pcie_inclusive_frames_per_s'), 'link', d.get('h2d_link_gbs'), d.get('pcie_inclusive_frac_of_link'), 'under load', d.get('h2d_link_gbs_under_load'), d.get('pcie_inclusive_frac_of_link_under_load'))
PY
fi
if has default; then  # the bench with no flags (100 steps, 10 warm-up: census and starve frames inside)
  timeout 1200 python bench.py > $OUT/bench_line_default_run.json 2> $OUT/bench_default.err
  python -c "import json; d=json.load(open('$OUT/bench_line_default_run.json')); print('default run: value', round(d['value']), 'steps', d['steps'], 'parity', d.get('parity_checked'))"
  MRH_PIPE=0 timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu --no-extras > $OUT/bench_line_driver_command_serial.json 2>/dev/null
fi
if has stats; then
  stats_leg driver_cmd python bench.py --pmc-inner --steps 20 --warmup 5
  stats_leg driver_cmd_mc python bench.py --pmc-inner-mc --steps 20 --warmup 5
  stats_leg lidar python bench.py --pmc-inner-lidar
  stats_leg sph python bench.py --pmc-inner-sph
fi
if has framepmc; then
  # what the two launches of a frame do with their cycles, pipelined and serial, at the driver's step count
  A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
  B="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
  pmc_leg frame_pipe_a "$A" python bench.py --pmc-inner --steps 20 --warmup 5
  pmc_leg frame_pipe_b "$B" python bench.py --pmc-inner --steps 20 --warmup 5
  MRH_PIPE=0 pmc_leg frame_serial_a "$A" python bench.py --pmc-inner --steps 20 --warmup 5
  MRH_PIPE=0 pmc_leg frame_serial_b "$B" python bench.py --pmc-inner --steps 20 --warmup 5
  { echo "# frame counters (rocprofv3 --pmc, one pass per counter set; SQ cycle counters in quad-cycles; commit $COMMIT)";
    for n in frame_pipe_a frame_pipe_b frame_serial_a frame_serial_b; do echo "## $n"; grep -A1 -E "k_back<|k_front<" $OUT/pmc_$n.txt; done; } > $OUT/frame_pmc.txt
  cat $OUT/frame_pmc.txt | cut -c1-400
fi
if has pmc; then  # raw FETCH_SIZE / WRITE_SIZE passes behind the bench line's `traffic` figures
  for leg in "driver --pmc-inner --steps 20 --warmup 5" "mc --pmc-inner-mc --steps 20 --warmup 5" "lidar --pmc-inner-lidar" "sph --pmc-inner-sph"; do
    set -- $leg; name=$1; shift
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${name}_$ctr/pmc_fetch -o p -- python bench.py "$@" > $OUT/pmc_${name}_$ctr.log 2>&1
      python tools/summarize_pmc.py $OUT/pmc_${name}_$ctr > $OUT/traffic_${name}_$ctr.txt 2>&1
      rm -rf $OUT/pmc_${name}_$ctr
    done
  done
  grep -h -A1 "k_back<\|k_mc<\|k_scan_walk" $OUT/traffic_*.txt | cut -c1-200 | head -40
fi
if has stress; then
  MRH_WIDEN_REPORT=1 python tools/stress_extract.py 400 0 > $OUT/stress_extract.txt 2>&1
  MRH_WIDEN_REPORT=1 python tools/stress_extract.py 400 300 >> $OUT/stress_extract.txt 2>&1
  cat $OUT/stress_extract.txt
fi
if has soak; then
  python tests/soak.py 900 ${SOAK_SEEDS:-100} > $OUT/soak.txt 2>&1; tail -3 $OUT/soak.txt
fi
if has churn; then
  # the 3 000-frame walk as a pytest gate: fails the section if the walk, its rebuild count or the parity at its end fails
  { sha256sum mrhash_amd/csrc/libmrhash_hip.so | cut -c1-16; MRH_SOAK=1 timeout 2400 python -m pytest tests/test_churn_gpu.py -m soak -q -s 2>&1 | tail -6; } > $OUT/soak_churn.txt 2>&1
  tail -3 $OUT/soak_churn.txt; grep -q " passed" $OUT/soak_churn.txt || echo "CHURN GATE FAILED"
fi
if has starve; then  # the kernels of starve frames inside the pipeline (every 5th of 50 resident frames) + the frame rates with and without them
  tools/rocprof_cmd.sh starve tools/bench_starve.py 60 5 > $OUT/starve_kernel_stats.txt 2>&1
  grep "period" gpurun_out/st_starve.log >> $OUT/starve_kernel_stats.txt; rm -rf gpurun_out/st_starve gpurun_out/st_starve.log
  grep "starve\|period" $OUT/starve_kernel_stats.txt
fi
if has trace; then
  STEPS=100 WARM=10 tools/trace_pipe2.sh > $OUT/pipeline_trace_110_frames.txt 2>&1; head -3 $OUT/pipeline_trace_110_frames.txt
fi
if has tracebuilds; then  # the two tracing variants of the library (never shipped): LiDAR scan phases, k_front / k_back phases alone on the chip
  tools/trace_scan.sh > $OUT/scan_trace.txt 2>&1; grep -c workgroups $OUT/scan_trace.txt
  MRH_PIPE=0 tools/trace_kback.sh 30 2>&1 | grep -v "warning\|hipDeviceSynchronize\|hipMemcpy(h.data\|\^~\|generated when\|amdgpu.ids" > $OUT/kfront_kback_trace_serial.txt; head -8 $OUT/kfront_kback_trace_serial.txt
  rm -f mrhash_amd/csrc/libmrhash_trace.so
fi
if has bigmap; then  # extraction around and beyond k_block_rank's 32 k blocks: what the canonical order costs
  for v in 0.0068 0.005 0.0035; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_big -o t -- python tools/bench_big_map.py $v 4 > $OUT/big_$v.log 2>&1
    grep "^voxel" $OUT/big_$v.log
    python - <<PY
import csv
for r in csv.DictReader(open("$OUT/tr_big/t_kernel_stats.csv")):
    n = r["Name"]
    if any(k in n for k in ("k_block_rank", "k_sort_", "k_mc_scan_total", "k_block_scatter", "k_list_keys", "k_mc<", "k_mc_neigh", "k_mc_emit")):
        print("   %-40s calls %3s avg %9.1f us" % (n.split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    rm -rf $OUT/tr_big
  done > $OUT/big_map.txt 2>&1
  cat $OUT/big_map.txt
fi
if has multi; then
  MRH_BENCH_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 30 --warmup 5 --blocks 65536 > $OUT/bench_8ranks_one_device_gloo.json 2> $OUT/bench_8ranks.err
  head -c 600 $OUT/bench_8ranks_one_device_gloo.json; echo
  gcc -std=c11 -O1 -Iinclude examples/comm_smoke.c -o /tmp/comm_smoke -Lmrhash_amd/csrc -lmrhash_hip -Wl,-rpath,$PWD/mrhash_amd/csrc -lm
  MRH_COMM_SELF_LOOP=1 /tmp/comm_smoke 1 > $OUT/comm_smoke_1rank_self_loop.txt 2>&1; tail -3 $OUT/comm_smoke_1rank_self_loop.txt
  # one rank through the N-rank function over RCCL: its value is the N = 1 line's
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --multi --steps 20 --warmup 5 > $OUT/bench_1rank_rccl_multi.json 2> $OUT/bench_1rank.err
  python -c "import json; d=json.load(open('$OUT/bench_1rank_rccl_multi.json')); print('one rank through bench_multi: value', round(d['value']), d['value_definition'][:80])"
fi
rm -f $OUT/*.log
