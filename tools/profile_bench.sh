#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box through gpurun).  Counters are collected in
# their own runs, separate from the kernel-trace/stats run (MI355X_MICROARCH.md, rocprofv3 section).
#   usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 100 --warmup 10 --no-cpu --no-pmc}  # bench.py runs its own PMC sub-processes: never nest them under rocprofv3
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
tail -1 $OUT/bench_trace.log | cut -c1-400
cat $OUT/trace/t_kernel_stats.csv | head -14
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -o p -- python bench.py $ARGS > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python bench.py $ARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_write -o p -- python bench.py $ARGS > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o p -- python bench.py $ARGS > $OUT/bench_pmc_tcc.log 2>&1
python tools/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
# keep the merged-back directory small: drop the per-dispatch traces, keep stats + summary
rm -f $OUT/trace/t_kernel_trace.csv
for d in pmc_sq pmc_fetch pmc_write pmc_tcc; do rm -rf $OUT/$d; done
