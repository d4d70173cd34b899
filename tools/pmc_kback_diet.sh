#!/bin/bash
# VERDICT r05 next-2, the counter-backed account: k_back ALONE on the chip (MRH_PIPE=0, one generation of workgroups), the shipped
# kernel (121 VGPRs, 4 waves a SIMD) against the block-in-two-halves variant at 4 waves (109 VGPRs) and capped at 96 VGPRs with the
# grid of 5 waves a SIMD (profiles/r06/kback_variants_measured_and_dropped.patch, built as libmrhash_seq4/5.so).  Two passes of
# eight SQ counters each per build + the kernel-trace time.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/diet; rm -rf $OUT; mkdir -p $OUT
A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
B="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
for spec in "hip 1024" "seq4 1024" "seq5 1280"; do
  set -- $spec; n=$1; g=$2
  for p in A B; do
    ctr=$A; [ $p = B ] && ctr=$B
    MRH_PIPE=0 MRH_FUSED_GRID=$g rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${n}_$p/pmc_sq -o p -- python tools/bench_with_lib.py mrhash_amd/csrc/libmrhash_$n.so --pmc-inner --steps 20 --warmup 5 > $OUT/log_${n}_$p.txt 2>&1
    echo "## $n grid $g pass $p"; python tools/summarize_pmc.py $OUT/pmc_${n}_$p | grep -A1 "k_back<true, false, false, false, 0, false>"
    rm -rf $OUT/pmc_${n}_$p
  done
  MRH_PIPE=0 MRH_FUSED_GRID=$g rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$n -o t -- python tools/bench_with_lib.py mrhash_amd/csrc/libmrhash_$n.so --pmc-inner --steps 20 --warmup 5 > $OUT/log_tr_$n.txt 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open("$OUT/tr_$n/t_kernel_stats.csv")):
    if "k_back<" in r["Name"]: print("## $n grid $g kernel-trace: k_back calls", r["Calls"], "avg %.2f us" % (float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/tr_$n
done
rm -f $OUT/log_*.txt
