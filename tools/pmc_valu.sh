#!/bin/bash
# VALU instruction / activity counters of the bench kernels (own rocprofv3 pass, no tracing).  usage: pmc_valu.sh <tag> [lib.so]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
LIB=${2:-mrhash_amd/csrc/libmrhash_hip.so}
OUT=gpurun_out/pmc_valu_${1:-x}; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o p -- python tools/bench_with_lib.py $LIB --steps 60 --warmup 10 --no-cpu > $OUT/log.txt 2>&1
python tools/summarize_pmc.py $OUT | grep -A1 -E "k_back<true, false, false>|k_front<false, false>"
rm -rf $OUT/pmc_sq
