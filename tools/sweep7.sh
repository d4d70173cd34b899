#!/bin/bash
for sw in 32 128 512 1024; do echo "SWEEP_WGS $sw"; MRH_SWEEP_WGS=$sw bash tools/rocprof_stats.sh sw$sw 2>&1 | grep -E "fps|k_front<false>|k_back<true, false>"; done
