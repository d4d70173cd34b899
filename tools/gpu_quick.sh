#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q
gcc -std=c11 -O1 -Wall -Iinclude examples/comm_smoke.c -o /tmp/comm_smoke -Lmrhash_amd/csrc -lmrhash_hip -Wl,-rpath,$PWD/mrhash_amd/csrc -lm
/tmp/comm_smoke 1 2>&1 | tail -12
python tools/exp_two_engines.py 60 2>&1 | tail -6
