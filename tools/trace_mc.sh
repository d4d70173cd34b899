#!/bin/bash
# build the tracing variant of the library (never shipped: -DMRH_MC_TRACE) and print where a k_mc workgroup's time goes per block class
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -DMRH_MC_TRACE -Iinclude -o mrhash_amd/csrc/libmrhash_trace.so mrhash_amd/csrc/mrh_capi.hip -ldl
python tools/trace_mc.py "$@"
