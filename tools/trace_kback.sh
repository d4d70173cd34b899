#!/bin/bash
# build the tracing variant of the library (never shipped: -DMRH_TRACE) and print the phase timeline of k_back
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -DMRH_TRACE -Iinclude -o mrhash_amd/csrc/libmrhash_trace.so mrhash_amd/csrc/mrh_capi.hip
python tools/trace_kback.py "$@"
