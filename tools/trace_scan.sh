#!/bin/bash
# build the tracing variant of the library (never shipped: -DMRH_SCAN_TRACE) and print where the workgroups of a LiDAR scan's
# kernels spend their time (wall-clock stamps of thread 0 at the phase boundaries, relative to the kernel's first workgroup)
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
  -DMRH_SCAN_TRACE -Iinclude -o mrhash_amd/csrc/libmrhash_trace.so mrhash_amd/csrc/mrh_capi.hip -ldl
python tools/trace_scan.py "$@"
