"""One-off soak on the GPU box: the randomised parity tests with many more seeds than the test suite runs.
usage: python tests/soak.py [first_seed] [n]
       python tests/soak.py churn [frames]      the 3 000-frame GC + paging walk of tests/test_churn_gpu.py (12 minutes)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import parity_utils as pu  # noqa: E402
import test_lidar_gpu as tl  # noqa: E402
import test_parity_gpu as tp  # noqa: E402
from mrhash_amd import capi  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "churn":
    import test_churn_gpu as tc

    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    t0 = time.time()
    line = tc.walk_with_gc_and_paging(capi.load_hip(), pu.oracle_lib(), frames=frames, pool=tc.POOL, min_paged=50000 * frames // 3000,
                                      min_rehashes=5 * frames // 3000)
    print(f"soak churn: OK in {time.time() - t0:.0f} s")
    sys.exit(0)

first = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
hip, orc = capi.load_hip(), pu.oracle_lib()
t0 = time.time()
bad = 0
for seed in range(first, first + n):
    for name, fn in (("rgbd", tp.test_randomised_parameters_and_shapes), ("lidar", tl.test_randomised_scans)):
        try:
            fn(hip, orc, seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"seed {seed} {name}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)
    if (seed - first) % 10 == 9:
        print(f"... {seed - first + 1} seeds, {bad} failures, {time.time() - t0:.0f} s", flush=True)
print(f"soak: {n} seeds x 2 fuzzers, {bad} failures, {time.time() - t0:.0f} s")
