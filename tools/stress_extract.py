"""Many extractions of one multi-resolution map, with and without busy host threads competing for the cores the widening helpers
run on: every extraction must succeed and give the same mesh.  usage: python tools/stress_extract.py [extractions] [busy threads]"""
import hashlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mrhash_amd import capi, synth

n_ext = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_busy = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if os.environ.get("MRH_LIB"):  # a tuning build of the library (mrhash_amd/build.py variant ...)
    capi.HIP_LIB_PATH = os.path.abspath(os.environ["MRH_LIB"])
hip = capi.load_hip()
K = synth.REPLICA_640
frames = bench.render_stream("replica", 25)
res = bench.Resident(frames, K)
e = bench.make_engine(hip, capi.Params(num_sdf_blocks=262144, device_id=0, **dict(synth.REPLICA_PARAMS, sdf_var_threshold=0.005)), K)
res.run(e, 0, 25)
e.sync()
stop = False


def burn():
    x = 0
    while not stop:
        x = (x * 1103515245 + 12345) & 0x7FFFFFFF


if n_busy:  # numpy-free spinning threads only contend through the GIL; real contention comes from processes
    import multiprocessing as mp
    procs = [mp.Process(target=burn, daemon=True) for _ in range(n_busy)]
    for p in procs:
        p.start()
ref = None
ts = []
for i in range(n_ext):
    t0 = time.perf_counter()
    n = e.extract_triangles(soup=False)
    ts.append((time.perf_counter() - t0) * 1e3)
    if i % 50 == 0 or i == n_ext - 1:
        V, F, C = e.extract_mesh()
        h = hashlib.sha256(V.tobytes() + F.tobytes() + C.tobytes()).hexdigest()
        if ref is None:
            ref = h
        assert h == ref, f"extraction {i}: mesh differs"
if n_busy:
    for p in procs:
        p.terminate()
e.close()  # MRH_WIDEN_REPORT=1: the library says how many chunks the calling thread redid
ts = np.array(ts[3:])
print(f"stress_extract: {n_ext} extractions, {n_busy} busy processes, {n} triangles, 0 failures; ms min {ts.min():.3f} median {np.median(ts):.3f} p99 {np.percentile(ts, 99):.3f} max {ts.max():.3f}")
