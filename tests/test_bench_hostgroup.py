"""bench.py's file-based host group (the timing fallback of `--gpus N` when the RCCL communicator cannot be created): barrier,
all-gather and max over three processes, on the CPU."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, {root!r})
    import bench
    r, n = int(sys.argv[1]), int(sys.argv[2])
    g = bench.HostGroup(r, n)
    g.barrier()
    a = g.allgather_f64([r * 1.5, 7])
    g.barrier()
    b = g.allgather_f64([float(r == n - 1)])
    g.close()
    print(json.dumps({{"rank": r, "a": a.tolist(), "max": float(b.max())}}), flush=True)
""")


def test_host_group_barrier_allgather_and_cleanup(tmp_path):
    n = 3
    env = dict(os.environ, MRH_RDZV_KEY="pytest_hostgroup", MRH_RDZV_DIR=str(tmp_path))
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(n)], env=env, stdout=subprocess.PIPE, text=True) for r in range(n)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    import json

    for r, o in enumerate(outs):
        d = json.loads(o.strip().splitlines()[-1])
        assert d["rank"] == r and d["a"] == [[0.0, 7.0], [1.5, 7.0], [3.0, 7.0]] and d["max"] == 1.0
    # every step's files are gone except the last (empty) ones
    left = sorted(os.listdir(tmp_path))
    assert [f for f in left if f.startswith("mrh_hostgrp_") and not f.split(".")[-2] == "5"] == [], left
