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


def test_rendezvous_ignores_a_stale_id_file_and_keeps_its_files_private(tmp_path, monkeypatch):
    """A file of the right name and size that is OLDER than the launcher (a crashed earlier run, a reused key) must not be taken
    for this run's ncclUniqueId — a rank that believed it would sit in ncclCommInitRank for ever; what is published is private
    (0600) and the default directory is per-user (0700)."""
    import pytest

    from mrhash_amd import capi, parallel

    monkeypatch.setenv("MRH_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("MRH_RDZV_KEY", "pytest_stale")
    stale = tmp_path / "mrh_rdzv_pytest_stale.id"
    stale.write_bytes(b"\x01" * capi.COMM_ID_BYTES)
    os.utime(stale, (1.0, 1.0))  # 1970: older than any launcher
    with pytest.raises(TimeoutError):
        parallel.rendezvous(None, rank=1, world=2, device_id=0, timeout_s=0.3)
    parallel.publish_file(str(tmp_path / "x"), b"abc")
    assert (os.stat(tmp_path / "x").st_mode & 0o777) == 0o600 and (tmp_path / "x").read_bytes() == b"abc"
    monkeypatch.delenv("MRH_RDZV_DIR")
    d = parallel.rdzv_dir()
    assert (os.stat(d).st_mode & 0o777) == 0o700 and os.stat(d).st_uid == os.getuid()
    assert parallel.launcher_start_time() > 1e9
