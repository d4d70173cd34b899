"""Multi-GPU path on the CPU: two processes over gloo (world_size 2), each a tile-sharded context, all-gather of
boundary blocks, per-rank extraction, merge on rank 0 — compared with a single-process run of the same frames.
The compute engine in this file is the oracle (test infrastructure); tests/test_sharding_gpu.py runs the same
protocol with the HIP library."""
import os
import subprocess
import sys

import numpy as np
import pytest

import parity_utils as pu
from mrhash_amd import capi, parallel, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_owner_function_is_a_partition():
    rng = np.random.default_rng(0)
    xyz = rng.integers(-200, 200, size=(5000, 3))
    for world in (2, 4, 8):
        o = parallel.owner_of_blocks(xyz, world, 3)
        assert o.min() >= 0 and o.max() < world
        # whole chunks move together
        same_chunk = parallel.owner_of_blocks((xyz >> 3) << 3, world, 3)
        assert np.array_equal(o, same_chunk)
        counts = np.bincount(o, minlength=world)
        assert counts.min() > 0.5 * len(xyz) / world
    assert np.all(parallel.owner_of_blocks(xyz, 1) == 0)


def test_frame_sharding_covers_the_stream():
    got = sorted(i for r in range(4) for i in parallel.shard_frames(25, r, 4))
    assert got == list(range(100))


def test_union_of_tile_shards_equals_single_map(oracle):
    """No communication needed for the map itself: shard r keeps exactly the blocks it owns."""
    frames = [synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.51)]
    params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2)
    single = pu.make_engine(oracle, synth.CFG1, params, 16384)
    shards = [pu.make_engine(oracle, synth.CFG1, params, 16384, shard_rank=r, shard_count=2, shard_chunk_log2=1) for r in range(2)]
    for f in frames:
        for e in [single] + shards:
            pu.feed(e, f)
    d0, v0 = single.dump_blocks()
    parts = [s.dump_blocks() for s in shards]
    for r, (d, _) in enumerate(parts):
        assert np.all(parallel.owner_of_blocks(np.stack([d["x"], d["y"], d["z"]], 1), 2, 1) == r)
    d = np.concatenate([p[0] for p in parts])
    v = np.concatenate([p[1] for p in parts])
    order = np.lexsort((d["z"], d["y"], d["x"]))
    assert np.array_equal(d[order], d0) and np.array_equal(v[order].view(np.uint8), v0.view(np.uint8))
    assert min(len(p[0]) for p in parts) > 0


def test_import_blocks_roundtrip(oracle):
    a = pu.make_engine(oracle, synth.CFG1, synth.CFG1_PARAMS, 8192)
    pu.feed(a, synth.cfg1_sphere())
    d, v = a.dump_blocks()
    b = pu.make_engine(oracle, synth.CFG1, synth.CFG1_PARAMS, 8192)
    b.import_blocks(d, v)
    d2, v2 = b.dump_blocks()
    assert np.array_equal(d, d2) and np.array_equal(v.view(np.uint8), v2.view(np.uint8))
    ta, tb = a.extract_triangles(), b.extract_triangles()
    assert np.array_equal(ta.view(np.uint8), tb.view(np.uint8))


WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import parity_utils as pu
from mrhash_amd import capi, parallel, synth
dist = parallel.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = capi.load_hip() if {use_hip} else pu.oracle_lib()
params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2)
e = pu.make_engine(lib, synth.CFG1, params, 16384, shard_rank=rank, shard_count=world, shard_chunk_log2=1)
for f in (synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.51), synth.cfg1_sphere(zc=1.5), synth.cfg1_sphere(zc=1.49)):
    pu.feed(e, f, dist=dist)   # frames 2 and 3... are starve frames: z-buffer MIN all-reduce over the ranks
e.sync()
n_own = len(e.dump_blocks()[0])
n_halo = parallel.exchange_halo(e, dist, chunk_log2=1)
res = parallel.gather_mesh(e, dist)
if rank == 0:
    tris, V, F, C = res
    np.savez({out!r}, tris=tris.view(np.uint8), V=V, F=F, C=C, n_own=n_own, n_halo=n_halo)
dist.barrier()
dist.destroy_process_group()
"""


def run_two_ranks(tmp_path, use_hip: bool):
    out = str(tmp_path / "rank0.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, use_hip=use_hip, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


def reference_single(lib):
    params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2)
    e = pu.make_engine(lib, synth.CFG1, params, 16384)
    for f in (synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.51), synth.cfg1_sphere(zc=1.5), synth.cfg1_sphere(zc=1.49)):
        pu.feed(e, f)
    t = e.extract_triangles()
    V, F, C = e.extract_mesh()
    return t, V, F, C


def test_two_rank_gloo_mesh_equals_single_process(oracle, tmp_path):
    got = run_two_ranks(tmp_path, use_hip=False)
    t, V, F, C = reference_single(oracle)
    assert int(got["n_halo"]) > 0 and int(got["n_own"]) > 0
    assert np.array_equal(got["tris"], t.view(np.uint8)), "merged triangle buffer differs from the single-process one"
    assert np.array_equal(got["F"], F) and np.array_equal(got["V"], V) and np.allclose(got["C"], C)
