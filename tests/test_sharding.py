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


def test_an_engine_keeps_its_sharding_to_itself(oracle):
    """Engine.set_sharding used to write shard_rank / shard_count into the caller's Params object, so every
    engine built from that object afterwards was created tile-sharded (bench.py's roofline and full-stream engines)."""
    p = capi.Params(num_sdf_blocks=4096, **synth.CFG1_PARAMS)
    a = capi.Engine(oracle, p)
    a.set_sharding(3, 8, 2)
    assert (p.shard_rank, p.shard_count, p.shard_chunk_log2) == (0, 1, 0)
    assert (a.params.shard_rank, a.params.shard_count, a.params.shard_chunk_log2) == (3, 8, 2)
    b = capi.Engine(oracle, p)  # a second engine from the same object: a whole map
    for e in (a, b):
        e.set_camera(synth.CFG1.fx, synth.CFG1.fy, synth.CFG1.cx, synth.CFG1.cy, synth.CFG1.rows, synth.CFG1.cols, p.min_depth, p.max_depth)
        f = synth.cfg1_sphere()
        e.set_pose(f.R, f.t)
        e.upload_depth(f.depth)
        e.upload_rgb(f.rgb)
        pending = e.integrate()
        while pending:
            pending = e.integrate_resume()
    na, nb = len(a.dump_blocks()[0]), len(b.dump_blocks()[0])
    assert 0 < na < nb and nb > 50
    a.close()
    b.close()


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
n_halo = parallel.exchange_halo(e, dist)
refused = False
try:
    e.integrate()           # halo blocks present: fusing on is refused until they are dropped
except capi.MrhError as ex:
    refused = ex.code == capi.MRH_ERR_STATE
res = parallel.gather_mesh(e, dist)
n_dropped = parallel.drop_halo(e)
n_after = len(e.dump_blocks()[0])
if rank == 0:
    tris, V, F, C = res
    np.savez({out!r}, tris=tris.view(np.uint8), V=V, F=F, C=C, n_own=n_own, n_halo=n_halo, refused=refused, n_dropped=n_dropped, n_after=n_after)
dist.barrier()
dist.destroy_process_group()
"""

# frame-sharded sub-maps (every rank fuses its own frames, owning everything) -> merge_submaps -> one tile-sharded map
MERGE_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import parity_utils as pu
from mrhash_amd import capi, parallel, synth
dist = parallel.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = capi.load_hip() if {use_hip} else pu.oracle_lib()
e = pu.make_engine(lib, synth.CFG1, dict(synth.CFG1_PARAMS), 16384)
frames = merge_frames()
for f in frames[rank::world]:
    pu.feed(e, f)
e.sync()
info = parallel.merge_submaps(e, dist, chunk_log2=1)
d, v = e.dump_blocks()
owners = parallel.owner_of_blocks(np.stack([d["x"], d["y"], d["z"]], 1), world, 1) if len(d) else np.zeros(0)
assert np.all(owners == rank), "a rank holds a block it does not own after the merge"
n_halo = parallel.exchange_halo(e, dist)
res = parallel.gather_mesh(e, dist)
parallel.drop_halo(e)
np.savez({out!r} + f".{{rank}}.npz", d=d, v=v.view(np.uint8), sent=info["sent"], received=info["received"])
if rank == 0:
    tris, V, F, C = res
    np.savez({out!r}, tris=tris.view(np.uint8), V=V, F=F, n_halo=n_halo)
dist.barrier()
dist.destroy_process_group()
"""


def merge_frames():
    return [synth.cfg1_sphere(zc=1.5 + 0.01 * k) for k in range(4)]


def run_two_ranks(tmp_path, use_hip: bool, worker: str = None, nproc: int = 2):
    out = str(tmp_path / "rank0.npz")
    script = tmp_path / "worker.py"
    script.write_text((worker or WORKER).format(root=ROOT, use_hip=use_hip, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
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
    assert bool(got["refused"]) and int(got["n_dropped"]) == int(got["n_halo"]) and int(got["n_after"]) == int(got["n_own"])


def test_four_rank_gloo_mesh_equals_single_process(oracle, tmp_path):
    """The same protocol on FOUR ranks (uneven shards, halo segments of different lengths padded to the longest, four triangle
    runs interleaved back into the canonical order): what the 8-GPU job does, at a size the CPU oracle handles."""
    got = run_two_ranks(tmp_path, use_hip=False, nproc=4)
    t, V, F, C = reference_single(oracle)
    assert int(got["n_halo"]) > 0 and int(got["n_own"]) > 0
    assert np.array_equal(got["tris"], t.view(np.uint8)), "merged triangle buffer differs from the single-process one"
    assert np.array_equal(got["F"], F) and np.array_equal(got["V"], V) and np.allclose(got["C"], C)
    assert bool(got["refused"]) and int(got["n_dropped"]) == int(got["n_halo"]) and int(got["n_after"]) == int(got["n_own"])


def check_merged_against_single(lib, tmp_path, got, world: int = 2, frames=None, min_same: float = None):
    """The merged tile-sharded map against ONE context that fused all frames.  Occupancy is the same (allocation depends on
    the depth frames only).  A voxel carries the same weight and — up to the order of the running mean, 1e-5 — the same
    TSDF value wherever every sub-map held the voxel's block while it fused; a sub-map that allocated the block late (or
    never) did not record its free-space observations of it, so there the merged weight is LOWER than the single-context
    one (never higher).  Colours are order-dependent by construction (50/50 blend) and not compared.
    The fold itself is exact: it equals the host restatement of combineVoxel over the two sub-maps, bit for bit."""
    frames = merge_frames() if frames is None else frames
    single = pu.make_engine(lib, synth.CFG1, dict(synth.CFG1_PARAMS), 16384)
    for f in frames:
        pu.feed(single, f)
    d0, v0 = single.dump_blocks()
    parts = [np.load(str(tmp_path / "rank0.npz") + f".{r}.npz") for r in range(world)]
    d = np.concatenate([p["d"] for p in parts])
    v = np.concatenate([p["v"].view(capi.VOXEL_DTYPE).reshape(-1, 512) for p in parts])
    order = np.lexsort((d["z"], d["y"], d["x"]))
    d, v = d[order], v[order]
    assert np.array_equal(d, d0), "occupancy of the merged map differs from the single-context map"
    assert np.all(v["weight"] <= v0["weight"])
    same = (v["weight"] == v0["weight"]) & (v0["weight"] > 0)
    if min_same is None:
        min_same = 0.8 if world == 2 else 0.5
    assert same.sum() > min_same * (v0["weight"] > 0).sum(), same.sum() / (v0["weight"] > 0).sum()
    assert float(np.max(np.abs(v["sdf"][same] - v0["sdf"][same]))) <= 1e-5
    assert all(int(p["sent"]) > 0 and int(p["received"]) > 0 for p in parts)
    # the fold, restated on the host from the sub-maps in rank order
    subs = []
    for r in range(world):
        e = pu.make_engine(lib, synth.CFG1, dict(synth.CFG1_PARAMS), 16384)
        for f in frames[r::world]:
            pu.feed(e, f)
        subs.append(e.dump_blocks())
    acc = {}
    for ds, vs in subs:
        for k in range(len(ds)):
            key = (int(ds["x"][k]), int(ds["y"][k]), int(ds["z"][k]))
            if key not in acc:
                acc[key] = vs[k].copy()
                continue
            a, b = acc[key], vs[k]
            w0, w1 = a["weight"].astype(np.int32), b["weight"].astype(np.int32)
            both, only_b = (w0 > 0) & (w1 > 0), (w0 == 0) & (w1 > 0)
            out = a.copy()
            out[only_b] = b[only_b]
            with np.errstate(invalid="ignore", divide="ignore"):
                s = (a["sdf"] * w0.astype(np.float32) + b["sdf"] * w1.astype(np.float32)) / (w0 + w1).astype(np.float32)
            out["sdf"][both] = s[both]
            out["sum_squared"][both] = b["sum_squared"][both]
            out["weight"][both] = np.minimum(255, w0 + w1)[both].astype(np.uint8)
            rgb = ((a["rgb"].astype(np.uint16) + b["rgb"].astype(np.uint16) + 1) >> 1).astype(np.uint8)
            out["rgb"][both] = rgb[both]
            acc[key] = out
    keys = sorted(acc)
    want = np.stack([acc[k] for k in keys])
    assert [tuple(int(d[a][i]) for a in "xyz") for i in range(len(d))] == keys
    assert np.array_equal(v.view(np.uint8), want.view(np.uint8)), "merge differs from the host restatement of combineVoxel"
    # mesh of the merged map: rank 0 gathered it; compare with the mesh of a context that imports the merged map
    ref = pu.make_engine(lib, synth.CFG1, dict(synth.CFG1_PARAMS), 16384)
    ref.import_blocks(d, v)
    t = ref.extract_triangles()
    V, F, C = ref.extract_mesh()
    assert np.array_equal(got["tris"], t.view(np.uint8)) and np.array_equal(got["F"], F) and np.array_equal(got["V"], V)
    assert int(got["n_halo"]) > 0


def test_frame_sharded_submaps_merge_into_one_tile_sharded_map(oracle, tmp_path):
    got = run_two_ranks(tmp_path, use_hip=False, worker=MERGE_WORKER.replace("merge_frames()", "[synth.cfg1_sphere(zc=1.5 + 0.01 * k) for k in range(4)]"))
    check_merged_against_single(oracle, tmp_path, got)


def test_three_rank_merge_of_uneven_submaps(oracle, tmp_path):
    """Three ranks, four frames (rank 0 fuses two): all-to-all with three different split lists, the fold in rank order."""
    got = run_two_ranks(tmp_path, use_hip=False, worker=MERGE_WORKER.replace("merge_frames()", "[synth.cfg1_sphere(zc=1.5 + 0.01 * k) for k in range(4)]"), nproc=3)
    check_merged_against_single(oracle, tmp_path, got, world=3)


def test_eight_rank_gloo_mesh_equals_single_process(oracle, tmp_path):
    """World size EIGHT — BASELINE.json configs[3]'s rank count: the 8-way owner function, eight halo segments of different
    lengths, eight triangle runs interleaved back into the canonical order, the two starve reductions over eight buffers."""
    got = run_two_ranks(tmp_path, use_hip=False, nproc=8)
    t, V, F, C = reference_single(oracle)
    assert int(got["n_halo"]) > 0 and int(got["n_own"]) > 0
    assert np.array_equal(got["tris"], t.view(np.uint8)), "merged triangle buffer differs from the single-process one"
    assert np.array_equal(got["F"], F) and np.array_equal(got["V"], V) and np.allclose(got["C"], C)
    assert bool(got["refused"]) and int(got["n_dropped"]) == int(got["n_halo"]) and int(got["n_after"]) == int(got["n_own"])


def test_eight_rank_merge_and_fold(oracle, tmp_path):
    """Eight frame-sharded sub-maps (one frame each) -> merge_submaps: eight split lists per rank, the fold of up to eight
    sub-maps per block in rank order equals the host restatement bit for bit; halo exchange + mesh gather on the result."""
    frames_src = "[synth.cfg1_sphere(zc=1.5 + 0.005 * k) for k in range(8)]"
    got = run_two_ranks(tmp_path, use_hip=False, worker=MERGE_WORKER.replace("merge_frames()", frames_src), nproc=8)
    check_merged_against_single(oracle, tmp_path, got, world=8, frames=[synth.cfg1_sphere(zc=1.5 + 0.005 * k) for k in range(8)], min_same=0.3)


def _multires_submaps(lib, make=None):
    """Two variance-adaptive sub-maps of the 128x128 sphere whose resolutions disagree both ways round (8 positions coarse in the
    first and fine in the second, 13 the other way)."""
    p = dict(synth.CFG1_PARAMS, sdf_var_threshold=0.5, n_frames_invalidate_voxels=3)
    a = pu.make_engine(lib, synth.CFG1, p, 16384)
    b = pu.make_engine(lib, synth.CFG1, p, 16384)
    for f in (synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.52), synth.cfg1_sphere(zc=1.49), synth.cfg1_sphere()):
        pu.feed(a, f)
    for f in (synth.cfg1_sphere(zc=1.55), synth.cfg1_sphere(zc=1.56), synth.cfg1_sphere(zc=1.55)):
        pu.feed(b, f)
    return a, b


def _records_of(e):
    """every block of the engine as mrh_block_record[] (host copy), sorted by position"""
    import ctypes

    from mrhash_amd import hipmem

    e.set_sharding(0, 1, 0)
    ptr, n, on_dev = e.pack_blocks(capi.PACK_OWNER, 0)
    if on_dev:
        raw = np.frombuffer(hipmem.read(ptr, n * capi.RECORD_BYTES), dtype=np.uint8).copy()
    else:
        raw = np.frombuffer((ctypes.c_char * (n * capi.RECORD_BYTES)).from_address(ptr), dtype=np.uint8).copy()
    r = raw.view(capi.RECORD_DTYPE)
    return r[np.lexsort((r["desc"]["z"], r["desc"]["y"], r["desc"]["x"]))]


def check_mixed_resolution_merge(lib):
    """VERDICT r05 missing-3: merging variance-adaptive sub-maps.  Rule (include/mrhash_hip.h, mrh_unpack_mode): a position that
    is fine in one sub-map and coarse in the other ends up coarse, with the coarse side's payload; same-resolution positions
    merge voxel by voxel; everything else is inserted.  The result must not depend on which sub-map is folded into which."""
    a, b = _multires_submaps(lib)
    ra, rb = _records_of(a), _records_of(b)
    key = lambda r: list(zip(r["desc"]["x"].tolist(), r["desc"]["y"].tolist(), r["desc"]["z"].tolist()))  # noqa: E731
    res_a, res_b = dict(zip(key(ra), ra["desc"]["resolution"].tolist())), dict(zip(key(rb), rb["desc"]["resolution"].tolist()))
    both = set(res_a) & set(res_b)
    mixed = [k for k in both if res_a[k] != res_b[k]]
    assert len(mixed) >= 10 and len(both) >= 30, (len(mixed), len(both))
    assert any(res_a[k] for k in mixed) and any(res_b[k] for k in mixed)  # both directions occur
    taken_ab = b.unpack_blocks(capi.UNPACK_MERGE, ra.ctypes.data, len(ra), False)  # a into b
    taken_ba = a.unpack_blocks(capi.UNPACK_MERGE, rb.ctypes.data, len(rb), False)  # b into a
    fine_onto_coarse_ab = sum(1 for k in mixed if res_a[k] == 0)  # a's fine records that met a coarse block of b: dropped
    fine_onto_coarse_ba = sum(1 for k in mixed if res_b[k] == 0)
    assert taken_ab == len(ra) - fine_onto_coarse_ab and taken_ba == len(rb) - fine_onto_coarse_ba
    (da, va), (db, vb) = a.dump_blocks(), b.dump_blocks()
    assert np.array_equal(da, db) and len(da) == len(set(res_a) | set(res_b))
    got = dict(zip(list(zip(da["x"].tolist(), da["y"].tolist(), da["z"].tolist())), da["resolution"].tolist()))
    for k, r in got.items():
        assert r == max(res_a.get(k, 0), res_b.get(k, 0)), k  # coarse wherever either side was coarse
    # the payload is the same both ways round (sum_squared aside: it is the later sub-map's term, as in the single-resolution merge)
    bits = lambda x: np.ascontiguousarray(x).view(np.uint8)  # noqa: E731
    assert np.array_equal(va["weight"], vb["weight"])
    seen = va["weight"] > 0  # a voxel neither side observed keeps whatever the receiving side held
    assert seen.sum() > 1000 and np.array_equal(va["sdf"][seen].view(np.uint32), vb["sdf"][seen].view(np.uint32))
    assert np.array_equal(va["rgb"][seen], vb["rgb"][seen])
    # a mixed position carries the coarse side's voxels, untouched
    idx = {k: i for i, k in enumerate(zip(da["x"].tolist(), da["y"].tolist(), da["z"].tolist()))}
    src = {True: (ra, dict(zip(key(ra), range(len(ra))))), False: (rb, dict(zip(key(rb), range(len(rb)))))}
    for k in mixed[:40]:
        recs, where = src[res_a[k] == 1]
        want = recs["voxels"][where[k]][:64]
        assert np.array_equal(bits(va[idx[k]][:64]["sdf"]), bits(want["sdf"])) and np.array_equal(va[idx[k]][:64]["weight"], want["weight"])
    assert a.stats().error_flags == 0 and b.stats().error_flags == 0
    # the merged maps keep fusing (the frame after a bulk change goes through the general kernels); the two contexts have different
    # frame counters, so their starve frames differ from here on: only that each of them carries on cleanly
    n_before = len(da)
    for e in (a, b):
        pu.feed(e, synth.cfg1_sphere(zc=1.5))
        assert e.stats().error_flags == 0 and len(e.dump_blocks()[0]) >= n_before - 40
    return a, b


def test_merging_variance_adaptive_submaps_coarse_wins(oracle):
    a, b = check_mixed_resolution_merge(oracle)
    a.close()
    b.close()


def test_two_rank_merge_of_variance_adaptive_submaps(oracle, tmp_path):
    """The whole frame-sharded protocol (merge_submaps -> exchange_halo -> gather_mesh) on two gloo ranks whose sub-maps are
    variance-adaptive and disagree about resolutions: every rank ends up owning exactly its tiles, a position is coarse in the merged
    map iff it was coarse in a sub-map, and the mesh over the merged shards is the mesh a single context extracts from the same
    merged blocks."""
    worker = (MERGE_WORKER
              .replace("dict(synth.CFG1_PARAMS)", "dict(synth.CFG1_PARAMS, sdf_var_threshold=0.5, n_frames_invalidate_voxels=3)")
              .replace("for f in frames[rank::world]:", "for f in frames[3 * rank: 3 * rank + (4 if rank == 0 else 3)]:")
              .replace("merge_frames()", "[synth.cfg1_sphere(zc=z) for z in (1.5, 1.52, 1.49, 1.5, 1.55, 1.56, 1.55)]"))
    # (rank 0: frames 0-3, rank 1: frames 3-5 of the list above, i.e. zc 1.5, 1.55, 1.56 -> two different sub-maps)
    got = run_two_ranks(tmp_path, use_hip=False, worker=worker)
    parts = [np.load(str(tmp_path / "rank0.npz") + f".{r}.npz") for r in range(2)]
    d = np.concatenate([p["d"] for p in parts])
    assert len(d) > 60 and len(np.unique(d[["x", "y", "z"]])) == len(d)  # one owner per position
    assert 0 < int((d["resolution"] != 0).sum()) < len(d)
    assert all(int(p["sent"]) > 0 and int(p["received"]) > 0 for p in parts)
    # the gathered mesh == the mesh of ONE context holding the merged blocks
    v = np.concatenate([p["v"] for p in parts]).view(capi.VOXEL_DTYPE).reshape(len(d), 512)
    single = pu.make_engine(oracle, synth.CFG1, dict(synth.CFG1_PARAMS, sdf_var_threshold=0.5, n_frames_invalidate_voxels=3), 16384)
    single.import_blocks(d, v)
    t = single.extract_triangles()
    assert len(t) > 300 and np.array_equal(got["tris"], t.view(np.uint8))


def test_pack_unpack_drop_single_process(oracle):
    """The exchange primitives without a process group: pack by owner, empty the map, fold the records back."""
    import ctypes

    e = pu.make_engine(oracle, synth.CFG1, dict(synth.CFG1_PARAMS), 16384)
    pu.feed(e, synth.cfg1_sphere())
    d0, v0 = e.dump_blocks()
    e.set_sharding(0, 2, 1)
    recs = []
    for dest in range(2):
        ptr, n, on_dev = e.pack_blocks(capi.PACK_OWNER, dest)
        assert not on_dev
        recs.append(np.frombuffer((ctypes.c_char * (n * capi.RECORD_BYTES)).from_address(ptr), dtype=capi.RECORD_DTYPE).copy() if n else np.zeros(0, capi.RECORD_DTYPE))
    assert len(recs[0]) + len(recs[1]) == len(d0) and min(len(r) for r in recs) > 0
    ptr, n, _ = e.pack_blocks(capi.PACK_HALO)
    halo = np.frombuffer((ctypes.c_char * (n * capi.RECORD_BYTES)).from_address(ptr), dtype=capi.RECORD_DTYPE).copy()
    own0 = recs[0]["desc"]
    assert np.array_equal(np.sort(halo["desc"], order=["x", "y", "z"]), np.sort(own0[parallel.boundary_mask(own0, 1)], order=["x", "y", "z"]))
    assert e.drop_blocks(capi.DROP_FOREIGN) == len(recs[1])
    assert e.drop_blocks(capi.DROP_ALL) == len(recs[0])
    assert len(e.dump_blocks()[0]) == 0
    e.set_sharding(0, 1, 0)
    for r in recs:
        assert e.unpack_blocks(capi.UNPACK_MERGE, r.ctypes.data, len(r), False) == len(r)
    d1, v1 = e.dump_blocks()
    assert np.array_equal(d0, d1) and np.array_equal(v0.view(np.uint8), v1.view(np.uint8))
