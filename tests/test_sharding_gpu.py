"""Tile-sharded path with the HIP library: two processes (gloo rendezvous, both on the one GPU of the test box),
halo all-gather, merged mesh == single-context mesh, bit for bit."""
import numpy as np
import pytest

import parity_utils as pu
from mrhash_amd import parallel, synth
from test_sharding import MERGE_WORKER, check_merged_against_single, reference_single, run_two_ranks

pytestmark = pytest.mark.gpu


def test_union_of_tile_shards_equals_single_map_hip(hip):
    frames = [synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.51)]
    params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2)
    single = pu.make_engine(hip, synth.CFG1, params, 16384)
    shards = [pu.make_engine(hip, synth.CFG1, params, 16384, shard_rank=r, shard_count=2, shard_chunk_log2=1) for r in range(2)]
    for f in frames:
        for e in [single] + shards:
            pu.feed(e, f)
    d0, v0 = single.dump_blocks()
    parts = [s.dump_blocks() for s in shards]
    d = np.concatenate([p[0] for p in parts])
    v = np.concatenate([p[1] for p in parts])
    order = np.lexsort((d["z"], d["y"], d["x"]))
    assert np.array_equal(d[order], d0) and np.array_equal(v[order].view(np.uint8), v0.view(np.uint8))


def _union(shards):
    parts = [s.dump_blocks() for s in shards]
    d = np.concatenate([p[0] for p in parts])
    v = np.concatenate([p[1] for p in parts])
    order = np.lexsort((d["z"], d["y"], d["x"]))
    return d[order], v[order]


@pytest.mark.parametrize("count,chunk_log2,var", [(8, 3, 0.0), (4, 2, 0.0), (3, 3, 0.005)])
def test_shards_that_skip_foreign_tiles_still_build_the_single_map_hip(hip, count, chunk_log2, var):
    """640x480 Replica stand-in on `count` tile shards (the fast path; with `var` the multi-resolution one): a shard skips the
    ray walk of every 16x16 pixel tile whose rays cannot reach one of its chunks (tile_reaches_owned_chunk).  The union of
    the shard maps must still be the single-context map, block for block and voxel for voxel, after a few frames of the
    orbit with garbage collection on — a tile skipped wrongly would show up as a missing block."""
    params = dict(synth.REPLICA_PARAMS, sdf_var_threshold=var)
    single = pu.make_engine(hip, synth.REPLICA_640, params, 65536)
    shards = [pu.make_engine(hip, synth.REPLICA_640, params, 65536, shard_rank=r, shard_count=count, shard_chunk_log2=chunk_log2)
              for r in range(count)]
    for f in synth.replica_stream(6):
        for e in [single] + shards:
            pu.feed(e, f)
    d0, v0 = single.dump_blocks()
    d, v = _union(shards)
    assert len(d0) > 5000 and np.array_equal(d, d0) and np.array_equal(v.view(np.uint8), v0.view(np.uint8))
    assert all(s.stats().error_flags == 0 for s in shards)
    owned = [int(s.stats().occupied_fine + s.stats().occupied_coarse) for s in shards]
    assert min(owned) > 0 and sum(owned) == len(d0)


def test_import_blocks_roundtrip_hip(hip):
    a = pu.make_engine(hip, synth.CFG1, dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=1000), 8192)  # GC every frame, no starve
    pu.feed(a, synth.cfg1_sphere())
    d, v = a.dump_blocks()
    b = pu.make_engine(hip, synth.CFG1, dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=1000), 8192)
    b.import_blocks(d, v)
    d2, v2 = b.dump_blocks()
    assert np.array_equal(d, d2) and np.array_equal(v.view(np.uint8), v2.view(np.uint8))
    assert np.array_equal(a.extract_triangles().view(np.uint8), b.extract_triangles().view(np.uint8))
    # an imported map keeps fusing exactly like the original (GC summaries were rebuilt on import)
    f = synth.cfg1_sphere(zc=1.52)
    pu.feed(a, f); pu.feed(b, f)
    da, va = a.dump_blocks(); db, vb = b.dump_blocks()
    assert np.array_equal(da, db) and np.array_equal(va.view(np.uint8), vb.view(np.uint8))


def test_import_of_coarse_blocks_into_a_fresh_context_hip(hip, oracle):
    """A variance-adaptive map restored into a context that has not fused a frame yet: its coarse free list is empty until a frame's
    refill (vds.cu:885-891), so the import sizes it first (round 6; the oracle's import used to spin on the empty list, the device's
    to drop the blocks with a pool error)."""
    from test_sharding import _multires_submaps

    a, _b = _multires_submaps(hip)
    d, v = a.dump_blocks()
    assert 0 < int((d["resolution"] != 0).sum()) < len(d)
    p = dict(synth.CFG1_PARAMS, sdf_var_threshold=0.5, n_frames_invalidate_voxels=3)
    for lib in (hip, oracle):
        e = pu.make_engine(lib, synth.CFG1, p, 16384)
        e.import_blocks(d, v)
        d2, v2 = e.dump_blocks()
        assert np.array_equal(d, d2) and np.array_equal(v.view(np.uint8), v2.view(np.uint8)) and e.stats().error_flags == 0
        assert np.array_equal(a.extract_triangles().view(np.uint8), e.extract_triangles().view(np.uint8))
        e.close()
    a.close()
    _b.close()


def test_two_rank_mesh_equals_single_context_hip(hip, tmp_path):
    got = run_two_ranks(tmp_path, use_hip=True)
    t, V, F, C = reference_single(hip)
    assert int(got["n_halo"]) > 0
    assert np.array_equal(got["tris"], t.view(np.uint8))
    assert np.array_equal(got["F"], F) and np.array_equal(got["V"], V)
    assert bool(got["refused"]) and int(got["n_dropped"]) == int(got["n_halo"]) and int(got["n_after"]) == int(got["n_own"])


def test_frame_sharded_submaps_merge_into_one_tile_sharded_map_hip(hip, tmp_path):
    """Two HIP ranks fuse disjoint halves of the stream, merge_submaps folds them into one tile-sharded map (device
    pack -> collective -> device unpack), halo exchange + mesh gather follow."""
    got = run_two_ranks(tmp_path, use_hip=True, worker=MERGE_WORKER.replace("merge_frames()", "[synth.cfg1_sphere(zc=1.5 + 0.01 * k) for k in range(4)]"))
    check_merged_against_single(hip, tmp_path, got)


RCCL_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import parity_utils as pu
from mrhash_amd import capi, parallel, synth
hip = capi.load_hip()
comm = parallel.rendezvous(hip)        # RCCL behind the C ABI: mrh_comm_unique_id -> file -> mrh_comm_create; one rank here
assert (comm.rank, comm.world) == (0, 1)
comm.barrier()
assert float(comm.allreduce([3.5], capi.COMM_MAX)[0]) == 3.5 and comm.allgather_i64([7, 9]).tolist() == [[7, 9]]
params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2)
frames = (synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.51), synth.cfg1_sphere(zc=1.5), synth.cfg1_sphere(zc=1.49))
# (1) a tile-sharded context with the communicator attached: starve frames run ncclAllReduce(int64, MIN) on the library's
#     z-buffer inside mrh_integrate (never MRH_PENDING_EXCHANGE); reference: the same frames with the host protocol and the
#     exchange skipped (one rank: MIN over one buffer is the buffer)
a = pu.make_engine(hip, synth.CFG1, params, 16384, shard_rank=0, shard_count=1)
a.attach_comm(comm)
a.set_sharding(0, 1, 1)
b = pu.make_engine(hip, synth.CFG1, params, 16384, shard_rank=0, shard_count=1)
# force the sharded code path on one rank: shard_count 1 owns everything, so run a second pair that owns one of two shards
a2 = pu.make_engine(hip, synth.CFG1, params, 16384, shard_rank=0, shard_count=2, shard_chunk_log2=1)
a2.attach_comm(comm)                   # world 1 != shard_count 2: a reduction over the wrong set of ranks would silently give
b2 = pu.make_engine(hip, synth.CFG1, params, 16384, shard_rank=0, shard_count=2, shard_chunk_log2=1)  # another map, so it is refused ...
a2.set_pose(frames[0].R, frames[0].t); a2.upload_depth(frames[0].depth); a2.upload_rgb(frames[0].rgb)
try:
    a2.integrate()
    raise SystemExit("a communicator whose ranks are not the map's shards was accepted")
except capi.MrhError as e:
    assert "shard 0 of 2" in str(e) and "rank 0 of 1" in str(e), str(e)
os.environ["MRH_COMM_ALLOW_SHARD_MISMATCH"] = "1"   # ... unless this test hook says otherwise (one-GPU box: one rank reduces its own buffer)
for f in frames:
    pu.feed(a, f); pu.feed(b, f)
    pu.feed(a2, f)                      # integrate() must not report a pending exchange
    b2.set_pose(f.R, f.t); b2.upload_depth(f.depth); b2.upload_rgb(f.rgb)
    pending = b2.integrate()
    while pending:
        pending = b2.integrate_resume()
pu.compare_maps(a, b)
pu.compare_maps(a2, b2)
ph = a2.comm_phase_times()
assert ph["allreduce_count"] == 2 and ph["allreduce_ms_sum"] > 0, ph   # one starve frame (frame 2 of 0..3), two reductions
# (2) mrh_comm_merge_submaps: a one-rank fold reproduces the map
c = pu.make_engine(hip, synth.CFG1, dict(synth.CFG1_PARAMS), 16384)
c.attach_comm(comm)
for f in frames:
    pu.feed(c, f)
d0, v0 = c.dump_blocks()
info = parallel.merge_submaps(c, comm, chunk_log2=1)
d1, v1 = c.dump_blocks()
assert len(d0) > 50 and np.array_equal(d0, d1) and np.array_equal(v0.view(np.uint8), v1.view(np.uint8))
assert info["kept"] == len(d0) and info["sent"] == 0
# (3) mrh_comm_exchange_halo (no other rank: nothing is taken) and mrh_comm_gather_mesh (device soup -> run merge -> mesh)
assert parallel.exchange_halo(c, comm) == 0
res = parallel.gather_mesh(c, comm)
ref = pu.make_engine(hip, synth.CFG1, dict(synth.CFG1_PARAMS), 16384)
for f in frames:
    pu.feed(ref, f)
t = ref.extract_triangles()
Vr, Fr, Cr = ref.extract_mesh()
assert len(t) > 1000 and np.array_equal(res[0].view(np.uint8), t.view(np.uint8))
assert np.array_equal(res[1], Vr) and np.array_equal(res[2], Fr) and np.array_equal(res[3], Cr)
ph = c.comm_phase_times()
assert ph["pack_ms"] > 0 and ph["unpack_ms"] > 0, ph
# (4) one HIP runtime, one RCCL, no torch in a product-path process
assert "torch" not in sys.modules
maps = open("/proc/self/maps").read()
hips = sorted({{ln.split()[-1] for ln in maps.splitlines() if "libamdhip64" in ln}})
rccls = sorted({{ln.split()[-1] for ln in maps.splitlines() if "librccl" in ln}})
hsas = sorted({{ln.split()[-1] for ln in maps.splitlines() if "libhsa-runtime64" in ln}})
assert len(hips) == 1 and len(rccls) == 1 and len(hsas) == 1, (hips, rccls, hsas)
assert os.path.dirname(os.path.realpath(hips[0])) == os.path.dirname(os.path.realpath(rccls[0])), (hips, rccls)
for e in (a, a2, c):
    e.attach_comm(None)
comm.close()
open({out!r}, "w").write("ok " + hips[0] + " " + rccls[0])
"""


@pytest.mark.parametrize("self_loop", [False, True])
def test_rccl_entry_points_of_the_c_abi_in_a_one_rank_group(hip, tmp_path, self_loop):
    """include/mrhash_comm.h on the GPU: communicator from the file rendezvous, the starve all-reduce inside mrh_integrate,
    mrh_comm_merge_submaps, mrh_comm_exchange_halo, mrh_comm_gather_mesh — RCCL on the library's own stream and buffers.
    Two RCCL ranks cannot share this box's one GPU ("Duplicate GPU detected", profiles/r02/two_ranks_one_device_nccl_outcome.txt),
    so the group has ONE rank; the worker is a process without torch and checks that it holds exactly one libamdhip64, one
    libhsa-runtime64 and one librccl, from the same directory.  self_loop: MRH_COMM_SELF_LOOP=1 sends the rank's own part of
    every exchange (sub-map blocks, halo candidates, block metadata and triangle runs) through grouped ncclSend / ncclRecv to
    itself, so the point-to-point calls carry real payloads at real offsets; every result must be the same."""
    import os
    import subprocess
    import sys

    from test_sharding import ROOT

    out = str(tmp_path / "ok.txt")
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MRH_RDZV_DIR=str(tmp_path))
    if self_loop:
        env["MRH_COMM_SELF_LOOP"] = "1"
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and os.path.exists(out), r.stdout[-2000:] + r.stderr[-4000:]
    assert "/opt/rocm" in os.path.realpath(open(out).read().split()[1])


def test_comm_create_gives_up_when_a_peer_never_joins(hip, tmp_path):
    """ncclCommInitRank has no time limit of its own.  Rank 0 of a two-rank group whose second rank never
    comes: mrh_comm_create must fail after MRH_COMM_INIT_TIMEOUT_S with a message that says so (bench.py then goes on over the
    host group), instead of keeping the process for ever.  In a process of its own: the abandoned init thread stays behind."""
    import os
    import subprocess
    import sys

    from test_sharding import ROOT

    code = (f"import sys, time; sys.path.insert(0, {ROOT!r})\n"
            "from mrhash_amd import capi\n"
            "lib = capi.load_hip(); uid = capi.Comm.unique_id(lib); t0 = time.time()\n"
            "try:\n"
            "    capi.Comm(lib, uid, 0, 2, 0)\n"
            "    print('CREATED')\n"
            "except capi.MrhError as e:\n"
            "    print('ERR', round(time.time() - t0, 1), e)\n"
            "import os; sys.stdout.flush(); os._exit(0)\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MRH_COMM_INIT_TIMEOUT_S="4"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ERR" in r.stdout and "did not return within 4 s" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_exchange_primitives_match_the_oracle(hip, oracle):
    """mrh_pack_blocks / mrh_unpack_blocks / mrh_drop_blocks on the device against the oracle's host versions: the same
    record sets (order aside), the same merged map."""
    from mrhash_amd import hipmem

    a = pu.make_engine(hip, synth.CFG1, dict(synth.CFG1_PARAMS), 16384)
    b = pu.make_engine(oracle, synth.CFG1, dict(synth.CFG1_PARAMS), 16384)
    for e in (a, b):
        pu.feed(e, synth.cfg1_sphere())
        pu.feed(e, synth.cfg1_sphere(zc=1.52))
        e.set_sharding(1, 3, 1)

    def records(e, mode, arg=0):
        import ctypes
        from mrhash_amd import capi

        ptr, n, on_dev = e.pack_blocks(mode, arg)
        if n == 0:
            return np.zeros(0, capi.RECORD_DTYPE)
        if on_dev:
            raw = np.frombuffer(hipmem.read(ptr, n * capi.RECORD_BYTES), dtype=np.uint8).copy()
        else:
            raw = np.frombuffer((ctypes.c_char * (n * capi.RECORD_BYTES)).from_address(ptr), dtype=np.uint8).copy()
        r = raw.view(capi.RECORD_DTYPE)
        return r[np.lexsort((r["desc"]["z"], r["desc"]["y"], r["desc"]["x"]))]

    from mrhash_amd import capi

    for mode, arg in ((capi.PACK_HALO, 0), (capi.PACK_OWNER, 0), (capi.PACK_OWNER, 1), (capi.PACK_OWNER, 2)):
        ra, rb = records(a, mode, arg), records(b, mode, arg)
        assert len(ra) == len(rb) > 0 and ra.tobytes() == rb.tobytes(), (mode, arg)
    # merge a second sub-map into both, from device memory on the HIP side
    c = pu.make_engine(oracle, synth.CFG1, dict(synth.CFG1_PARAMS), 16384)
    pu.feed(c, synth.cfg1_sphere(zc=1.51))
    c.set_sharding(0, 1, 0)
    rc_ = records(c, capi.PACK_OWNER, 0)
    dev = hipmem.DeviceBuffer.from_numpy(rc_.view(np.uint8))
    assert a.unpack_blocks(capi.UNPACK_MERGE, dev.ptr, len(rc_), True) == len(rc_)
    assert b.unpack_blocks(capi.UNPACK_MERGE, rc_.ctypes.data, len(rc_), False) == len(rc_)
    pu.compare_maps(a, b)
    # halo import keeps exactly the adjacent foreign blocks; drop removes them again
    for e in (a, b):
        e.drop_blocks(capi.DROP_FOREIGN)
    na, nb_ = len(a.dump_blocks()[0]), len(b.dump_blocks()[0])
    assert na == nb_ > 0
    ta = a.unpack_blocks(capi.UNPACK_HALO, dev.ptr, len(rc_), True)
    tb = b.unpack_blocks(capi.UNPACK_HALO, rc_.ctypes.data, len(rc_), False)
    assert ta == tb > 0
    pu.compare_maps(a, b)
    assert a.drop_blocks(capi.DROP_HALO) == b.drop_blocks(capi.DROP_HALO) == ta
    pu.compare_maps(a, b)
    assert len(a.dump_blocks()[0]) == na


def test_merging_variance_adaptive_submaps_hip(hip, oracle):
    """VERDICT r05 missing-3: mrh_unpack_blocks(MRH_UNPACK_MERGE) / mrh_comm_merge_submaps on multi-resolution maps (they used to
    return MRH_ERR_UNSUPPORTED).  The rule's properties on the device (coarse wins, order-independent: test_sharding's check), then
    the device against the oracle on the same records: the same map, and the same map again one fused frame later (both sides
    have the same history, so their starve frames coincide)."""
    from mrhash_amd import capi, hipmem
    from test_sharding import _multires_submaps, _records_of, check_mixed_resolution_merge

    a, b = check_mixed_resolution_merge(hip)
    a.close()
    b.close()
    ha, hb = _multires_submaps(hip)
    oa, ob = _multires_submaps(oracle)
    r_h, r_o = _records_of(hb), _records_of(ob)
    assert r_h.tobytes() == r_o.tobytes()  # the same sub-map on both sides
    dev = hipmem.DeviceBuffer.from_numpy(r_h.view(np.uint8))
    th = ha.unpack_blocks(capi.UNPACK_MERGE, dev.ptr, len(r_h), True)  # from device memory, as a collective leaves them
    to = oa.unpack_blocks(capi.UNPACK_MERGE, r_o.ctypes.data, len(r_o), False)
    assert th == to > 0
    r = pu.compare_maps(ha, oa)
    assert r["blocks"] > 60
    sh, so = ha.stats(), oa.stats()
    assert (sh.occupied_fine, sh.occupied_coarse) == (so.occupied_fine, so.occupied_coarse) and sh.error_flags == 0
    for e in (ha, oa):
        pu.feed(e, synth.cfg1_sphere(zc=1.5))
    pu.compare_maps(ha, oa)
    pu.compare_meshes(ha, oa)
    for e in (ha, hb, oa, ob):
        e.close()


def test_lidar_and_splat_seeds_on_tile_shards_hip(hip):
    """Tile sharding beyond the RGB-D path: LiDAR scans (allocation inserts owned blocks only, the per-voxel updates find
    only local blocks) and splat seeds (a seed needs the centre voxel, which exactly one shard holds): the union over the
    shards equals the single-context result."""
    from mrhash_amd import capi

    # LiDAR
    p = dict(synth.VBR_PARAMS, min_weight_threshold=1)

    def lidar_engine(**extra):
        e = capi.Engine(hip, capi.Params(num_sdf_blocks=32768, **p, **extra))
        e.set_camera(1.0, 1.0, 0.0, 0.0, 1, 1, 0.2, 100.0, model=1)
        return e

    single = lidar_engine()
    shards = [lidar_engine(shard_rank=r, shard_count=2, shard_chunk_log2=1) for r in range(2)]
    scene = synth.street_canyon()
    for t, q in synth.drive_poses(2, step=2.0):
        pts = synth.lidar_scan(scene, t, q, rows=16, cols=256)
        for e in [single] + shards:
            e.set_pose(synth.quat_to_rot(q), t)
            e.upload_points(pts)
            e.integrate_points()
    d0, v0 = single.dump_blocks()
    parts = [s.dump_blocks() for s in shards]
    assert all(len(pp[0]) > 50 for pp in parts)
    d = np.concatenate([pp[0] for pp in parts])
    v = np.concatenate([pp[1] for pp in parts])
    order = np.lexsort((d["z"], d["y"], d["x"]))
    assert np.array_equal(d[order], d0) and np.array_equal(v[order].view(np.uint8), v0.view(np.uint8))
    # splat seeds
    params = dict(synth.CFG1_PARAMS)
    single = pu.make_engine(hip, synth.CFG1, params, 16384)
    shards = [pu.make_engine(hip, synth.CFG1, params, 16384, shard_rank=r, shard_count=2, shard_chunk_log2=1) for r in range(2)]
    f = synth.cfg1_sphere()
    for e in [single] + shards:
        pu.feed(e, f)
    s0 = single.splat_seeds(0.0025, 1)
    ss = [e.splat_seeds(0.0025, 1) for e in shards]
    assert all(len(x) > 0 for x in ss) and sum(len(x) for x in ss) == len(s0) > 100
    both = np.concatenate(ss)
    key = lambda a: np.lexsort((a["p"][:, 2], a["p"][:, 1], a["p"][:, 0], a["scale"]))  # noqa: E731
    assert both[key(both)].tobytes() == s0[key(s0)].tobytes()
    assert np.array_equal(single.qtree_leaves(), shards[0].qtree_leaves())  # the tree depends on the image only


@pytest.mark.parametrize("self_loop", [False, True])
def test_plain_c_program_drives_the_comm_abi(hip, tmp_path, self_loop):
    """examples/comm_smoke.c: N processes in plain C — mrh_comm_create, frame-sharded sub-maps, mrh_comm_merge_submaps,
    mrh_comm_exchange_halo, mrh_comm_gather_mesh, the merged map checked against a single context — so that first contact with
    an 8-GPU node can be diagnosed without Python.  On this box: one rank (and, with MRH_COMM_SELF_LOOP, its own parts through
    ncclSend / ncclRecv to itself); MRH_TEST_WORLD=N runs it with N ranks where N devices exist."""
    import os
    import subprocess

    from mrhash_amd import hipmem
    from test_sharding import ROOT

    exe = str(tmp_path / "comm_smoke")
    libdir = os.path.join(ROOT, "mrhash_amd", "csrc")
    subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "comm_smoke.c"), "-o", exe,
                    "-L" + libdir, "-lmrhash_hip", "-Wl,-rpath," + libdir, "-lm"], check=True)
    world = min(int(os.environ.get("MRH_TEST_WORLD", "1")), max(hipmem.device_count(), 1))
    env = dict(os.environ, XDG_RUNTIME_DIR=str(tmp_path))
    if self_loop:
        env["MRH_COMM_SELF_LOOP"] = "1"
    r = subprocess.run([exe, str(world)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert f"comm_smoke: {world} of {world} ranks passed" in r.stdout and r.stdout.count("PASS") == world
    assert "position checksum equal" in r.stdout and "mesh on root" in r.stdout


NRANK_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import parity_utils as pu
from mrhash_amd import capi, hipmem, parallel, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
hip = capi.load_hip()
hipmem.set_device(rank)
comm = parallel.rendezvous(hip)          # rank r on device r, RCCL over xGMI
assert (comm.rank, comm.world) == (rank, world)
comm.barrier()
params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2)
frames = [synth.cfg1_sphere(zc=1.5 + 0.005 * k) for k in range(2 * world)]
# (1) tile-sharded fusion of ONE stream: starve frames all-reduce inside mrh_integrate; halo; mesh on rank 0 == single context
te = pu.make_engine(hip, synth.CFG1, params, 16384, device_id=rank, shard_rank=rank, shard_count=world, shard_chunk_log2=1)
te.attach_comm(comm)
for f in frames:
    pu.feed(te, f)
te.sync()
n_halo = parallel.exchange_halo(te, comm)
res = parallel.gather_mesh(te, comm)
if rank == 0:
    ref = pu.make_engine(hip, synth.CFG1, params, 16384, device_id=rank)
    for f in frames:
        pu.feed(ref, f)
    t = ref.extract_triangles()
    Vr, Fr, Cr = ref.extract_mesh()
    assert len(t) > 1000 and np.array_equal(res[0].view(np.uint8), t.view(np.uint8)), (len(t), len(res[0]))
    assert np.array_equal(res[1], Vr) and np.array_equal(res[2], Fr) and np.array_equal(res[3], Cr)
ph = te.comm_phase_times()
assert ph["allreduce_count"] == 2 * ((2 * world - 1) // 2), ph
parallel.drop_halo(te)
te.attach_comm(None)
# (2) frame-sharded sub-maps -> one tile-sharded map: the union of the owned blocks has the single fusion's occupancy
fe = pu.make_engine(hip, synth.CFG1, dict(synth.CFG1_PARAMS), 16384, device_id=rank)
fe.attach_comm(comm)
for f in frames[2 * rank: 2 * rank + 2]:
    pu.feed(fe, f)
info = parallel.merge_submaps(fe, comm, chunk_log2=1)
d, v = fe.dump_blocks()
own = parallel.owner_of_blocks(np.stack([d["x"], d["y"], d["z"]], 1), world, 1)
assert (own == rank).all()
counts = comm.allgather_i64([len(d), info["sent"], info["received"]])
se = pu.make_engine(hip, synth.CFG1, dict(synth.CFG1_PARAMS), 16384, device_id=rank)
for f in frames:
    pu.feed(se, f)
ds, vs = se.dump_blocks()
assert int(counts[:, 0].sum()) == len(ds), (counts.tolist(), len(ds))
assert int(counts[:, 1].sum()) == int(counts[:, 2].sum()) > 0     # what was sent arrived
fe.attach_comm(None)
comm.barrier()
comm.close()
open({out!r} + str(rank), "w").write("ok")
"""


def test_rccl_exchanges_between_real_devices(hip, tmp_path):
    """The grouped ncclSend / ncclRecv of include/mrhash_comm.h between DISTINCT devices — what a one-GPU box cannot run.
    Un-skips itself where at least two devices are visible: WORLD_SIZE = min(devices, MRH_TEST_WORLD or 8) ranks, rank r on
    device r; tile-sharded fusion with the starve all-reduce, halo exchange and the gathered mesh == a single context's,
    byte for byte; frame-sharded sub-maps merged into the single fusion's occupancy."""
    import os
    import subprocess
    import sys

    from mrhash_amd import hipmem
    from test_sharding import ROOT

    ndev = hipmem.device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} HIP device(s) visible: RCCL refuses two ranks on one device")
    world = min(ndev, int(os.environ.get("MRH_TEST_WORLD", "8")))
    out = str(tmp_path / "ok")
    script = tmp_path / "nrank_worker.py"
    script.write_text(NRANK_WORKER.format(root=ROOT, out=out))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MRH_RDZV_DIR=str(tmp_path), MRH_RDZV_KEY="pytest_nrank")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT " + p.communicate()[0])
    assert all(p.returncode == 0 for p in procs) and all(os.path.exists(out + str(r)) for r in range(world)), "\n----\n".join(o[-2000:] for o in outs)
