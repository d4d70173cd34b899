"""The drop-in surface: `from mrhash.src.pygeowrapper import GeoWrapper` (pybind11 over the C++ host over the C ABI)
driven exactly like the reference's runner does (mrhash/apps/rgbd_runner.py:136-150), checked against the oracle."""
import os

import numpy as np
import pytest

import parity_utils as pu
from mrhash_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def geowrapper_cls(monkeypatch):
    monkeypatch.setenv("MRHASH_NUM_SDF_BLOCKS", "32768")
    from mrhash.src.pygeowrapper import GeoWrapper

    return GeoWrapper


def _make(GeoWrapper, **over):
    kw = dict(sdf_truncation=0.06, sdf_truncation_scale=0.0, integration_weight_sample=1, virtual_voxel_size=0.02,
              n_frames_invalidate_voxels=2, voxel_extents_scale=1, viewer_active=False, marching_cubes_threshold=1.5,
              min_weight_threshold=5, min_depth=0.01, max_depth=30.0)
    kw.update(over)
    return GeoWrapper(**kw)


def test_runner_call_sequence_matches_oracle(geowrapper_cls, oracle, tmp_path):
    g = _make(geowrapper_cls)
    K = synth.CFG1
    g.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 30.0, 0)
    params = dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2)
    b = pu.make_engine(oracle, K, params, 32768)
    frames = [synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.51), synth.cfg1_sphere(zc=1.5)]
    for f in frames:
        g.setCurrPose(f.t, f.q)
        g.setDepthImage(f.depth)
        g.setRGBImage(f.rgb)
        g.compute()
        pu.feed(b, f)
    g.streamAllOut()
    out = tmp_path / "mesh.ply"
    g.extractMesh(str(out))
    V, F, C = g.getVertices(), g.getFaces(), g.getColors()
    b.extract_triangles()
    Vb, Fb, Cb = b.extract_mesh()
    assert V.dtype == np.float64 and F.dtype == np.int32 and C.dtype == np.float64
    assert V.shape == Vb.shape and np.array_equal(F, Fb)
    assert np.array_equal(V, Vb) and np.allclose(C, Cb, atol=1e-5)
    # ASCII PLY with the reference's header (geowrapper.cpp:201-212)
    txt = out.read_text().splitlines()
    assert txt[0] == "ply" and txt[1] == "format ascii 1.0"
    assert txt[2] == f"element vertex {len(V)}" and f"element face {len(F)}" in txt
    hdr_end = txt.index("end_header")
    assert len(txt) == hdr_end + 1 + len(V) + len(F)
    first = txt[hdr_end + 1].split()
    assert len(first) == 6 and abs(float(first[0]) - V[0, 0]) < 1e-4
    assert txt[hdr_end + 1 + len(V)].split() == ["3"] + [str(int(x)) for x in F[0]]
    # every row as the reference's ofstream prints it: doubles through %g (precision 6), colours as (unsigned char) ints
    want = ["%g %g %g %d %d %d" % (v[0], v[1], v[2], int(c[0]) & 255, int(c[1]) & 255, int(c[2]) & 255) for v, c in zip(V, C)]
    want += ["3 %d %d %d" % tuple(f) for f in F]
    assert txt[hdr_end + 1:] == want
    P = g.getCurrPose()
    assert P.shape == (4, 4) and np.allclose(P[:3, :3], frames[-1].R) and np.allclose(P[:3, 3], frames[-1].t)


def test_getters_setters_and_errors(geowrapper_cls):
    g = _make(geowrapper_cls)
    assert g.getNumSdfBlocks() == 32768 and g.getHashNumBuckets() == 32768 and g.getHashBucketSize() == 10
    assert g.getIntegrationWeightMax() == 255 and g.getLinkedListSize() == 7
    assert abs(g.getVirtualVoxelSize() - 0.02) < 1e-6 and g.getNFramesInvalidateVoxels() == 2
    g.setSdfTruncation(0.1)
    assert abs(g.getSdfTruncation() - 0.1) < 1e-6  # host copy only, like the reference (geowrapper.h:98-109)
    with pytest.raises(RuntimeError, match="2D numpy array"):
        g.setDepthImage(np.zeros((4, 4, 1), np.float32))
    with pytest.raises(RuntimeError, match="3D numpy array"):
        g.setRGBImage(np.zeros((4, 4), np.uint8))
    with pytest.raises(RuntimeError, match="3 channels"):
        g.setRGBImage(np.zeros((4, 4, 4), np.uint8))
    with pytest.raises(RuntimeError):
        g.setDepthImage(np.ones((8, 8), np.float32)); g.setRGBImage(np.zeros((8, 8, 3), np.uint8)); g.compute()  # no camera
    g.setRGBImage(np.zeros((8, 8, 3), np.float32))  # the reference runner passes float32 RGB (depth_reader.py:88-91)
    g.GSFinalOpt()
    g.clearBuffers()


def test_sync_compute_reports_a_frames_own_flags(monkeypatch, capfd):
    """compute() enqueues (and keeps a host-fed frame back for one call), so an exhausted pool is announced a
    frame or two late through the peeks.  MRH_SYNC_COMPUTE=1 (or _setSyncCompute) gives the reference's contract back
    (geowrapper.cpp:118-148 ends in cudaDeviceSynchronize): compute() blocks, and the flags ITS frame raised are announced and
    readable in the same call.  A pool of 64 blocks cannot hold a 128x128 sphere view: bit 0 on the very first compute()."""
    monkeypatch.setenv("MRHASH_NUM_SDF_BLOCKS", "64")
    monkeypatch.setenv("MRH_SYNC_COMPUTE", "1")
    from mrhash.src.pygeowrapper import GeoWrapper

    K, f = synth.CFG1, synth.cfg1_sphere()
    g = _make(GeoWrapper)
    g.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 30.0, 0)
    g.setCurrPose(f.t, f.q); g.setDepthImage(f.depth); g.setRGBImage(f.rgb)
    capfd.readouterr()
    g.compute()
    assert g._lastComputeFlags() & 1 and "SDF block pool exhausted" in capfd.readouterr().err
    g.setCurrPose(f.t, f.q); g.setDepthImage(f.depth); g.setRGBImage(f.rgb); g.compute()
    assert g._lastComputeFlags() == 0  # announced once; the map stays usable
    # the default (asynchronous) wrapper: the same frame's flag is NOT known when its compute() returns ...
    monkeypatch.delenv("MRH_SYNC_COMPUTE")
    h = _make(GeoWrapper)
    h.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 30.0, 0)
    h.setCurrPose(f.t, f.q); h.setDepthImage(f.depth); h.setRGBImage(f.rgb); h.compute()
    assert h._lastComputeFlags() == 0
    h._setSyncCompute(True)  # ... and switching over at run time reports it with the next frame
    h.setCurrPose(f.t, f.q); h.setDepthImage(f.depth); h.setRGBImage(f.rgb); h.compute()
    assert h._lastComputeFlags() & 1


def test_serialize_outputs(geowrapper_cls, tmp_path):
    g = _make(geowrapper_cls, n_frames_invalidate_voxels=0)
    K = synth.CFG1
    g.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 30.0, 0)
    f = synth.cfg1_plane()
    g.setCurrPose(f.t, f.q); g.setDepthImage(f.depth); g.setRGBImage(f.rgb); g.compute()
    g.serializeData(str(tmp_path / "hash.ply"), str(tmp_path / "vox.ply"))
    assert "element vertex 100" in (tmp_path / "hash.ply").read_text()
    g.serializeGrid(str(tmp_path / "grid.bin"))
    raw = (tmp_path / "grid.bin").read_bytes()
    assert raw[:8] == b"MRHGRID1" and int.from_bytes(raw[8:16], "little") == 100
    assert len(raw) == 16 + 100 * (16 + 512 * 12)
    # checkpoint / resume: a fresh wrapper restored from the grid file yields the same mesh and keeps fusing identically
    g.extractMesh(str(tmp_path / "a.ply"))
    h = _make(geowrapper_cls, n_frames_invalidate_voxels=0)
    h.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 30.0, 0)
    h.deserializeGrid(str(tmp_path / "grid.bin"))
    h.extractMesh(str(tmp_path / "b.ply"))
    assert np.array_equal(g.getVertices(), h.getVertices()) and np.array_equal(g.getFaces(), h.getFaces())
    f2 = synth.cfg1_plane(z=1.01)
    for w in (g, h):
        w.setCurrPose(f2.t, f2.q); w.setDepthImage(f2.depth); w.setRGBImage(f2.rgb); w.compute()
        w.extractMesh(str(tmp_path / "c.ply"))
    assert np.array_equal(g.getVertices(), h.getVertices()) and np.array_equal(g.getFaces(), h.getFaces())
    with pytest.raises(RuntimeError):
        h.deserializeGrid(str(tmp_path / "hash.ply"))


def test_lidar_runner_call_sequence_matches_oracle(geowrapper_cls, oracle, tmp_path):
    """The reference's LiDAR runners (apps/kitti_runner.py:95-106, rosbag_runner.py:115-126): spherical camera,
    setCurrPose, setPointCloud(points[:, :3], False), compute(); then extractMesh — against the oracle's point path."""
    p = dict(synth.VBR_PARAMS, min_weight_threshold=1)
    g = geowrapper_cls(sdf_truncation=p["sdf_truncation"], sdf_truncation_scale=0.0, integration_weight_sample=1,
                       virtual_voxel_size=p["virtual_voxel_size"], n_frames_invalidate_voxels=0, voxel_extents_scale=1,
                       viewer_active=False, marching_cubes_threshold=1.5, min_weight_threshold=1, min_depth=0.2,
                       max_depth=100.0, projective_sdf=True)
    g.setCamera(1.0, 1.0, 0.0, 0.0, 1, 1, 0.2, 100.0, 1)
    from mrhash_amd import capi

    b = capi.Engine(oracle, capi.Params(num_sdf_blocks=32768, **p))
    b.set_camera(1.0, 1.0, 0.0, 0.0, 1, 1, 0.2, 100.0, model=1)
    scene = synth.street_canyon()
    for k, (t, q) in enumerate(synth.drive_poses(3, step=2.0)):
        pts = synth.lidar_scan(scene, t, q, rows=16, cols=256)
        with_intensity = np.concatenate([pts, np.ones((len(pts), 1), np.float32)], axis=1)  # x y z i, as read from a bag
        g.setCurrPose(t, q)
        if k == 1:  # the (points, normals) overload: with the projective SDF the normals are not used
            g.setPointCloud(with_intensity[:, :3], np.tile(np.array([[0, 0, 1]], np.float32), (len(pts), 1)))
        else:
            g.setPointCloud(with_intensity[:, :3], False)
        g.compute()
        b.set_pose(synth.quat_to_rot(q), t)
        b.upload_points(pts)
        b.integrate_points()
    g.extractMesh(str(tmp_path / "lidar.ply"))
    b.extract_triangles()
    Vb, Fb, Cb = b.extract_mesh()
    assert len(Fb) > 500
    assert np.array_equal(g.getVertices(), Vb) and np.array_equal(g.getFaces(), Fb)


def test_lidar_normal_direction_sdf_and_gc_through_the_wrapper(geowrapper_cls, oracle, tmp_path):
    """GeoWrapper(projective_sdf=False): setPointCloud(points, normals) + compute() with garbage collection on scans, the
    spherical intrinsics of the sensor given to setCamera — against the oracle driven through the C ABI."""
    from mrhash_amd import capi
    from test_lidar import spherical_camera

    cam = spherical_camera(16, 256)
    p = dict(synth.VBR_PARAMS, min_weight_threshold=1, projective_sdf=False, n_frames_invalidate_voxels=3)
    g = geowrapper_cls(sdf_truncation=p["sdf_truncation"], sdf_truncation_scale=0.0, integration_weight_sample=1,
                       virtual_voxel_size=p["virtual_voxel_size"], n_frames_invalidate_voxels=3, voxel_extents_scale=1,
                       viewer_active=False, marching_cubes_threshold=1.5, min_weight_threshold=1, min_depth=0.2,
                       max_depth=100.0, projective_sdf=False)
    g.setCamera(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["rows"], cam["cols"], 0.2, 100.0, 1)
    b = capi.Engine(oracle, capi.Params(num_sdf_blocks=32768, **p))
    b.set_camera(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["rows"], cam["cols"], 0.2, 100.0, model=1)
    scene = synth.street_canyon()
    for t, q in synth.drive_poses(5, step=2.0):
        pts = synth.lidar_scan(scene, t, q, rows=16, cols=256)
        nrm = synth.scan_normals(pts)
        g.setCurrPose(t, q)
        g.setPointCloud(pts, nrm)
        g.compute()
        b.set_pose(synth.quat_to_rot(q), t)
        b.upload_points(pts)
        b.upload_normals(nrm)
        b.integrate_points()
    g.extractMesh(str(tmp_path / "lidar_n.ply"))
    b.extract_triangles()
    Vb, Fb, _ = b.extract_mesh()
    assert len(Fb) > 200
    assert np.array_equal(g.getVertices(), Vb) and np.array_equal(g.getFaces(), Fb)


@pytest.mark.parametrize("var_threshold", [0.0, 0.02])
def test_streamer_pages_far_blocks_out_and_back(monkeypatch, tmp_path, var_threshold):
    """Streamer: with a pool too small for the whole walk, compute() pages blocks farther than max_depth
    from the camera out to the host chunk grid (free blocks <= 15 % of the pool, geowrapper.cpp:137-138) and back in
    when the camera returns; extractMesh sees the whole map.  Paging is transparent: the mesh equals the one of a run
    whose pool holds everything."""
    from mrhash_amd import synth as sy

    K = sy.Intrinsics(160.0, 160.0, 79.5, 59.5, 120, 160)
    kw = dict(sdf_truncation=0.08, sdf_truncation_scale=0.0, integration_weight_sample=1, virtual_voxel_size=0.02,
              n_frames_invalidate_voxels=1000, voxel_extents_scale=1, viewer_active=False, marching_cubes_threshold=1.5,
              min_weight_threshold=1, min_depth=0.01, max_depth=2.0, sdf_var_threshold=var_threshold)
    scene = sy.Scene(sy.Box((-0.6, -0.6, -9.0), (0.6, 0.6, 9.0)), seed=3)  # a corridor: walk along z, then come back
    zs = list(np.arange(-6.0, 6.01, 0.25)) + list(np.arange(5.75, -6.01, -0.25))
    poses = [(np.array([0.0, 0.0, z], np.float32), np.array([0, 0, 0, 1], np.float32)) for z in zs]
    frames = [sy.render(scene, K, t, q, depth_scaling=5000.0) for t, q in poses]

    def run(blocks):
        monkeypatch.setenv("MRHASH_NUM_SDF_BLOCKS", str(blocks))
        from mrhash.src.pygeowrapper import GeoWrapper

        g = GeoWrapper(**kw)
        g.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 2.0, 0)
        peak = 0
        for f in frames:
            g.setCurrPose(f.t, f.q)
            g.setDepthImage(f.depth)
            g.setRGBImage(f.rgb)
            g.compute()
            peak = max(peak, g._hostGridBlocks())
        g.streamAllOut()
        g.extractMesh(str(tmp_path / f"m{blocks}.ply"))
        print('pool', blocks, 'peak host-grid blocks', peak)
        return g.getVertices(), g.getFaces(), peak, g._hostGridBlocks()

    Vb, Fb, peak_big, _ = run(65536)
    Vs, Fs, peak_small, left = run(int(os.environ.get('MRH_TEST_SMALL_POOL', '3072')))
    assert peak_big == 0 and peak_small > 300  # the small pool really paged
    assert left == 0  # extractMesh brought everything back
    assert len(Fb) > 5000
    assert np.array_equal(Vb, Vs) and np.array_equal(Fb, Fs)


def _chunk_of(desc, vs, ext=1.0):
    """Streamer::worldToChunks of a block's origin (streamer.cuh:251-262, streamer.cpp:214-247)."""
    out = []
    for a in ("x", "y", "z"):
        p = np.float32(np.float32(int(desc[a])) * np.float32(8.0 * vs)) / np.float32(ext)
        s = np.float32(int(p > 0) - int(p < 0))
        out.append(int(np.float32(p + s * np.float32(0.5))))
    return tuple(out)


def reference_chunk_loop(e, vs, max_depth, ext=1.0):
    """GeoWrapper::extractMesh's control flow (geowrapper.cpp:150-190) over an engine behind the C ABI — here the ORACLE, whose
    merge mode restates MeshExtractor::processTriangles on (running mesh + new soup) literally: stream everything out into a
    chunk grid, computeBounds, walk the grid in steps of int(10 * max depth) chunks, stream the sphere in, extract, merge,
    stream everything out again.  Returns (V, F, C, iterations that found blocks, blocks left on the engine)."""
    descs, vox = e.stream_out((0.0, 0.0, 0.0), -1.0)
    grid = {}
    for k in range(len(descs)):
        grid.setdefault(_chunk_of(descs[k], vs, ext), []).append(k)
    keys = np.array(sorted(grid))
    lo, hi = keys.min(axis=0), keys.max(axis=0)
    hi = np.where(lo == hi, hi + 1, hi)
    radius = np.float32(10.0) * np.float32(max_depth)
    radiusi = max(1, int(radius))
    chunk_radius = np.float32(np.sqrt(np.float32(3.0 * ext * ext)) / np.float32(2.0))
    e.mesh_merge_begin()
    used = 0
    for x in range(lo[0], hi[0], radiusi):
        for y in range(lo[1], hi[1], radiusi):
            for z in range(lo[2], hi[2], radiusi):
                c = np.array([x, y, z], np.float32) * np.float32(ext)
                sel = []
                for ch, idx in grid.items():
                    d = np.array(ch, np.float32) * np.float32(ext) - c
                    if np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])) <= abs(radius - chunk_radius):
                        sel += idx
                if not sel:
                    continue
                used += 1
                e.import_blocks(descs[sel], vox[sel])
                e.extract_triangles(soup=False)
                e.stream_out((0.0, 0.0, 0.0), -1.0)  # the host copies above ARE the grid: nothing changed on the engine
    total = e.mesh_merge_end()
    V, F, C = e.extract_mesh()
    return V, F, C, used, len(e.dump_blocks()[0]), total


def test_extract_mesh_walks_the_chunk_grid_when_the_map_exceeds_sphere_and_pool(monkeypatch, oracle, tmp_path):
    """The unbuilt half of SURVEY.md 8f-1, built: a 24 m wall seen from 0.4 m with max_depth 0.5 m — 10 * max depth = 5 m, so the
    reference's extractMesh walks the chunk grid in five overlapping spheres (geowrapper.cpp:162-188) — fused through a pool
    SMALLER than the final map (compute() pages), extracted through the chunk loop (no single sphere holds the map, and the
    map does not fit the pool).  V / F / C must equal the oracle's, driven through the same control flow by
    reference_chunk_loop above; afterwards the device map is empty and every block is on the host grid, as in the reference."""
    from mrhash_amd import synth as sy

    K = sy.CFG1
    vs, max_depth = 0.02, 0.5
    kw = dict(sdf_truncation=0.06, sdf_truncation_scale=0.0, integration_weight_sample=1, virtual_voxel_size=vs,
              n_frames_invalidate_voxels=1000, voxel_extents_scale=1, viewer_active=False, marching_cubes_threshold=1.5,
              min_weight_threshold=1, min_depth=0.01, max_depth=max_depth)
    scene = sy.Scene(sy.Box((-2.0, -1.5, -1.0), (27.0, 1.5, 0.4)), seed=5)
    poses = [(np.array([x, 0.0, 0.0], np.float32), np.array([0, 0, 0, 1], np.float32)) for x in np.arange(0.0, 24.01, 0.16)]
    frames = [sy.render(scene, K, t, q, depth_scaling=5000.0) for t, q in poses]
    pool = 384
    monkeypatch.setenv("MRHASH_NUM_SDF_BLOCKS", str(pool))
    from mrhash.src.pygeowrapper import GeoWrapper

    g = GeoWrapper(**kw)
    g.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, max_depth, 0)
    b = pu.make_engine(oracle, K, dict(sy.CFG1_PARAMS, sdf_truncation=0.06, virtual_voxel_size=vs, n_frames_invalidate_voxels=1000,
                                       min_weight_threshold=1, max_depth=max_depth), 16384)
    b.set_camera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, max_depth)
    peak = 0
    for f in frames:
        g.setCurrPose(f.t, f.q)
        g.setDepthImage(f.depth)
        g.setRGBImage(f.rgb)
        g.compute()
        peak = max(peak, g._hostGridBlocks())
        pu.feed(b, f)
    n_map = len(b.dump_blocks()[0])
    assert n_map > pool and peak > 0, (n_map, peak)  # the map does not fit the pool; compute() paged
    Vb, Fb, Cb, used, left_b, total_b = reference_chunk_loop(b, vs, max_depth)
    assert used >= 4 and left_b == 0
    g.extractMesh(str(tmp_path / "walk.ply"))
    V, F, C = g.getVertices(), g.getFaces(), g.getColors()
    assert len(Fb) > 20000 and total_b > len(Fb)  # overlapping spheres extract the same triangles more than once
    assert V.shape == Vb.shape and F.shape == Fb.shape
    assert np.array_equal(V, Vb) and np.array_equal(F, Fb) and np.allclose(C, Cb, atol=1e-5)
    assert g._hostGridBlocks() == n_map  # post-state of the reference: the device map is empty, everything is on the host
    # ... and fusion simply continues: the next frame pages its surroundings back in
    f = frames[len(frames) // 2]
    g.setCurrPose(f.t, f.q); g.setDepthImage(f.depth); g.setRGBImage(f.rgb); g.compute()
    assert 0 < g._hostGridBlocks() < n_map
    # a second extraction gives the same mesh (the running mesh starts empty: D10)
    g.extractMesh(str(tmp_path / "walk2.ply"))
    assert np.array_equal(g.getVertices(), Vb) and np.array_equal(g.getFaces(), Fb)


def test_extract_mesh_refuses_a_sphere_larger_than_the_pool_without_moving_blocks(monkeypatch, tmp_path):
    """A map that neither fits the pool nor splits into spheres that do: extractMesh raises BEFORE importing anything, and
    every block stays in exactly one place (round 2: the import failed half-way, blocks on both sides)."""
    from mrhash_amd import synth as sy

    K = sy.CFG1
    kw = dict(sdf_truncation=0.06, sdf_truncation_scale=0.0, integration_weight_sample=1, virtual_voxel_size=0.02,
              n_frames_invalidate_voxels=1000, voxel_extents_scale=1, viewer_active=False, marching_cubes_threshold=1.5,
              min_weight_threshold=1, min_depth=0.01, max_depth=2.0)
    scene = sy.Scene(sy.Box((-2.0, -1.5, -1.0), (23.0, 1.5, 0.4)), seed=5)
    poses = [(np.array([x, 0.0, 0.0], np.float32), np.array([0, 0, 0, 1], np.float32)) for x in np.arange(0.0, 20.01, 0.16)]
    monkeypatch.setenv("MRHASH_NUM_SDF_BLOCKS", "384")
    from mrhash.src.pygeowrapper import GeoWrapper

    g = GeoWrapper(**kw)
    g.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 2.0, 0)
    for t, q in poses:
        f = sy.render(scene, K, t, q, depth_scaling=5000.0)
        g.setCurrPose(f.t, f.q); g.setDepthImage(f.depth); g.setRGBImage(f.rgb); g.compute()
    g.streamAllOut()
    before = g._hostGridBlocks()
    assert before > 0
    with pytest.raises(RuntimeError, match="holds .* blocks, the pool has"):
        g.extractMesh(str(tmp_path / "no.ply"))
    after = g._hostGridBlocks()
    assert after > 384 and after >= before  # everything on the host grid, nothing lost, nothing doubled
    assert len(g.getVertices()) == 0


def test_gs_runner_sequence_accumulates_splat_seeds(geowrapper_cls, oracle, tmp_path):
    """apps/rgbd_gs_runner.py: the constructor gets configurations/params.json, every compute() seeds splats from the
    frame's quad-tree (geowrapper.cpp:142-143), GSSavePointCloud writes them; the optimiser itself is out of scope."""
    js = tmp_path / "params.json"
    js.write_text('{\n  "kf_thresh": 2000,\n  "qtree_thresh": 0.002,\n  "qtree_min_pixel_size": 1,\n  "kf_iters": 10\n}\n')
    g = _make(geowrapper_cls, gs_optimization_param_path=str(js), n_frames_invalidate_voxels=0)
    K = synth.CFG1
    g.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 30.0, 0)
    b = pu.make_engine(oracle, K, dict(synth.CFG1_PARAMS), 32768)
    want = []
    for f in [synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.56), synth.cfg1_plane(z=1.0)]:
        g.setCurrPose(f.t, f.q)
        g.setDepthImage(f.depth)
        g.setRGBImage(f.rgb)
        g.compute()
        pu.feed(b, f)
        want.append(b.splat_seeds(0.002, 1))
    want = np.concatenate(want)
    xyz, scale, rgb = g._splatSeeds()
    assert len(want) > 100 and xyz.shape == (len(want), 3)
    assert np.array_equal(xyz, want["p"]) and np.array_equal(scale[:, 0], want["scale"]) and np.array_equal(rgb, want["rgb"])
    out = tmp_path / "gs_out"
    g.GSSavePointCloud(str(out))
    txt = (out / "point_cloud.ply").read_text().splitlines()
    assert txt[2] == f"element vertex {len(want)}" and len(txt) == txt.index("end_header") + 1 + len(want)
    with pytest.raises(RuntimeError):
        g.GSFinalOpt()
    with pytest.raises(RuntimeError):
        _make(geowrapper_cls, gs_optimization_param_path=str(tmp_path / "missing.json"))


def test_streamer_single_stream_like_the_reference(monkeypatch, tmp_path):
    """The reference's own streamer test (tests/test_streamer.cu STREAMER.SingleStream): a circular camera path over a
    constant-depth image, Streamer::stream(camera position, radius 3 m) before every integrate, a second lap that only
    streams, then streamAllOut; it passes when fewer than 15 % of the blocks exist twice (host grid + device).  Here the
    same call sequence must leave NO block twice, and — the geometry lies within the radius — the final mesh must equal
    the mesh of a run that never streams."""
    monkeypatch.setenv("MRHASH_NUM_SDF_BLOCKS", "131072")
    from mrhash.src.pygeowrapper import GeoWrapper

    rows = cols = 150
    steps, t_step, radius = 40, 1.5, 3.0
    kw = dict(sdf_truncation=0.02, sdf_truncation_scale=0.01, integration_weight_sample=3, virtual_voxel_size=0.005,
              n_frames_invalidate_voxels=10, voxel_extents_scale=1, viewer_active=False, marching_cubes_threshold=1.5,
              min_weight_threshold=0, min_depth=0.0, max_depth=5.0)
    depth = np.full((rows, cols), 1.0, np.float32)
    rgb = np.zeros((rows, cols, 3), np.uint8)
    rgb[..., 0] = 255
    # makeCameraCircularTrajectory (tests/test_utils.cuh:20-32)
    a = 2.0 * np.pi / steps
    step = np.eye(4)
    step[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
    step[:3, 3] = [t_step, 0, t_step]
    T, path = np.eye(4), []
    for k in range(steps + 1):
        T = T @ step
        th = (k + 1) * a  # rotation about y by th: (qx, qy, qz, qw) = (0, sin(th/2), 0, cos(th/2))
        path.append((T[:3, 3].astype(np.float32), np.array([0.0, np.sin(th / 2), 0.0, np.cos(th / 2)], np.float32)))

    def positions(g, name):
        g.serializeGrid(str(tmp_path / name))
        raw = np.frombuffer((tmp_path / name).read_bytes(), np.uint8)
        n = int(raw[8:16].view(np.uint64)[0])
        rec = raw[16:].reshape(n, 16 + 512 * 12)
        return np.ascontiguousarray(rec[:, :12]).view(np.int32).reshape(n, 3)

    def run(stream):
        g = GeoWrapper(**kw)
        g.setCamera(200.0, 200.0, cols / 2.0, rows / 2.0, rows, cols, 0.0, 5.0, 0)
        peak = 0
        for t, q in path:
            if stream:
                g._stream(t, radius)
                peak = max(peak, g._hostGridBlocks())
            g.setCurrPose(t, q)
            g.setDepthImage(depth)
            g.setRGBImage(rgb)
            g.compute()
        if stream:
            pos = positions(g, "lap1.bin")
            assert len(np.unique(pos, axis=0)) == len(pos), "a block exists twice after the fusing lap"
            for t, _ in path:  # second lap: stream only
                g._stream(t, radius)
            pos2 = positions(g, "lap2.bin")
            assert len(np.unique(pos2, axis=0)) == len(pos2) == len(pos), "streaming alone changed the set of blocks"
        g.streamAllOut()
        g.extractMesh(str(tmp_path / ("s.ply" if stream else "n.ply")))
        return g.getVertices(), g.getFaces(), peak

    Vn, Fn, _ = run(False)
    Vs, Fs, peak = run(True)
    assert peak > 1000, "the path never left the streaming radius"
    assert len(Fn) > 10000
    assert np.array_equal(Vn, Vs) and np.array_equal(Fn, Fs)


def test_example_runner_script_runs(tmp_path):
    """examples/fuse_synthetic.py — the reference's RGB-D runner loop through `from mrhash.src.pygeowrapper import GeoWrapper` —
    end to end: a mesh file with the header the reference writes, single- and multi-resolution."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ([], ["--var", "0.005"]):
        out = tmp_path / "mesh.ply"
        r = subprocess.run([sys.executable, os.path.join(root, "examples", "fuse_synthetic.py"), "--frames", "12", "--out", str(out)] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        head = open(out).read(400).splitlines()
        assert head[0] == "ply" and head[1] == "format ascii 1.0" and head[2].startswith("element vertex ")
        assert int(head[2].split()[-1]) > 10000


def test_geowrapper_comm_entry_points_one_rank(geowrapper_cls, oracle, tmp_path):
    """The multi-GPU surface of the C++ GeoWrapper (include/mrhash_comm.h behind it: _commUniqueId / _commInit / _mergeSubmaps)
    with the one rank this box allows: the communicator comes up on the wrapper's device, a tile-sharded wrapper fuses and
    extracts as before (world 1: the shard owns everything), a frame-sharded one folds its sub-map through mrh_comm_merge_submaps
    and then extracts — the same mesh as the oracle's."""
    K = synth.CFG1
    frames = [synth.cfg1_sphere(), synth.cfg1_sphere(zc=1.51), synth.cfg1_sphere(zc=1.5)]
    b = pu.make_engine(oracle, K, dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2), 32768)
    for f in frames:
        pu.feed(b, f)
    b.extract_triangles()
    Vb, Fb, Cb = b.extract_mesh()
    for tile_sharded in (True, False):
        g = _make(geowrapper_cls)
        g.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, 0.01, 30.0, 0)
        uid = geowrapper_cls._commUniqueId()
        assert isinstance(uid, bytes) and len(uid) == 128
        g._commInit(uid, 0, 1, 1, tile_sharded)
        with pytest.raises(RuntimeError, match="already attached"):
            g._commInit(uid, 0, 1, 1, tile_sharded)
        for f in frames:
            g.setCurrPose(f.t, f.q)
            g.setDepthImage(f.depth)
            g.setRGBImage(f.rgb)
            g.compute()
        if not tile_sharded:
            g._mergeSubmaps()
        g.extractMesh(str(tmp_path / f"comm{int(tile_sharded)}.ply"))
        assert np.array_equal(g.getVertices(), Vb) and np.array_equal(g.getFaces(), Fb)
        del g
