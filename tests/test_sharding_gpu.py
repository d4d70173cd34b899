"""Tile-sharded path with the HIP library: two processes (gloo rendezvous, both on the one GPU of the test box),
halo all-gather, merged mesh == single-context mesh, bit for bit."""
import numpy as np
import pytest

import parity_utils as pu
from mrhash_amd import parallel, synth
from test_sharding import reference_single, run_two_ranks

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


def test_two_rank_mesh_equals_single_context_hip(hip, tmp_path):
    got = run_two_ranks(tmp_path, use_hip=True)
    t, V, F, C = reference_single(hip)
    assert int(got["n_halo"]) > 0
    assert np.array_equal(got["tris"], t.view(np.uint8))
    assert np.array_equal(got["F"], F) and np.array_equal(got["V"], V)
