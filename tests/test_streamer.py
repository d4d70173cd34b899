"""Streamer, device half: mrh_stream_out + mrh_import_blocks on the oracle (CPU)."""
import numpy as np

import parity_utils as pu
from mrhash_amd import synth


def build(lib, n=3):
    K = synth.Intrinsics(160.0, 160.0, 79.5, 59.5, 120, 160)
    params = dict(synth.REPLICA_PARAMS, virtual_voxel_size=0.02, sdf_truncation=0.08, n_frames_invalidate_voxels=1000)
    e = pu.make_engine(lib, K, params, 65536)
    scene = synth.scannet_room()
    for t, q in synth.walk_poses(n, seed=7):
        pu.feed(e, synth.render(scene, K, t, q, depth_scaling=5000.0))
    return e


def canon(d, v):
    order = np.lexsort((d["z"], d["y"], d["x"]))
    return d[order], v[order]


def test_stream_out_partitions_the_map_and_import_restores_it(oracle):
    e = build(oracle)
    d0, v0 = canon(*e.dump_blocks())
    centre, radius = (0.0, 0.0, 0.0), 2.5
    origin = np.stack([d0["x"], d0["y"], d0["z"]], 1).astype(np.float32) * np.float32(8) * np.float32(0.02)
    far = np.sqrt(((origin - np.asarray(centre, np.float32)) ** 2).sum(1)) >= radius
    assert 0 < far.sum() < len(d0)
    ds, vs = e.stream_out(centre, radius)
    assert len(ds) == int(far.sum())
    assert ds.tobytes() == d0[far].tobytes() and vs.tobytes() == v0[far].tobytes()  # position order, exact payload
    dr, vr = canon(*e.dump_blocks())
    assert dr.tobytes() == d0[~far].tobytes() and vr.tobytes() == v0[~far].tobytes()
    st = e.stats()
    assert st.occupied_fine == int((~far).sum()) and st.occupied_fine + st.free_fine == 65536  # slots went back to the free list
    e.import_blocks(ds, vs)  # Streamer::streamInToGPU
    d1, v1 = canon(*e.dump_blocks())
    assert d1.tobytes() == d0.tobytes() and v1.tobytes() == v0.tobytes()
    # streamAllOut
    da, va = e.stream_out(centre, -1.0)
    assert da.tobytes() == d0.tobytes() and va.tobytes() == v0.tobytes()
    assert e.stats().occupied_fine == 0
    e.close()
