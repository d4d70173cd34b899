#!/usr/bin/env python3
"""Extraction on maps around and beyond k_block_rank's limit of 32 k blocks (run under rocprofv3 --kernel-trace --stats to see
k_block_rank / k_sort_* / k_mc_scan_total): the bench's room at a finer voxel size, a few frames, two extractions.
usage: python tools/bench_big_map.py <voxel size in m> [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mrhash_amd import capi, synth  # noqa: E402

vs = float(sys.argv[1]) if len(sys.argv) > 1 else 0.005
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 6
hip = capi.load_hip()
P = dict(synth.REPLICA_PARAMS, virtual_voxel_size=vs, sdf_truncation=7 * vs)
K = synth.REPLICA_640
e = capi.Engine(hip, capi.Params(num_sdf_blocks=1 << 20, device_id=0, **P))
e.set_camera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, P["min_depth"], P["max_depth"])
frames = bench.render_stream("replica", nf)
for f in frames:
    e.set_pose(f.R, f.t); e.upload_depth(f.depth); e.upload_rgb(f.rgb); e.integrate()
e.sync()
for rep in range(2):
    t0 = time.perf_counter()
    tris = e.extract_triangles()
    dt = time.perf_counter() - t0
    st = e.stats()
    print(f"voxel {vs}: {int(st.occupied_fine)} live blocks, {len(tris)} triangles, extraction {dt * 1e3:.2f} ms (call {rep})")
e.close()
