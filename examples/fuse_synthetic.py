#!/usr/bin/env python3
"""The reference's RGB-D runner loop (mrhash/apps/rgbd_runner.py:96-152) on a synthetic stream: same import, same
constructor keywords, same per-frame calls — only the dataset reader is replaced by `mrhash_amd.synth` (there are no
datasets in this environment).

    python examples/fuse_synthetic.py --frames 200 --out /tmp/mesh.ply [--var 0.005] [--gs params.json]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mrhash.src.pygeowrapper import GeoWrapper  # noqa: E402  (the reference's import path)
from mrhash_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--out", default="/tmp/mrhash_mesh.ply")
    ap.add_argument("--var", type=float, default=0.0, help="sdf_var_threshold (> 0: variance-adaptive resolution)")
    ap.add_argument("--gs", default="", help="gs_optimization_param_path: turns the 3DGS splat seeding on")
    args = ap.parse_args()

    K = synth.REPLICA_640
    p = synth.REPLICA_PARAMS  # configurations/replica.cfg
    geo_wrapper = GeoWrapper(
        sdf_truncation=p["sdf_truncation"], sdf_truncation_scale=p["sdf_truncation_scale"],
        integration_weight_sample=p["integration_weight_sample"], virtual_voxel_size=p["virtual_voxel_size"],
        n_frames_invalidate_voxels=p["n_frames_invalidate_voxels"], voxel_extents_scale=p["voxel_extents_scale"],
        viewer_active=False, marching_cubes_threshold=p["marching_cubes_threshold"],
        min_weight_threshold=p["min_weight_threshold"], min_depth=p["min_depth"], max_depth=p["max_depth"],
        gs_optimization_param_path=args.gs, sdf_var_threshold=args.var,
        vertices_merging_threshold=p["vertices_merging_threshold"], projective_sdf=True)
    geo_wrapper.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, p["min_depth"], p["max_depth"], 0)

    frames = list(synth.replica_stream(args.frames))  # stands in for the dataset reader
    t0 = time.perf_counter()
    for f in frames:
        geo_wrapper.setCurrPose(f.t, f.q)
        geo_wrapper.setDepthImage(f.depth)
        geo_wrapper.setRGBImage(f.rgb)
        geo_wrapper.compute()
    geo_wrapper.streamAllOut()
    dt = time.perf_counter() - t0
    print(f"fused {len(frames)} frames in {dt * 1e3:.1f} ms ({len(frames) / dt:.0f} frames/s, host numpy inputs)")
    t0 = time.perf_counter()
    geo_wrapper.extractMesh(args.out)
    print(f"extractMesh: {1e3 * (time.perf_counter() - t0):.1f} ms, {len(geo_wrapper.getVertices())} vertices, {len(geo_wrapper.getFaces())} faces")
    if args.gs:
        geo_wrapper.GSSavePointCloud(os.path.splitext(args.out)[0] + "_gs")


if __name__ == "__main__":
    main()
