#!/usr/bin/env python3
"""Generates tests/golden/cfg1_golden.json and tests/golden/lidar_golden.json from the CPU oracle (the reference cannot be built or run here, and
its own tests store no golden values for this path — SURVEY.md §8c).  The fixture holds, per case, the
parameters, a description of the synthetic frames, summary counts and SHA-256 of the canonical buffers:
sorted occupancy list, position-keyed voxel payload, canonical triangle buffer and face index buffer."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import parity_utils as pu  # noqa: E402
from mrhash_amd import synth  # noqa: E402

CASES = {
    "plane_1frame": dict(params=dict(synth.CFG1_PARAMS), frames=[dict(kind="plane", z=1.0)]),
    "sphere_3frames_gc": dict(params=dict(synth.CFG1_PARAMS, n_frames_invalidate_voxels=2),
                              frames=[dict(kind="sphere", zc=1.5), dict(kind="sphere", zc=1.51), dict(kind="sphere", zc=1.5)]),
    "sphere_multires": dict(params=dict(synth.CFG1_PARAMS, sdf_var_threshold=0.5, n_frames_invalidate_voxels=3),
                            frames=[dict(kind="sphere", zc=1.5), dict(kind="sphere", zc=1.52), dict(kind="sphere", zc=1.49), dict(kind="sphere", zc=1.5)]),
}


SPLAT = (0.0025, 1)  # quad-tree threshold / min pixel size of the splat-seed hashes (128x128 image: 0.1 * 128^2 / 640x480 ~ 0.005)

LIDAR_CASES = {
    # scans are rebuilt by parity_utils.lidar_scans_from_spec: street-canyon scene, drive poses, points rounded to 1 mm
    "street_3scans": dict(params=dict(synth.VBR_PARAMS, min_weight_threshold=1), max_depth=100.0,
                          scans=dict(rows=16, cols=256, n=3, step=2.0)),
    "street_clipped_2scans": dict(params=dict(synth.VBR_PARAMS, min_weight_threshold=1, virtual_voxel_size=0.25, sdf_truncation=0.5),
                                  max_depth=30.0, scans=dict(rows=8, cols=512, n=2, step=4.0)),
}


def summarize(e):
    d, v = e.dump_blocks()
    t = e.extract_triangles()
    V, F, C = e.extract_mesh()
    return dict(blocks=int(len(d)), coarse_blocks=int((d["resolution"] == 1).sum()), weighted_voxels=int((v["weight"] > 0).sum()),
                triangles=int(t.shape[0]), vertices=int(V.shape[0]), faces=int(F.shape[0]),
                sha256_occupancy=hashlib.sha256(d.tobytes()).hexdigest(), sha256_payload=hashlib.sha256(v.tobytes()).hexdigest(),
                sha256_triangles=hashlib.sha256(t.tobytes()).hexdigest(), sha256_faces=hashlib.sha256(F.tobytes()).hexdigest())


def main_lidar(orc):
    out = {"generator": "tests/golden/make_golden.py (oracle/mrh_oracle.c)", "cases": {}}
    for name, case in LIDAR_CASES.items():
        e = pu.make_lidar_engine(orc, case["params"], case["max_depth"], 32768)
        for t, q, pts in pu.lidar_scans_from_spec(case["scans"]):
            e.set_pose(synth.quat_to_rot(q), t)
            e.upload_points(pts)
            e.integrate_points()
        out["cases"][name] = dict(params=case["params"], max_depth=case["max_depth"], scans=case["scans"], **summarize(e))
        print(name, {k: out["cases"][name][k] for k in ("blocks", "weighted_voxels", "triangles", "faces")})
        e.close()
    json.dump(out, open(os.path.join(HERE, "lidar_golden.json"), "w"), indent=1, sort_keys=True)


def main():
    orc = pu.oracle_lib()
    main_lidar(orc)
    out = {"generator": "tests/golden/make_golden.py (oracle/mrh_oracle.c)", "cases": {}}
    for name, case in CASES.items():
        e = pu.make_engine(orc, synth.CFG1, case["params"], 16384)
        for spec in case["frames"]:
            pu.feed(e, pu.frame_from_spec(spec))
        d, v = e.dump_blocks()
        t = e.extract_triangles()
        V, F, C = e.extract_mesh()
        seeds = e.splat_seeds(*SPLAT)  # 3DGS splat seeds of the last frame
        leaves = e.qtree_leaves()
        out["cases"][name] = dict(
            splat=dict(qtree_thresh=SPLAT[0], qtree_min_pixel_size=SPLAT[1], leaves=int(len(leaves)), seeds=int(len(seeds)),
                       sha256_leaves=hashlib.sha256(leaves.tobytes()).hexdigest(), sha256_seeds=hashlib.sha256(seeds.tobytes()).hexdigest()),
            params=case["params"], frames=case["frames"], blocks=int(len(d)), coarse_blocks=int((d["resolution"] == 1).sum()),
            weighted_voxels=int((v["weight"] > 0).sum()), triangles=int(t.shape[0]), vertices=int(V.shape[0]), faces=int(F.shape[0]),
            sha256_occupancy=hashlib.sha256(d.tobytes()).hexdigest(), sha256_payload=hashlib.sha256(v.tobytes()).hexdigest(),
            sha256_triangles=hashlib.sha256(t.tobytes()).hexdigest(), sha256_faces=hashlib.sha256(F.tobytes()).hexdigest(),
        )
        print(name, {k: out["cases"][name][k] for k in ("blocks", "coarse_blocks", "weighted_voxels", "triangles", "faces")})
    json.dump(out, open(os.path.join(HERE, "cfg1_golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
