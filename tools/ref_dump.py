#!/usr/bin/env python3
"""Reference-side golden vectors for the result-carrying functions (the one road to `parity: green`).

Run on ANY box where the reference (rvp-group/mrhash) is built with CUDA and importable as `mrhash.src.pygeowrapper`:

    PYTHONPATH=<reference checkout>:<this repo> python tools/ref_dump.py [out_dir = tests/golden]

It feeds this repo's committed synthetic frames (mrhash_amd/synth.py: pure numpy, seeded) through the REFERENCE's own
GeoWrapper — setCamera / setCurrPose / setDepthImage / setRGBImage / compute() per frame, then streamAllOut + serializeData
(streamer.cpp:104-160: block origins and weighted voxels with weight and sdf) and extractMesh (getVertices / getFaces /
getColors) — and writes one `ref_cuda_<case>.npz` per case: the canonical occupancy (sorted block coordinates), the weighted
voxels (position, weight, sdf) and V / F / C.  `tests/test_reference_dump.py` consumes such a fixture when it is present and
compares this library's GeoWrapper on the same frames: occupancy and face indices exactly, weights exactly, TSDF values and
vertex positions within 1e-5.  Nothing of this has been run yet — no CUDA device in the build container.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mrhash_amd import synth  # noqa: E402

CASES = {
    # name: (intrinsics, params, frames)
    "cfg1_plane": (synth.CFG1, synth.CFG1_PARAMS, lambda: [synth.cfg1_plane()]),
    "cfg1_sphere_x3": (synth.CFG1, synth.CFG1_PARAMS, lambda: [synth.cfg1_sphere(zc=1.5 + 0.01 * k) for k in range(3)]),
    "replica_640x480_x4": (synth.REPLICA_640, synth.REPLICA_PARAMS, lambda: list(synth.replica_stream(4))),
}


def read_ply_points(path):
    """x y z r g b weight [sdf] rows of a PointCloudSerializer file (ascii or binary_little_endian, float properties)."""
    with open(path, "rb") as f:
        header, fmt, n, props = [], "ascii", 0, []
        while True:
            ln = f.readline().decode("ascii", "replace").strip()
            header.append(ln)
            if ln.startswith("format"):
                fmt = ln.split()[1]
            elif ln.startswith("element vertex"):
                n = int(ln.split()[-1])
            elif ln.startswith("property"):
                props.append(ln.split()[1:])
            elif ln == "end_header":
                break
        if fmt == "ascii":
            a = np.loadtxt(f, dtype=np.float64, ndmin=2) if n else np.zeros((0, len(props)))
        else:
            tmap = {"float": "<f4", "float32": "<f4", "double": "<f8", "uchar": "u1", "uint8": "u1", "int": "<i4", "uint": "<u4"}
            dt = np.dtype([(p[1], tmap[p[0]]) for p in props])
            rec = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
            a = np.stack([rec[p[1]].astype(np.float64) for p in props], 1) if n else np.zeros((0, len(props)))
    return a, [p[1] for p in props]


def run_case(GeoWrapper, name, out_dir):
    K, P, make = CASES[name]
    g = GeoWrapper(sdf_truncation=P["sdf_truncation"], sdf_truncation_scale=P["sdf_truncation_scale"],
                   integration_weight_sample=P["integration_weight_sample"], virtual_voxel_size=P["virtual_voxel_size"],
                   n_frames_invalidate_voxels=P["n_frames_invalidate_voxels"], voxel_extents_scale=P.get("voxel_extents_scale", 1),
                   viewer_active=False, marching_cubes_threshold=P["marching_cubes_threshold"], min_weight_threshold=P["min_weight_threshold"],
                   min_depth=P["min_depth"], max_depth=P["max_depth"], sdf_var_threshold=P.get("sdf_var_threshold", 0.0),
                   vertices_merging_threshold=P.get("vertices_merging_threshold", 0.0), projective_sdf=True)
    g.setCamera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, P["min_depth"], P["max_depth"], 0)
    for f in make():
        g.setCurrPose(np.asarray(f.t, np.float32), np.asarray(f.q, np.float32))
        g.setDepthImage(np.ascontiguousarray(f.depth, np.float32))
        g.setRGBImage(np.ascontiguousarray(f.rgb, np.uint8))
        g.compute()
    mesh_path = os.path.join(out_dir, f"_ref_{name}.ply")
    g.extractMesh(mesh_path)  # pages the map out through the chunk grid as a side effect (geowrapper.cpp:150-190)
    V, F, C = np.asarray(g.getVertices()), np.asarray(g.getFaces()), np.asarray(g.getColors())
    g.streamAllOut()
    hp, vp = os.path.join(out_dir, f"_ref_{name}_hash.ply"), os.path.join(out_dir, f"_ref_{name}_vox.ply")
    g.serializeData(hp, vp)
    H, _ = read_ply_points(hp)
    X, cols = read_ply_points(vp)
    vs = P["virtual_voxel_size"]
    occ = np.unique(np.rint(H[:, :3] / (8 * vs)).astype(np.int32), axis=0)            # blocks with at least one weighted voxel
    vox = np.rint(X[:, :3] / vs).astype(np.int32)
    order = np.lexsort((vox[:, 2], vox[:, 1], vox[:, 0]))
    np.savez_compressed(os.path.join(out_dir, f"ref_cuda_{name}.npz"), case=name, occupancy=occ, voxel_pos=vox[order],
                        voxel_weight=X[order, cols.index("weight")].astype(np.uint8), voxel_sdf=X[order, cols.index("sdf")].astype(np.float32),
                        V=V.astype(np.float64), F=F.astype(np.int32), C=C.astype(np.float64))
    for p in (mesh_path, hp, vp):
        os.remove(p)
    print(f"{name}: {len(occ)} weighted blocks, {len(vox)} weighted voxels, {len(V)} vertices, {len(F)} faces")


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")
    from mrhash.src.pygeowrapper import GeoWrapper  # the REFERENCE's module (its build puts it there, apps/rgbd_runner.py:9)

    if "mrhash_amd" in (getattr(sys.modules[GeoWrapper.__module__], "__file__", "") or ""):
        raise SystemExit("ref_dump.py: `mrhash.src.pygeowrapper` resolved to THIS repository's shim; put the reference checkout first on PYTHONPATH")
    for case in CASES:
        run_case(GeoWrapper, case, out)
