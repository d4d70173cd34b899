"""The oracle against the committed golden fixtures (tests/golden/*.json, written by tests/golden/make_golden.py): a
drift guard for the checker itself.  The same fixtures are the targets of the HIP path in the -m gpu tests."""
import hashlib
import json
import os

import pytest

import parity_utils as pu
from mrhash_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def check(e, case, name):
    d, v = e.dump_blocks()
    assert len(d) == case["blocks"], name
    assert int((d["resolution"] == 1).sum()) == case["coarse_blocks"], name
    assert hashlib.sha256(d.tobytes()).hexdigest() == case["sha256_occupancy"], name
    assert hashlib.sha256(v.tobytes()).hexdigest() == case["sha256_payload"], name
    t = e.extract_triangles()
    V, F, C = e.extract_mesh()
    assert (t.shape[0], V.shape[0], F.shape[0]) == (case["triangles"], case["vertices"], case["faces"]), name
    assert hashlib.sha256(t.tobytes()).hexdigest() == case["sha256_triangles"], name
    assert hashlib.sha256(F.tobytes()).hexdigest() == case["sha256_faces"], name
    if "splat" in case:  # 3DGS splat seeds of the last frame
        sp = case["splat"]
        seeds = e.splat_seeds(sp["qtree_thresh"], sp["qtree_min_pixel_size"])
        leaves = e.qtree_leaves()
        assert (len(leaves), len(seeds)) == (sp["leaves"], sp["seeds"]), name
        assert hashlib.sha256(leaves.tobytes()).hexdigest() == sp["sha256_leaves"], name
        assert hashlib.sha256(seeds.tobytes()).hexdigest() == sp["sha256_seeds"], name


def run_rgbd(lib, case):
    e = pu.make_engine(lib, synth.CFG1, case["params"], 16384)
    for spec in case["frames"]:
        pu.feed(e, pu.frame_from_spec(spec))
    return e


def run_lidar(lib, case):
    e = pu.make_lidar_engine(lib, case["params"], case["max_depth"], 32768)
    for t, q, pts in pu.lidar_scans_from_spec(case["scans"]):
        e.set_pose(synth.quat_to_rot(q), t)
        e.upload_points(pts)
        e.integrate_points()
    return e


@pytest.mark.parametrize("fixture,runner", [("cfg1_golden.json", run_rgbd), ("lidar_golden.json", run_lidar)])
def test_oracle_reproduces_golden(oracle, fixture, runner):
    g = json.load(open(os.path.join(GOLDEN, fixture)))
    assert g["cases"]
    for name, case in g["cases"].items():
        e = runner(oracle, case)
        check(e, case, name)
        e.close()
