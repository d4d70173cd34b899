"""bench.py keeps its contract: ONE JSON line on stdout with the driver's keys, the roofline and (when asked) the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
        "roofline", "cpu_baseline")


def _run(args, env=None, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must hold exactly the result line, got {len(lines)} lines"
    d = json.loads(lines[0])
    d["_stderr_failures"] = [ln for ln in r.stderr.splitlines() if "failed" in ln or "Error" in ln or "error" in ln][-12:]
    return d


_SINGLE = {}


def _single_line(steps, warmup, extra=()):
    """The N = 1 line at the given step counts, run once per module (the N > 1 tests compare their rank 0 with it)."""
    key = (steps, warmup, tuple(extra))
    if key not in _SINGLE:
        _SINGLE[key] = _run(["--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--no-pmc"] + list(extra))
    return _SINGLE[key]


def test_single_gpu_line():
    d = _single_line(6, 2, ("--cpu-frames", "2"))
    assert all(k in d for k in KEYS)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["value"] > 1000 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["launches"] == 6
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert d["roofline_hbm"]["exceeds_infinity_cache"] in (True, False) and d["mc"]["triangles"] > 0 and d["lidar"]["scans_per_s"] > 0 and d["splat"]["seeds_per_frame"] > 0
    assert d["pcie_inclusive_frames_per_s"] > 0 and "workload" in d["config"]
    assert d["periodic_frames"]["steady_frame_ms"] > 0 and d["periodic_frames"]["starve_frame_ms"] > d["periodic_frames"]["steady_frame_ms"]
    assert d["spherical_images"]["frames_per_s"] > 0 and d["roofline"]["k_front_ms_avg"] > 0
    # the parts of the legs that need profile mode run after every timed region and fill these in
    assert len(d["spherical_images"]["passes_ms_per_frame"]) == 3 and d["spherical_images"]["roofline"]["frac"] > 0
    assert d["mc"]["k_mc_count_ms"] > 0 and d["mc"]["roofline"]["frac"] > 0 and d["mc"]["first_extract_ms"] >= d["mc"]["extract_ms_in_library"]
    assert d["lidar"]["roofline"]["frac"] > 0 and d["lidar"]["roofline"]["updated_voxels_per_scan"] > 0
    # the oracle of the cpu_baseline leg also checks the map of the same frames through the timed loop's entry points
    assert d["parity_checked"] is True and c["parity"]["frames"] == 2 and c["parity"]["sdf_bit_exact"] and d["blocks"] == c["parity"]["oracle_blocks"] > 1000


@pytest.mark.parametrize("ranks", [2, 8])
def test_n_ranks_on_one_device_line(ranks):
    """The N > 1 code path (self-spawned ranks, frame-sharded value, merge, tile-sharded mode) with all ranks on device 0 over
    gloo — 8 ranks = BASELINE.json configs[3]'s rank count: the 8-way owner function, eight split lists, eight halo segments.
    `n_gpus` counts DEVICES (one here); `ranks` says how many processes shared it."""
    import warnings

    first = None
    for attempt in range(2):
        d = _run(["--gpus", str(ranks), "--steps", "6", "--warmup", "2", "--blocks", "65536"],
                 env={"MRH_BENCH_SHARE_DEVICE": "1", "MRH_BENCH_FULL_STREAM": str(6 * ranks)}, timeout=900)  # full-stream leg: 6 frames per rank here, 500 / N in a real run
        assert all(k in d for k in KEYS)
        err = d.get("phases_error")  # an exception inside the exchange phases is reported in the line, not as an exit code
        # Seen twice in ~30 runs of round 5 with eight processes on the one device: gloo's TCP transport drops a pair in the middle of
        # the host-staged all-to-all ("Connection closed by peer").  That is the stand-in transport of this test, not the path under
        # test: ONE retry, for that message only, with what the ranks wrote to stderr kept in a warning; anything else fails at once.
        if err and attempt == 0 and "gloo" in err and "Connection" in err:
            first = (err, d["_stderr_failures"])
            warnings.warn(f"test_n_ranks_on_one_device_line[{ranks}]: gloo transport error, retrying once: {first}")
            continue
        assert err is None, (err, d["_stderr_failures"], first)
        break
    assert d["n_gpus"] == 1 and d["ranks"] == ranks and d["scaling"] == "weak" and d["fuse_only_frames_per_s"] > 1000
    # RCCL's own view of the group has its place in the line; over gloo it says why it is empty
    assert d["rccl"]["rccl_ranks"] is None and "gloo" in d["rccl"]["reason"] and d["rccl"]["devices"] == [0]
    fs = d["full_stream"]
    assert fs["cadence_frames"] == 6 and fs["frames"] == 6 * ranks and fs["merge_ms"] > 0 and fs["halo_exchange_ms"] > 0
    assert 0 < fs["frames_per_s"] < fs["fuse_only_frames_per_s"] and len(fs["sub_map_blocks_per_rank"]) == ranks
    assert d["merge"]["cadence_frames"] == 6 and d["merge"]["cadence_frames_full_stream"] == 6
    # `value` contains the sub-map merge and the boundary-block exchange (here through gloo and host tensors: slow)
    assert 0 < d["value"] < d["fuse_only_frames_per_s"] and "mrh_comm_merge_submaps" in d["value_definition"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - ranks) < 1e-6 * ranks
    m = d["merge"]
    assert m["merge_ms"] > 0 and len(m["blocks_sent_per_rank"]) == ranks and all(v > 0 for v in m["blocks_sent_per_rank"])
    assert all(v > 0 for v in m["owned_blocks_after_merge_per_rank"]) and all(v > 0 for v in m["halo_blocks_taken_per_rank"])
    assert d["tile_sharded"]["frames_per_s"] > 0
    assert d["roofline"]["launches"] == 6 and d["cpu_baseline"] is None
    assert d["phases"]["k_front_ms"] > 0 and d["phases"]["k_back_ms"] > 0
    # every engine of the N > 1 line after the first merge used to be built tile-sharded 1-of-N (a shared
    # Params object that Engine.set_sharding wrote into).  The full-stream leg must move blocks between all ranks, its sub-maps
    # must be whole sub-maps, and the roofline pass on rank 0 must be a whole frame's integration: rank 0's segment IS the
    # N = 1 line's stream, so the updated voxels per launch are the same number.
    assert all(v > 0 for v in fs["blocks_sent_per_rank"]) and all(v > 0 for v in fs["halo_blocks_taken_per_rank"])
    assert all(v > 3000 for v in fs["sub_map_blocks_per_rank"]), fs["sub_map_blocks_per_rank"]
    assert all(v > 3000 for v in d["config"]["sub_map_blocks_per_rank"])
    one = _single_line(6, 2, ("--cpu-frames", "2"))
    u1, un = one["roofline"]["updated_voxels_per_launch"], d["roofline"]["updated_voxels_per_launch"]
    assert abs(un - u1) <= 0.2 * u1 and un == u1, (un, u1)
    assert d["roofline"]["compact_blocks_per_launch"] == one["roofline"]["compact_blocks_per_launch"]


def _multi_one_rank(steps, warmup, env=None, full=40):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--multi", "--steps", str(steps), "--warmup", str(warmup), "--blocks", "65536"],
                       capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MRH_BENCH_FULL_STREAM=str(full), **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_one_rank_rccl_line():
    """bench.py --gpus N as the driver launches it (RANK / LOCAL_RANK / WORLD_SIZE in the environment), over RCCL behind the C
    ABI — with the one rank this box's one GPU allows: `--gpus 1 --multi` enters the N-rank function.  MRH_COMM_SELF_LOOP=1: the
    rank's own sub-map travels through ncclSend / ncclRecv to itself and the exchange is counted in `value` as at N > 1."""
    d = _multi_one_rank(12, 2, env={"MRH_COMM_SELF_LOOP": "1"})
    assert d["n_gpus"] == 1 and d["ranks"] == 1 and d["fuse_only_frames_per_s"] > 1000 and 100 < d["value"] < d["fuse_only_frames_per_s"]
    assert "mrh_comm_exchange_halo" in d["value_definition"]
    ph = d["phases"]
    assert ph["merge"]["pack_ms"] > 0 and ph["merge"]["unpack_ms"] > 0 and ph["halo"]["pack_ms"] > 0
    assert ph["starve_allreduce_count"] == 0  # one shard: nothing to reduce (tests/test_sharding_gpu.py drives the all-reduce)
    assert "backend rccl" in d["config"]["parallelism"]
    # what RCCL itself reports (mrh_comm_status): one rank on device 0, no asynchronous error before or after the phases
    rc = d["rccl"]
    assert rc["rccl_ranks"] == 1 == d["ranks"] and rc["ranks_seen_by_each_rank"] == [1] and rc["rccl_device_of_each_rank"] == [0] and rc["devices"] == [0]
    assert rc["async_error_code_of_each_rank"] == [0] and rc["async_error_after_phases"] == rc["async_error_after_init"] and rc["rccl_version"] > 0
    assert d["full_stream"]["cadence_frames"] == 40 and d["full_stream"]["frames_per_s"] > 0
    assert d["full_stream"]["sub_map_blocks_per_rank"][0] > 3000 and d["merge"]["owned_blocks_after_merge_per_rank"][0] > 3000


def test_one_rank_multi_line_agrees_with_the_single_gpu_line():
    """the driver divides value(N) by N x value(1), and value(1) comes from bench_single.  The N-rank
    function with ONE rank must therefore be the same job with the same number: same stream, same parameters, same timed loop,
    nothing to exchange — within 5 % at the driver's step counts (best of three pairs: both sides are 0.7 ms samples)."""
    best = None
    for _ in range(3):
        one = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-pmc", "--no-cpu", "--no-extras"])
        d = _multi_one_rank(20, 5, full=0)
        assert d["ranks"] == 1 and d["rccl"]["rccl_ranks"] == 1 and d["value_definition"].startswith("one rank")
        assert d["value"] == d["fuse_only_frames_per_s"] and d["config"]["workload"].startswith("replica-room0 stand-in 640x480")
        assert d["roofline"]["updated_voxels_per_launch"] == one["roofline"]["updated_voxels_per_launch"]
        assert d["config"]["sub_map_blocks_per_rank"] == [one["config"]["live_blocks_end"]]
        ratio = d["value"] / one["value"]
        best = ratio if best is None or abs(ratio - 1) < abs(best - 1) else best
        if abs(ratio - 1) <= 0.05:
            break
    assert abs(best - 1) <= 0.05, best


def test_value_survives_without_a_communicator():
    """The measured value must not depend on the exchange phases: with the file-based host group (what bench.py falls back to
    when the RCCL communicator cannot be created) two ranks report the frame-sharded value, the roofline of rank 0's
    segment, and say that the exchange phases were left out."""
    d = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--blocks", "65536"], env={"MRH_BENCH_SHARE_DEVICE": "1", "MRH_BENCH_BACKEND": "host"}, timeout=900)
    assert all(k in d for k in KEYS)
    assert d["ranks"] == 2 and d["value"] > 1000 and d["backend"] == "host" and d["value_definition"].startswith("fuse only")
    assert d["merge"] is None and d["tile_sharded"] is None and d["roofline"]["launches"] == 6
    assert len(d["config"]["sub_map_blocks_per_rank"]) == 2 and all(v > 0 for v in d["config"]["sub_map_blocks_per_rank"])


def test_value_survives_a_stuck_exchange_phase():
    """A watchdog ends the run with the value it has if the phases beside the value do not finish in time (forced here
    with a limit the merge of two sub-maps cannot meet)."""
    d = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--blocks", "65536"], env={"MRH_BENCH_SHARE_DEVICE": "1", "MRH_BENCH_PHASE_TIMEOUT": "0.001"}, timeout=900)
    assert d["ranks"] == 2 and d["value"] > 1000 and "phases_error" in d and d["merge"] is None


def test_the_timed_entry_point_of_bench_py_matches_the_oracle():
    """The timed region of bench.py feeds the engine through mrh_set_depth_device / mrh_set_rgb_device (caller-owned device
    pointers, `Resident.run`), not through the upload ring every other parity test uses.  Exactly that loop — the driver's 5
    warm-up + 20 timed frames of the 640x480 stream, one sync at the end — against the oracle fed the same frames from host
    memory: complete map and mesh (VoxelContainer::integrate as the reference brackets it, voxel_data_structures.cpp:90-110)."""
    sys.path.insert(0, ROOT)
    import bench
    import parity_utils as pu
    from mrhash_amd import capi, synth

    hip, orc = capi.load_hip(), pu.oracle_lib()
    W, K = 5, 20
    frames = bench.render_stream("replica", W + K)  # the very frames bench.py renders
    res = bench.Resident(frames, synth.REPLICA_640)
    a = bench.make_engine(hip, capi.Params(num_sdf_blocks=131072, device_id=0, **synth.REPLICA_PARAMS), synth.REPLICA_640)
    b = pu.make_engine(orc, synth.REPLICA_640, synth.REPLICA_PARAMS, 131072)
    res.run(a, 0, W)
    a.sync()
    res.run(a, W, W + K)  # the timed loop: no synchronisation between the frames
    a.sync()
    for f in frames:
        pu.feed(b, f)
    r = pu.compare_maps(a, b)
    assert r["blocks"] > 10000 and r["sdf_bit_exact"] and r["sumsq_bit_exact"]
    assert a.stats().frames_integrated == W + K and a.stats().error_flags == 0
    m = pu.compare_meshes(a, b)
    assert m["triangles"] > 300000 and m["pos_bit_exact"]
    a.close()
    b.close()
