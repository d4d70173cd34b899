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
    return json.loads(lines[0])


def test_single_gpu_line():
    d = _run(["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-pmc", "--cpu-frames", "2"])
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


def test_two_ranks_on_one_device_line():
    """The N > 1 code path (self-spawned ranks, frame-sharded value, merge, tile-sharded mode) with both ranks on device 0 over gloo."""
    d = _run(["--gpus", "2", "--steps", "6", "--warmup", "2"], env={"MRH_BENCH_SHARE_DEVICE": "1"})
    assert all(k in d for k in KEYS)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 1000
    assert d["merge"]["merge_ms"] > 0 and len(d["merge"]["blocks_sent_per_rank"]) == 2 and d["tile_sharded"]["frames_per_s"] > 0
    assert d["roofline"]["launches"] == 6 and d["cpu_baseline"] is None
