#!/usr/bin/env python3
"""bench.py — depth frames/s integrated at 640x480 (BASELINE.json metric) on N MI355X, with the HBM-roofline
fraction of the integrate kernel and the CPU restatement timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
      N > 1 without WORLD_SIZE in the environment: starts N copies of itself, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE set)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N = 1 — workload BASELINE.json configs[1]: "Replica room0" stand-in — analytic 6x3x4 m box room, orbit trajectory,
Replica intrinsics rescaled to 640x480, depth quantised to 1/6553.5 m, replica.cfg parameters (1 cm voxels, 7 cm
truncation, GC every frame, starve every 100th).  One step = one frame through the whole VoxelContainer::integrate
chain (allocate along rays -> frustum compaction -> depth->TSDF integrate -> GC).  `value` = frames/s with the frames
resident in HBM before the timed region starts.  Beside it, in the same JSON line:
  roofline            k_back (integrate + GC) on that workload: algorithmic bytes / HIP-event time of the launch; `traffic`
                      = HBM bytes per launch from rocprofv3 PMC passes of THIS workload run as sub-processes (null if
                      rocprofv3 is not available)
  roofline_hbm        the same kernel on a 2.5x larger room: the per-frame working set exceeds the 256 MiB Infinity Cache
  mc                  configs[2]: variance-adaptive multi-resolution map of the same stream + marching-cubes extraction,
                      roofline of the two k_mc launches (6144 B per fine block + 768 B per coarse block + 72 B per triangle)
  lidar               configs[4], LiDAR half: 128 x 1024-point scans along a street (vbr.cfg parameters): scans/s, points/s
  splat               configs[4], 3DGS half: splat seeds per frame (mrh_splat_seeds after every fused frame)
  pcie_inclusive_frames_per_s   the drop-in number: host numpy images -> mrh_upload_* every frame (never `value`)
  cpu_baseline        the oracle on a bounded sample of the same stream

N > 1 — workload BASELINE.json configs[3]: "ScanNet scene0000" stand-in (furnished 8x3x6 m room, hand-held walk).
  `value`: FRAME-SHARDED fusion + the exchange that makes the sub-maps one map, weak scaling: every rank fuses its own K-frame
           segment into its own sub-map (no collective per frame), then ONE mrh_comm_merge_submaps + ONE mrh_comm_exchange_halo;
           value = N*K / (max-over-ranks fusion time + merge time + halo time).  `fuse_only_frames_per_s` is the fusion alone.
  `merge`: the sub-maps are folded into ONE tile-sharded map (mrh_comm_merge_submaps: all-to-all of blocks to
           their tile owner over RCCL + weighted merge on the device) and the boundary blocks exchanged
           (mrh_comm_exchange_halo): times, bytes and per-phase HIP-event times (pack, counts, collective, unpack) of both.
  `tile_sharded`: the result-identical mode on the same K frames (rank 0's segment, seen by every rank; starve frames
           run their MIN all-reduce inside mrh_integrate): frames/s = K / max-over-ranks time (strong scaling).

No torch on this path: device buffers through mrhash_amd.hipmem (hipMalloc on the runtime the library is bound to), the
communicator through include/mrhash_comm.h (RCCL on the library's own stream) — one HIP runtime per rank.  The only
exception is the test mode MRH_BENCH_SHARE_DEVICE=1 (N ranks on ONE device: RCCL refuses that, the ranks talk over gloo).
"""
from __future__ import annotations

import argparse
import gc
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MALL_BYTES = 256 << 20

_RESULT_FD = None  # the process's original stdout, once main() has pointed fd 1 at stderr


def emit(out: dict) -> None:
    """The ONE JSON line of the run, on the real stdout.  Everything else that lands on fd 1 during the run — gloo's
    connection banner, anything a runtime library prints — has been pointed at stderr by main(), so the line stands alone."""
    line = (json.dumps(out) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=262144, help="SDF block pool capacity (1.6 GB at 262144)")
    ap.add_argument("--cpu-frames", type=int, default=64, help="frames of the same stream timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC sub-process passes (roofline.traffic = null)")
    ap.add_argument("--no-extras", action="store_true", help="N=1: only the headline pass and its roofline (no mc / hbm / pcie legs)")
    ap.add_argument("--pmc-inner", action="store_true", help=argparse.SUPPRESS)  # the workload alone, under rocprofv3
    ap.add_argument("--pmc-inner-big", action="store_true", help=argparse.SUPPRESS)  # the roofline_hbm workload alone
    ap.add_argument("--pmc-inner-mc", action="store_true", help=argparse.SUPPRESS)  # the multi-resolution map + two extractions
    ap.add_argument("--pmc-inner-lidar", action="store_true", help=argparse.SUPPRESS)  # the LiDAR scans alone
    ap.add_argument("--pmc-inner-sph", action="store_true", help=argparse.SUPPRESS)  # the spherical images alone
    ap.add_argument("--frames-cache", default="", help=argparse.SUPPRESS)
    ap.add_argument("--multi", action="store_true", help="--gpus 1 through the N-rank function (one rank, RCCL communicator of size 1): its value must "
                                                         "equal the N = 1 line's, which is what makes value(N) / (N x value(1)) meaningful")
    return ap.parse_args()


# ---- workload helpers --------------------------------------------------------------------------------------------

def render_stream(kind: str, n: int, start: int = 0, cache: str = ""):
    """[Frame] of the named stand-in stream, frames [start, start + n)."""
    from mrhash_amd import synth

    if cache and os.path.exists(cache):
        z = np.load(cache)
        return [synth.Frame(z["t"][i], z["q"][i], z["R"][i], z["depth"][i], z["rgb"][i]) for i in range(len(z["t"]))]
    if kind == "replica":
        scene, K, poses, scaling = synth.replica_room(), synth.REPLICA_640, synth.orbit_poses(start + n)[start:], 6553.5
    elif kind == "replica_big":  # the same room scaled 2.5x, the orbit with it: 6.25x the surface in view
        scene = synth.Scene(synth.Box((-7.5, -3.75, -5.0), (7.5, 3.75, 5.0)), seed=0)
        K, poses, scaling = synth.REPLICA_640, synth.orbit_poses(start + n, radius=2.5)[start:], 6553.5
    elif kind == "scannet":
        scene, K, poses, scaling = synth.scannet_room(), synth.SCANNET, synth.walk_poses(start + n, seed=0)[start:], 5000.0
    else:
        raise ValueError(kind)
    frames = [synth.render(scene, K, t, q, depth_scaling=scaling) for t, q in poses]
    if cache:
        np.savez(cache, t=np.stack([f.t for f in frames]), q=np.stack([f.q for f in frames]), R=np.stack([f.R for f in frames]),
                 depth=np.stack([f.depth for f in frames]), rgb=np.stack([f.rgb for f in frames]))
    return frames


class Resident:
    """A stream uploaded to HBM once; `run` feeds frames [lo, hi) through the zero-copy setters."""

    def __init__(self, frames, K):
        from mrhash_amd import hipmem

        self.frames, self.K = frames, K
        self.depth = hipmem.DeviceBuffer.from_numpy(np.stack([f.depth for f in frames]))
        self.rgb = hipmem.DeviceBuffer.from_numpy(np.stack([f.rgb for f in frames]))
        hipmem.synchronize()
        self.ds, self.rs = K.rows * K.cols * 4, K.rows * K.cols * 3

    def run(self, engine, lo, hi, integrate=None):
        K = self.K
        for i in range(lo, hi):
            f = self.frames[i]
            engine.set_pose(f.R, f.t)
            engine.set_depth_device(self.depth.ptr + i * self.ds, K.rows, K.cols)
            engine.set_rgb_device(self.rgb.ptr + i * self.rs, K.rows, K.cols)
            if integrate is None:
                engine.integrate()
            else:
                integrate(engine)


def make_engine(hip, params, K):
    from mrhash_amd import capi

    e = capi.Engine(hip, params)
    e.set_camera(K.fx, K.fy, K.cx, K.cy, K.rows, K.cols, params.min_depth, params.max_depth)
    return e


def profiled_roofline(eng, res: Resident, W: int, total: int, label: str):
    """Second pass over the same frames with the event pair attached to every k_back launch and the device-side U / M
    counters (12 B read + 12 B write per updated voxel, 24 B per compact block, 7 B per pixel)."""
    eng.reset()
    res.run(eng, 0, W)
    eng.sync()
    eng.set_profile(True)
    s0 = eng.stats()
    t1 = time.perf_counter()
    res.run(eng, W, total)
    eng.sync()
    prof_elapsed = time.perf_counter() - t1
    s1 = eng.stats()
    eng.set_profile(False)
    n_k = int(s1.n_integrate_kernel - s0.n_integrate_kernel)
    k_ms = float(s1.sum_integrate_kernel_ms - s0.sum_integrate_kernel_ms) / max(n_k, 1)
    n_f = int(s1.n_front_kernel - s0.n_front_kernel)
    f_ms = float(s1.sum_front_kernel_ms - s0.sum_front_kernel_ms) / max(n_f, 1)
    U = (int(s1.total_updated_voxels) - int(s0.total_updated_voxels)) / max(n_k, 1)
    M = (int(s1.total_compact_blocks) - int(s0.total_compact_blocks)) / max(n_k, 1)
    alg = 24.0 * U + 24.0 * M + 7 * res.K.rows * res.K.cols
    achieved = alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    touched = (U / 512.0) * 6144.0  # payload of the block-equivalents the launch rewrites
    return {"bound": "hbm", "kernel": "k_back (depth->TSDF integrate + GC summary + GC decision)", "workload": label,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": alg, "kernel_ms_avg": k_ms, "launches": n_k,
            "k_front_ms_avg": f_ms, "updated_voxels_per_launch": U,
            "compact_blocks_per_launch": M, "working_set_bytes_per_frame": M * 6144.0 + 7 * res.K.rows * res.K.cols,
            "rewritten_payload_bytes_per_frame": touched, "exceeds_infinity_cache": bool(M * 6144.0 > MALL_BYTES),
            "profiled_pass_ms_per_step": prof_elapsed / max(total - W, 1) * 1e3}


def pmc_traffic(args, kernel_prefix, cache: str, inner: str = "--pmc-inner", steps: int = 0, warmup: int = 0, per_run_of: int = 0):
    """HBM bytes per launch of `kernel_prefix` from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs,
    no trace flags beside --pmc) of this workload run as a sub-process.  Units are KiB; on gfx950 FETCH_SIZE counts a
    wide coalesced read at half its bytes, so it is doubled (MI355X_MICROARCH.md, HBM section)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    import csv

    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mrh_pmc_", dir="/tmp")
        steps, warmup = steps or args.steps, warmup or args.warmup
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), inner,
               "--steps", str(steps), "--warmup", str(warmup), "--blocks", str(args.blocks), "--frames-cache", cache]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=180)
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"rocprofv3 --pmc {counter} timed out"
        rows = []
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                name = row["Kernel_Name"].replace("void ", "")
                if name.startswith(kernel_prefix) and row["Counter_Name"] == counter:
                    rows.append((int(row.get("Dispatch_Id", len(rows))), float(row["Counter_Value"])))
        shutil.rmtree(d, ignore_errors=True)
        rows.sort()
        if per_run_of:  # every dispatch of the prefix counts; the figure is per run (e.g. the count + emit launches of ONE extraction)
            tot, n = sum(v for _, v in rows) / per_run_of, len(rows)
            rows = [(0, tot)] if n else []
        else:
            rows = rows[warmup:] if len(rows) > warmup else rows  # the timed frames only (the map still grows during the warm-up)
        tot, n = sum(v for _, v in rows), len(rows)
        if n == 0:
            return None, f"rocprofv3 --pmc {counter}: no dispatch of {kernel_prefix} recorded (rc {r.returncode}: {r.stderr[-200:]})"
        vals[counter] = (tot / n, n)
    bytes_per_launch = (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
    return bytes_per_launch, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE sub-process passes of the same workload, "
                              f"{vals['FETCH_SIZE'][1]} dispatches, 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction)")


LIDAR_SCANS, LIDAR_WARMUP = 25, 5


def lidar_setup(hip, blocks):
    """configs[4], LiDAR half: engine + the 128 x 1024 scans of the street resident in HBM + the loop that feeds them."""
    from mrhash_amd import capi, hipmem, synth

    n_scans = LIDAR_SCANS
    lcache = os.path.join(tempfile.gettempdir(), f"mrh_bench_vbr_{n_scans}.npz")
    poses = synth.drive_poses(n_scans, step=0.5)
    if os.path.exists(lcache):
        scans = list(np.load(lcache)["scans"])
    else:
        scene = synth.street_canyon()
        scans = [synth.lidar_scan(scene, t, q, rows=128, cols=1024) for t, q in poses]
        np.savez(lcache, scans=np.stack(scans))
    d_scans = [hipmem.DeviceBuffer.from_numpy(np.ascontiguousarray(sc, dtype=np.float32)) for sc in scans]
    le = capi.Engine(hip, capi.Params(num_sdf_blocks=blocks, device_id=0, **synth.VBR_PARAMS))
    le.set_camera(1, 1, 0, 0, 1, 1, 0.2, 100.0, model=1)
    le.set_scan_layout(1024)  # the scans lie in HBM (not looked at): 128 rows of 1 024 points, as the sensor delivers them

    def run_scans(lo, hi):
        for i in range(lo, hi):
            t, q = poses[i]
            le.set_pose(synth.quat_to_rot(q), t)
            le.set_points_device(d_scans[i].ptr, len(scans[i]))
            le.integrate_points()

    return le, scans, d_scans, run_scans


SPH_IMAGES, SPH_WARMUP = 24, 4


def spherical_setup(hip, blocks):
    """The image path under the spherical camera model: engine + 128 x 1024 range images of the street drive resident in HBM + the
    loop that feeds them."""
    from mrhash_amd import capi, hipmem, synth

    cam = synth.spherical_camera(128, 1024)
    n_img, w_img = SPH_IMAGES, SPH_WARMUP
    scache = os.path.join(tempfile.gettempdir(), f"mrh_bench_sph_{n_img}.npz")
    sposes = synth.drive_poses(n_img, step=0.5)
    if os.path.exists(scache):
        z = np.load(scache)
        imgs = [(z["d"][i], z["c"][i]) for i in range(n_img)]
    else:
        scene = synth.street_canyon()
        imgs = [synth.spherical_range_image(scene, t, q, cam) for t, q in sposes]
        np.savez(scache, d=np.stack([a for a, _ in imgs]), c=np.stack([b for _, b in imgs]))
    dd = hipmem.DeviceBuffer.from_numpy(np.stack([a for a, _ in imgs]).astype(np.float32))
    dc = hipmem.DeviceBuffer.from_numpy(np.stack([b for _, b in imgs]).astype(np.uint8))
    sp = dict(synth.VBR_PARAMS, n_frames_invalidate_voxels=100)
    se_ = capi.Engine(hip, capi.Params(num_sdf_blocks=blocks, device_id=0, **sp))
    se_.set_camera(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["rows"], cam["cols"], sp["min_depth"], 100.0, model=1)
    npx = cam["rows"] * cam["cols"]

    def run_imgs(lo, hi):
        for i in range(lo, hi):
            t, q = sposes[i]
            se_.set_pose(synth.quat_to_rot(q), t)
            se_.set_depth_device(dd.ptr + i * npx * 4, cam["rows"], cam["cols"])
            se_.set_rgb_device(dc.ptr + i * npx * 3, cam["rows"], cam["cols"])
            se_.integrate()

    return se_, run_imgs, n_img, w_img, npx, (dd, dc)


# ---- N = 1 ---------------------------------------------------------------------------------------------------------

def bench_single(args):
    from mrhash_amd import capi, hipmem, synth

    K, W = args.steps, args.warmup
    total = W + K
    hip = capi.load_hip()  # no fallback: raises if the HIP library is missing
    Kc = synth.REPLICA_640
    params = capi.Params(num_sdf_blocks=args.blocks, device_id=0, **synth.REPLICA_PARAMS)
    cache = args.frames_cache or os.path.join(tempfile.gettempdir(), f"mrh_bench_replica_{total}.npz")
    if args.pmc_inner_big:  # the roofline_hbm workload alone (under rocprofv3 --pmc)
        big = render_stream("replica_big", total, cache=cache)
        rb = Resident(big, Kc)
        be = make_engine(hip, capi.Params(num_sdf_blocks=max(args.blocks, 786432), device_id=0, **synth.REPLICA_PARAMS), Kc)
        rb.run(be, 0, total)
        be.sync()
        be.close()
        return
    if args.pmc_inner_sph:  # the spherical images alone (under rocprofv3 --pmc): warm-up + timed images, once
        se_, run_imgs, n_img, w_img, npx, _bufs = spherical_setup(hip, args.blocks)
        run_imgs(0, n_img)
        se_.sync()
        se_.close()
        return
    if args.pmc_inner_lidar:  # configs[4]: the scans alone (under rocprofv3 --pmc)
        le, scans, d_scans, run_scans = lidar_setup(hip, args.blocks)
        run_scans(0, LIDAR_SCANS)
        le.sync()
        le.close()
        return
    frames = render_stream("replica", total, cache=cache)
    res = Resident(frames, Kc)

    if args.pmc_inner_mc:  # configs[2]: the multi-resolution map, then two extractions (under rocprofv3 --pmc)
        mp = capi.Params(num_sdf_blocks=args.blocks, device_id=0, **dict(synth.REPLICA_PARAMS, sdf_var_threshold=0.005))
        me = make_engine(hip, mp, Kc)
        res.run(me, 0, total)
        me.sync()
        me.extract_triangles(soup=False)
        me.extract_triangles(soup=False)
        me.close()
        return

    if args.pmc_inner:  # the timed workload alone (under rocprofv3 --pmc)
        eng = make_engine(hip, params, Kc)
        res.run(eng, 0, total)
        eng.sync()
        eng.close()
        return

    # ---- pass A: the timed region (profile hooks off)
    eng = make_engine(hip, params, Kc)
    res.run(eng, 0, W)
    eng.sync()
    hipmem.synchronize()  # the whole device, not only the library's stream
    t0 = time.perf_counter()
    res.run(eng, W, total)
    eng.sync()
    hipmem.synchronize()
    elapsed = time.perf_counter() - t0
    st = eng.stats()
    occupied = int(st.occupied_fine)
    table = {"hash_slots": int(st.hash_slots), "tombstones": int(st.tombstones), "max_probe_length": int(st.max_probe_length), "rehash_count": int(st.rehash_count)}
    # the map the timed loop produced, kept for the check against the oracle in the cpu_baseline leg (which runs the oracle over
    # the same frames anyway): taken now, before the profiled pass resets the engine
    want_check = not args.no_cpu and args.cpu_frames > 0
    timed_map = eng.dump_blocks() if want_check and min(args.cpu_frames, total) == total else None

    # ---- the boundary as the reference uses it: host buffers -> mrh_upload_depth / mrh_upload_rgb each frame.  Runs BEFORE
    # any profiled pass: launches that carry start / stop events switch the queue to profiling mode, which slows every
    # later dispatch of the process.
    pcie_fps = None
    link = None
    if not args.no_extras:
        pe = make_engine(hip, params, Kc)
        for i in range(W):
            f = frames[i]
            pe.set_pose(f.R, f.t); pe.upload_depth(f.depth); pe.upload_rgb(f.rgb); pe.integrate()
        pe.sync()
        # at least 100 frames (the timed frames again and again when --steps is smaller): a host-fed loop needs a few frames to
        # reach its cadence (helper threads awake, rings turning), and 20 frames are 1.3 ms
        pcie_passes = max(1, -(-100 // K))
        t2 = time.perf_counter()
        for _ in range(pcie_passes):
            for i in range(W, total):
                f = frames[i]
                pe.set_pose(f.R, f.t); pe.upload_depth(f.depth); pe.upload_rgb(f.rgb); pe.integrate()
        pe.sync()
        pcie_fps = pcie_passes * K / (time.perf_counter() - t2)
        pe.close()
        # ... and what the link gives those 2.15 MB per frame at best (pinned -> device, one stream per image, nothing else on the
        # device), measured in this run: the ceiling of any per-frame host hand-over
        link = hipmem.h2d_link_rate()
        # ... and what it gives them NEXT TO the kernels of resident frames (a second context fusing the stream on another host
        # thread meanwhile): the transfers of a host-fed loop share HBM and the fabric with the integration of the frame before
        import threading

        le2 = make_engine(hip, params, Kc)
        res.run(le2, 0, W)
        le2.sync()
        stop_load = [False]

        def _load():
            while not stop_load[0]:
                res.run(le2, W, total)

        th = threading.Thread(target=_load)
        th.start()
        time.sleep(0.003)
        link_loaded = hipmem.h2d_link_rate(frames=200)
        stop_load[0] = True
        th.join()
        le2.sync()
        le2.close()
        link["gbs_next_to_resident_frames"] = link_loaded["gbs"]

    # Launches that carry events switch the process's queues to their profiling mode, which slows every later dispatch (37.8-43.7 us
    # per spherical image behind the profiled extraction where tools/bench_spherical.py measures 27-29 in a process that never
    # profiled): every leg times first, and the parts that need profile mode (kernel times, update counters) run after ALL timed
    # regions, in the order they were queued here.
    deferred = []
    # ---- configs[2]: multi-resolution map of the same stream + marching cubes (timed before the profiled passes, too)
    mc = None
    if not args.no_extras:
        mp = capi.Params(num_sdf_blocks=args.blocks, device_id=0, **dict(synth.REPLICA_PARAMS, sdf_var_threshold=0.005))
        me = make_engine(hip, mp, Kc)
        res.run(me, 0, W)
        me.sync()
        t3 = time.perf_counter()
        res.run(me, W, total)
        me.sync()
        mr_elapsed = time.perf_counter() - t3
        first_ms = None
        for _ in range(3):  # warm: allocations, growth of the device scratch, first touch of the host result buffers
            t4 = time.perf_counter()
            me.extract_triangles(soup=False)
            if first_ms is None:
                first_ms = (time.perf_counter() - t4) * 1e3  # what GeoWrapper::extractMesh usually is: a context's FIRST extraction
        ext = []
        for _ in range(5):
            t4 = time.perf_counter()
            ntri = me.extract_triangles(soup=False)
            ext.append((time.perf_counter() - t4) * 1e3)
        extract_ms = float(np.median(ext))
        ms0 = me.stats()
        alg_mc = 6144.0 * int(ms0.occupied_fine) + 768.0 * int(ms0.occupied_coarse) + 72.0 * ntri
        mc = {"workload": "replica-room0 stand-in 640x480, sdf_var_threshold 0.005 (configs[2]): multi-resolution fusion, then extraction",
              "multires_frames_per_s": K / mr_elapsed, "multires_ms_per_step": mr_elapsed / K * 1e3,
              "fine_blocks": int(ms0.occupied_fine), "coarse_blocks": int(ms0.occupied_coarse), "triangles": int(ntri),
              "extract_ms_in_library": extract_ms, "extract_ms_runs": ext, "first_extract_ms": first_ms, "k_mc_count_ms": None, "k_mc_emit_ms": None,
              "roofline": {"bound": "hbm", "kernel": "k_mc<count> + k_mc_emit_records", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": None, "traffic": None, "algorithmic_bytes": alg_mc,
                           "note": "6144 B per fine block + 768 B per coarse block read once + 72 B per triangle written; the count pass parks 72 B of corner values per productive voxel for the emit pass (in `traffic`, not in the algorithmic bytes); latency / issue-bound, far below the HBM roof"}}

        def mc_profiled(me=me, mc=mc, alg_mc=alg_mc):  # kernel times come from one more extraction, in profile mode (see `deferred`)
            me.set_profile(True)
            me.extract_triangles(soup=False)
            ms = me.stats()
            me.set_profile(False)
            mc_ms = float(ms.last_mc_count_ms + ms.last_mc_emit_ms)
            ach = alg_mc / (mc_ms * 1e-3) / 1e9 if mc_ms > 0 else 0.0
            mc["k_mc_count_ms"], mc["k_mc_emit_ms"] = float(ms.last_mc_count_ms), float(ms.last_mc_emit_ms)
            mc["roofline"]["achieved"], mc["roofline"]["frac"] = ach, ach / HBM_PEAK_GBS
            me.close()
        deferred.append(mc_profiled)

    # ---- configs[4], LiDAR half: 128 x 1024 scans along a street (vbr.cfg parameters), scans resident in HBM
    lidar = None
    if not args.no_extras:
        n_scans, w_scans = LIDAR_SCANS, LIDAR_WARMUP
        le, scans, d_scans, run_scans = lidar_setup(hip, args.blocks)

        run_scans(0, w_scans)
        le.sync()
        t6 = time.perf_counter()
        run_scans(w_scans, n_scans)
        le.sync()
        dt = time.perf_counter() - t6
        npts = int(sum(len(sc) for sc in scans[w_scans:]))
        live_end = int(le.stats().occupied_fine)
        us_scan = dt / (n_scans - w_scans) * 1e6
        lidar = {"workload": "VBR stand-in (configs[4], LiDAR half): 128 x 1024 scans along a 100 m street, vbr.cfg parameters (voxel 0.20 m, "
                             "truncation 0.40 m, projective SDF), scans resident in HBM",
                 "scan_layout": "organised, 1 024 points per row, said through mrh_set_scan_layout (clouds in device memory are not looked at; host clouds are: mrh_detect_scan_layout)",
                 "scans_per_s": (n_scans - w_scans) / dt, "us_per_scan": us_scan, "points_per_s": npts / dt,
                 "points_per_scan": int(len(scans[0])), "live_blocks_end": live_end,
                 "roofline": {"bound": "hbm", "kernel": "one scan: k_alloc3d + k_scan_walk + k_scan_offsets + k_scan_place + k_scan_apply (mrh_scan.h: voxel buckets, no sort)", "achieved": None,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                              "algorithmic_bytes_per_scan": None, "updated_voxels_per_scan": None,
                              "note": "12 B per point read + 24 B per updated voxel (12 B read + 12 B write); a scan is a chain of five latency-bound "
                                      "launches over ~10^5 points and ~10^6 records (each record written twice and read twice), two orders of "
                                      "magnitude below the HBM roof"}}

        def lidar_profiled(le=le, lidar=lidar, scans=scans, d_scans=d_scans, run_scans=run_scans, us_scan=us_scan):
            # voxels a scan updates (one per (voxel, scan) pair: the runs k_scan_apply folds), counted in a second pass in profile mode
            le.reset()
            le.set_profile(True)
            run_scans(0, w_scans)
            u0 = int(le.stats().total_updated_voxels)
            run_scans(w_scans, n_scans)
            upd = (int(le.stats().total_updated_voxels) - u0) / (n_scans - w_scans)
            le.set_profile(False)
            alg_l = 12.0 * len(scans[0]) + 24.0 * upd
            ach_l = alg_l / (us_scan * 1e-6) / 1e9
            r = lidar["roofline"]
            r["achieved"], r["frac"], r["algorithmic_bytes_per_scan"], r["updated_voxels_per_scan"] = ach_l, ach_l / HBM_PEAK_GBS, alg_l, upd
            le.close()
        deferred.append(lidar_profiled)

    # ---- configs[4], 3DGS half: splat seeds of every frame (quad-tree over the colour image + one map lookup per leaf), the
    # blocking call GeoWrapper::compute makes after the fusion of a frame when a gs_optimization_param_path is set
    splat = None
    if not args.no_extras:
        se = make_engine(hip, params, Kc)
        n_seed_frames = min(total, 40)
        res.run(se, 0, 2)
        se.sync()
        t_seed, n_seeds, n_leaves = 0.0, 0, 0
        t7 = time.perf_counter()
        for i in range(2, n_seed_frames):
            res.run(se, i, i + 1)
            c0 = time.perf_counter()
            seeds = se.splat_seeds(0.1, 1)  # qtree_thresh, qtree_min_pixel_size of the reference's params.json
            t_seed += time.perf_counter() - c0
            n_seeds += len(seeds)
        se.sync()
        dt7 = time.perf_counter() - t7
        nf = n_seed_frames - 2
        n_leaves = len(se.qtree_leaves()) * nf  # of the last frame (outside the timed loop: a 35 k-row host copy)
        splat = {"workload": "3DGS splat initialisation (configs[4], second half) on the 640x480 stream: mrh_integrate + mrh_splat_seeds per frame",
                 "frames_per_s_with_seeding": nf / dt7, "seed_call_us": t_seed / nf * 1e6, "leaves_per_frame": n_leaves / nf, "seeds_per_frame": n_seeds / nf}
        se.close()

    # ---- what the timed region may not contain: the starve frame (every 100th, replica.cfg) and the table census (every 64th).
    # Single frames bracketed by synchronisations (so each figure carries one sync of overhead, the steady one too): the census
    # falls on the 65th frame by itself; the starve sequence is triggered on the frame after the loop by passing its own index
    # as the period (mrh_integrate(n): starve when frames % n == 0).
    periodic = None
    if not args.no_extras:
        pf = make_engine(hip, params, Kc)
        n_leg = min(total, 70)
        per = []
        for i in range(n_leg):
            pf.sync()
            c0 = time.perf_counter()
            res.run(pf, i, i + 1)
            pf.sync()
            per.append((time.perf_counter() - c0) * 1e3)
        st_ms = None
        if n_leg < total or total > 1:
            j = n_leg % total
            pf.sync()
            c0 = time.perf_counter()
            res.run(pf, j, j + 1, integrate=lambda e: e.integrate(n_leg))  # frames == n_leg here: the starve frame of period n_leg
            pf.sync()
            st_ms = (time.perf_counter() - c0) * 1e3
        steady = float(np.median(per[8:64])) if n_leg > 16 else float(np.median(per))
        census = per[64] if n_leg > 64 else None
        # what a starve frame costs INSIDE the pipeline (round 6: it no longer flushes it): the same frames twice from the same start,
        # no synchronisation inside, the second pass with one starve frame in the middle; median of three differences
        never = 1 << 30
        lo_p, hi_p = min(5, total // 4), total
        mid_p = (lo_p + hi_p) // 2

        def pipelined_pass(starve_at):
            pf.reset()
            res.run(pf, 0, lo_p, integrate=lambda e: e.integrate(never))
            pf.sync()
            c1 = time.perf_counter()
            for i in range(lo_p, hi_p):
                res.run(pf, i, i + 1, integrate=(lambda e, i=i: e.integrate(i)) if i == starve_at else (lambda e: e.integrate(never)))
            pf.sync()
            return (time.perf_counter() - c1) * 1e3

        extra = None
        if hi_p - lo_p >= 8:
            pipelined_pass(-1)
            diffs = sorted(pipelined_pass(mid_p) - pipelined_pass(-1) for _ in range(3))
            extra = diffs[1]
        periodic = {"what": "single frames, each bracketed by mrh_sync (one synchronisation of overhead in every figure)",
                    "steady_frame_ms": steady, "census_frame_ms": census, "starve_frame_ms": st_ms, "census_period": 64,
                    "starve_period": int(synth.REPLICA_PARAMS["n_frames_invalidate_voxels"]),
                    "starve_frame_extra_ms_in_pipeline": extra, "starve_frame_extra_what": f"{hi_p - lo_p} resident frames without a synchronisation, "
                    f"with and without a starve frame at frame {mid_p}: difference of the two wall times (median of three)",
                    "amortised_extra_ms_per_frame": (((census - steady) / 64 if census else 0.0) +
                                                     ((st_ms - steady) / synth.REPLICA_PARAMS["n_frames_invalidate_voxels"] if st_ms else 0.0))}
        pf.close()

    # ---- the image path under the SPHERICAL camera model (general kernels, mrh_softmath.h): 128 x 1024 range images of the street
    spherical = None
    if not args.no_extras:
        se_, run_imgs, n_img, w_img, npx, _sph_buffers = spherical_setup(hip, args.blocks)

        # three passes over the same drive (the map is reset in between), the median reported: the first pass of a context also
        # pays for the first use of the spherical instantiations and of the pool (r04: 37.8-43.7 us over 9 frames where
        # tools/bench_spherical.py measured 27-29 on later passes)
        pass_s = []
        for _ in range(3):
            se_.reset()
            run_imgs(0, w_img)
            se_.sync()
            c0 = time.perf_counter()
            run_imgs(w_img, n_img)
            se_.sync()
            pass_s.append(time.perf_counter() - c0)
        dts = sorted(pass_s)[1]
        live_sph = int(se_.stats().occupied_fine)
        spherical = {"workload": "128 x 1024 range images of the street scene through mrh_integrate under the spherical camera model "
                                 "(vbr.cfg parameters; the two launches k_front / k_back templated on the camera model, pipelined)",
                     "frames_per_s": (n_img - w_img) / dts, "ms_per_frame": dts / (n_img - w_img) * 1e3, "live_blocks_end": live_sph,
                     "frames_timed": n_img - w_img, "passes_ms_per_frame": [x / (n_img - w_img) * 1e3 for x in pass_s],
                     "roofline": {"bound": "hbm", "kernel": "k_back<spherical>", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                                  "traffic": None, "algorithmic_bytes_per_launch": None, "kernel_ms_avg": None, "launches": None,
                                  "updated_voxels_per_launch": None, "compact_blocks_per_launch": None,
                                  "note": "24 B per updated voxel + 24 B per compact block + 7 B per pixel; every voxel projects through sqrt / atan2 / asin of "
                                          "the shared fp32 library (mrh_softmath.h): the kernel is arithmetic-bound far below the HBM roof"}}

        def spherical_profiled(se_=se_, spherical=spherical, run_imgs=run_imgs, keep=_sph_buffers):
            # the integrate kernel of these frames against the HBM roof, as for the pinhole stream: profiled pass, U and M from the device
            se_.reset()
            run_imgs(0, w_img)
            se_.sync()
            se_.set_profile(True)
            q0 = se_.stats()
            run_imgs(w_img, n_img)
            se_.sync()
            q1 = se_.stats()
            se_.set_profile(False)
            nk = max(int(q1.n_integrate_kernel - q0.n_integrate_kernel), 1)
            kms = float(q1.sum_integrate_kernel_ms - q0.sum_integrate_kernel_ms) / nk
            Us = (int(q1.total_updated_voxels) - int(q0.total_updated_voxels)) / nk
            Ms = (int(q1.total_compact_blocks) - int(q0.total_compact_blocks)) / nk
            alg_s = 24.0 * Us + 24.0 * Ms + 7.0 * npx
            ach_s = alg_s / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
            spherical["roofline"].update({"achieved": ach_s, "frac": ach_s / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_s, "kernel_ms_avg": kms,
                                          "launches": nk, "updated_voxels_per_launch": Us, "compact_blocks_per_launch": Ms})
            se_.close()
        deferred.append(spherical_profiled)

    # ---- the parts of the legs above that need profile mode, after every timed region of this process
    for fn in deferred:
        fn()
    deferred.clear()

    # ---- pass B: same frames, HIP events around every integrate-kernel launch + device-side U/M counters
    roof = profiled_roofline(eng, res, W, total, "configs[1] (value's workload)")
    eng.close()
    roof["schedule"] = ("pipelined, as in the timed region: the front half of frame g + 1 (k_front, second stream) runs next to this kernel, "
                        "so its duration includes the share of the machine it gives away; `serial` = the same kernel with MRH_PIPE=0 "
                        "(two serial launches per frame: the kernel alone on the chip)")
    if os.environ.get("MRH_PIPE", "1") != "0":
        os.environ["MRH_PIPE"] = "0"
        try:
            se0 = make_engine(hip, params, Kc)
        finally:
            del os.environ["MRH_PIPE"]
        rs = profiled_roofline(se0, res, W, total, "configs[1], serial launches")
        se0.close()
        roof["serial"] = {k: rs[k] for k in ("achieved", "frac", "kernel_ms_avg", "k_front_ms_avg", "algorithmic_bytes_per_launch", "launches", "profiled_pass_ms_per_step")}
    # the whole frame against the same roof: integrate bytes + what the front half moves (4 B depth + 3 B colour read and 8 B written per
    # pixel, 24 B per descriptor swept) over the frame time of the timed region
    frame_bytes = roof["algorithmic_bytes_per_launch"] + 15.0 * Kc.rows * Kc.cols + 24.0 * occupied
    roof["frame"] = {"algorithmic_bytes_per_frame": frame_bytes, "achieved": frame_bytes / (elapsed / K) / 1e9, "frac": frame_bytes / (elapsed / K) / 1e9 / HBM_PEAK_GBS,
                     "note": "k_back's algorithmic bytes + 15 B per pixel (front half: 7 read, 8 written) + 24 B per live block (descriptor sweep) over ms_per_step"}
    roof["cache_note"] = "the per-frame working set sits inside the 256 MiB Infinity Cache between frames: see roofline_hbm for the same kernel outside it"

    # ---- the same kernel with a working set above the Infinity Cache
    roof_hbm = None
    if not args.no_extras:
        nb = min(total, 60)
        wb = min(W, 10)
        big_cache = os.path.join(tempfile.gettempdir(), f"mrh_bench_replica_big_{nb}.npz")
        big = render_stream("replica_big", nb, cache=big_cache)
        rb = Resident(big, Kc)
        bp = capi.Params(num_sdf_blocks=max(args.blocks, 786432), device_id=0, **synth.REPLICA_PARAMS)
        be = make_engine(hip, bp, Kc)
        rb.run(be, 0, wb)
        be.sync()
        t5 = time.perf_counter()
        rb.run(be, wb, nb)
        be.sync()
        big_fps = (nb - wb) / (time.perf_counter() - t5)
        roof_hbm = profiled_roofline(be, rb, wb, nb, "the same room scaled 2.5x (15 x 7.5 x 10 m, orbit radius 2.5 m): depths 6.5-7.5 m")
        roof_hbm["frames_per_s"] = big_fps
        roof_hbm["live_blocks_end"] = int(be.stats().occupied_fine)
        be.close()
        del rb
        if not args.no_pmc:
            roof_hbm["traffic"], roof_hbm["traffic_note"] = pmc_traffic(args, "mrh::k_back<true, false", big_cache, "--pmc-inner-big", nb - wb, wb)

    if mc is not None and not args.no_pmc:  # HBM bytes of the two k_mc launches of one extraction (two extractions in the sub-process)
        mc["roofline"]["traffic"], mc["roofline"]["traffic_note"] = pmc_traffic(args, ("mrh::k_mc<", "mrh::k_mc_emit_records"), cache, "--pmc-inner-mc", per_run_of=2)

    if spherical is not None and not args.no_pmc:  # HBM bytes of the spherical integration launches (the timed images of one pass)
        spherical["roofline"]["traffic"], spherical["roofline"]["traffic_note"] = pmc_traffic(args, "mrh::k_back<", cache, "--pmc-inner-sph", steps=SPH_IMAGES - SPH_WARMUP, warmup=SPH_WARMUP)
    if lidar is not None and not args.no_pmc:  # HBM bytes of all kernels of a scan (every mrh:: launch of the sub-process / scans)
        lidar["roofline"]["traffic"], lidar["roofline"]["traffic_note"] = pmc_traffic(args, ("mrh::k_alloc3d", "mrh::k_points_", "mrh::k_sort_", "mrh::k_scan_"), cache, "--pmc-inner-lidar", per_run_of=LIDAR_SCANS)

    # ---- HBM traffic of the headline kernel, measured now (sub-processes under rocprofv3)
    if not args.no_pmc:
        traffic, note = pmc_traffic(args, "mrh::k_back<true, false", cache)
        roof["traffic"], roof["traffic_note"] = traffic, note
    else:
        roof["traffic_note"] = "--no-pmc"

    # ---- CPU baseline: the oracle on a bounded sample of the same stream
    cpu = None
    if not args.no_cpu and args.cpu_frames > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes

        import parity_utils as pu

        orc = pu.oracle_lib()
        orc.orc_num_threads.restype = ctypes.c_int
        cores = int(orc.orc_num_threads())
        ce = make_engine(orc, capi.Params(num_sdf_blocks=131072, **synth.REPLICA_PARAMS), Kc)
        n_cpu = min(args.cpu_frames, total)
        tc = 0.0
        for i in range(n_cpu):
            f = frames[i]
            ce.set_pose(f.R, f.t)
            ce.upload_depth(f.depth)
            ce.upload_rgb(f.rgb)
            c0 = time.perf_counter()
            ce.integrate()  # the reference brackets exactly this call (voxel_data_structures.cpp:94-109)
            tc += time.perf_counter() - c0
        # the oracle's map after these frames against the HIP engine's: the TIMED engine's own map when the sample covers the whole
        # run (the driver's --steps 20 --warmup 5 does), otherwise a replay of the sample through the same entry points
        if timed_map is not None:
            da, va, which = timed_map[0], timed_map[1], "the timed engine (map taken right after the timed loop)"
        else:
            rp = make_engine(hip, params, Kc)
            res.run(rp, 0, n_cpu)
            rp.sync()
            da, va = rp.dump_blocks()
            rp.close()
            which = f"a replay of the first {n_cpu} frames through the timed loop's entry points (mrh_set_depth_device / mrh_set_rgb_device)"
        db, vb = ce.dump_blocks()
        same_occ = bool(len(da) == len(db) and np.array_equal(da, db))
        same_u8 = bool(same_occ and np.array_equal(va["weight"], vb["weight"]) and np.array_equal(va["rgb"], vb["rgb"]))
        same_sdf = bool(same_occ and np.array_equal(va["sdf"].view(np.uint32), vb["sdf"].view(np.uint32)))
        same_ssq = bool(same_occ and np.array_equal(va["sum_squared"].view(np.uint32), vb["sum_squared"].view(np.uint32)))
        max_dsdf = float(np.nanmax(np.abs(va["sdf"] - vb["sdf"]), initial=0.0)) if same_occ else None
        parity = {"checked": which, "frames": n_cpu, "blocks": int(len(da)), "oracle_blocks": int(len(db)),
                  "weighted_voxels": int((va["weight"] > 0).sum()), "occupancy_equal": same_occ, "weights_and_colours_equal": same_u8,
                  "sdf_bit_exact": same_sdf, "sum_squared_bit_exact": same_ssq, "max_abs_sdf_diff": max_dsdf,
                  "ok": bool(same_occ and same_u8 and max_dsdf is not None and max_dsdf <= 1e-5)}
        del da, va, db, vb
        ce.close()
        cpu = {"value": n_cpu / tc, "unit": "frames/s", "cores": cores, "kind": "port", "parity": parity,
               "sample": f"first {n_cpu} frames of the same 640x480 stream through the test oracle (oracle/mrh_oracle.c, gcc -O2 -fopenmp, "
                         f"allocation single-threaded: a restatement for checking results, not a tuned CPU implementation), time of mrh_integrate only"}

    out = {
        "metric": "depth frames/sec integrated (640x480)",
        "value": K / elapsed, "unit": "frames/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "replica-room0 stand-in 640x480, single-resolution hash TSDF integrate (alloc+compact+integrate+GC per frame, "
                               "replica.cfg params), frames resident in HBM",
                   "frames_per_gpu": K, "voxel_size_m": 0.01, "truncation_m": 0.07, "parallelism": "single GPU", "live_blocks_end": occupied,
                   "hash_table": table},
        "roofline": roof, "roofline_hbm": roof_hbm, "mc": mc, "cpu_baseline": cpu,
        "parity_checked": bool(cpu and cpu["parity"]["ok"]), "blocks": (cpu["parity"]["blocks"] if cpu else None),
        "lidar": lidar, "splat": splat, "pcie_inclusive_frames_per_s": pcie_fps, "pcie_inclusive_frames_timed": (max(1, -(-100 // K)) * K) if pcie_fps else None,
        "h2d_link_gbs": link["gbs"] if link else None, "h2d_link": link,
        "pcie_inclusive_frac_of_link": (pcie_fps * link["bytes_per_frame"] / 1e9 / link["gbs"]) if link and pcie_fps else None,
        "h2d_link_gbs_under_load": link.get("gbs_next_to_resident_frames") if link else None,
        "pcie_inclusive_frac_of_link_under_load": (pcie_fps * link["bytes_per_frame"] / 1e9 / link["gbs_next_to_resident_frames"]) if link and pcie_fps and link.get("gbs_next_to_resident_frames") else None,
        "periodic_frames": periodic, "spherical_images": spherical,
    }
    emit(out)


# ---- N > 1 ---------------------------------------------------------------------------------------------------------

class HostGroup:
    """Barrier / max / all-gather of a few numbers through files under MRH_RDZV_DIR (one node): what the TIMING of the frame-sharded
    run needs, and nothing more.  Used only when the RCCL communicator cannot be created (the run then reports `value` with
    `"backend": "host"` and the reason, and leaves the exchange phases out) or when MRH_BENCH_BACKEND=host asks for it (tests)."""

    def __init__(self, rank, world):
        key = os.environ.get("MRH_RDZV_KEY") or f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
        from mrhash_amd import parallel

        self.base = os.path.join(parallel.rdzv_dir(), f"mrh_hostgrp_{key}")  # per-user directory, private files
        self.rank, self.world, self.seq, self.mine = rank, world, 0, []
        self.not_before = parallel.launcher_start_time() - 2.0  # files older than the launcher are a previous run's
        self._publish = parallel.publish_file

    def _exchange(self, payload: str, timeout_s: float = 600.0):
        self.seq += 1
        path = f"{self.base}.{self.seq}.{self.rank}"
        self._publish(path, payload.encode())
        self.mine.append(path)
        out, t0 = [], time.time()
        for r in range(self.world):
            p = f"{self.base}.{self.seq}.{r}"
            while True:
                try:
                    if os.stat(p).st_mtime < self.not_before:
                        raise FileNotFoundError(p)  # left over from an earlier run with the same key
                    with open(p) as f:
                        out.append(f.read())
                    break
                except FileNotFoundError:
                    if time.time() - t0 > timeout_s:
                        raise TimeoutError(f"host group: rank {r} never reached step {self.seq}")
                    time.sleep(2e-5)
        return out

    def barrier(self):
        self._exchange("")

    def allgather_f64(self, values):
        return np.array([json.loads(v) for v in self._exchange(json.dumps([float(x) for x in values]))], dtype=np.float64)

    def close(self):
        self._exchange("")  # everybody has read every earlier step; this last step's (empty) files stay behind
        for p in self.mine[:-1]:
            try:
                os.unlink(p)
            except OSError:
                pass


class Group:
    """The ranks of the run: RCCL behind the C ABI (capi.Comm; the product path, no torch in the process) or — test mode,
    N ranks sharing one device — a gloo group; `host` (files, timing only) if the communicator cannot be created."""

    def __init__(self, hip, rank, world, device_index, backend):
        from mrhash_amd import capi, parallel

        self.rank, self.world, self.backend, self.note = rank, world, backend, None
        self.dist = self.handle = self.host = None
        if backend == "rccl":
            # every rank learns whether EVERY rank got its communicator: a rank that failed alone would otherwise leave the
            # others inside their first collective
            host = HostGroup(rank, world)
            err = ""
            try:
                self.handle = parallel.rendezvous(hip, rank, world, device_index, timeout_s=120.0)
            except Exception as e:  # noqa: BLE001  (library missing, bootstrap refused, ...)
                err = f"{type(e).__name__}: {e}"
            errs = [e for e in host._exchange(err) if e]
            if errs:
                if self.handle is not None:
                    self.handle.close()
                    self.handle = None
                self.backend, self.host, self.note = "host", host, "RCCL communicator not created: " + errs[0][:300]
                if rank == 0:
                    print(f"[bench.py] {self.note}; timing through the host group, exchange phases skipped", file=sys.stderr)
            else:
                host.close()
        elif backend == "host":
            self.host = HostGroup(rank, world)
        else:
            self.dist = parallel.init_process_group("gloo")
            self.handle = self.dist
            if self.dist.get_world_size() != world:
                raise SystemExit(f"bench.py: the process group reports {self.dist.get_world_size()} ranks, --gpus says {world}")
        self._capi = capi

    @property
    def exchanges(self) -> bool:
        return self.host is None

    def barrier(self):
        from mrhash_amd import hipmem

        hipmem.synchronize()
        if self.host is not None:
            self.host.barrier()
        elif self.dist is None:
            self.handle.barrier()
        else:
            self.dist.barrier()

    def max(self, x: float) -> float:
        if self.host is not None:
            return float(self.host.allgather_f64([x]).max())
        if self.dist is None:
            return float(self.handle.allreduce([x], self._capi.COMM_MAX)[0])
        import torch

        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allgather(self, values) -> np.ndarray:
        if self.host is not None:
            return self.host.allgather_f64(values).astype(np.int64)
        if self.dist is None:
            return self.handle.allgather_i64(values)
        import torch

        mine = torch.tensor(list(values), dtype=torch.int64)
        out = torch.empty(self.world * len(mine), dtype=torch.int64)
        self.dist.all_gather_into_tensor(out, mine)
        return out.numpy().reshape(self.world, len(mine))

    def close(self):
        if self.host is not None:
            self.host.close()
        elif self.dist is None:
            self.handle.close()
        else:
            self.dist.destroy_process_group()


def bench_multi(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    K, W = args.steps, args.warmup
    total = W + K
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: launched with WORLD_SIZE={world}; refusing to report n_gpus = {args.gpus}")
    # MRH_BENCH_SHARE_DEVICE=1 exists so that the multi-rank code path can be exercised on a 1-GPU box: the ranks share device
    # 0 and talk over gloo (RCCL refuses two ranks on one device).  The driver's runs use one GPU per rank over RCCL.
    share = os.environ.get("MRH_BENCH_SHARE_DEVICE") == "1"
    device_index = 0 if share else local_rank
    backend = os.environ.get("MRH_BENCH_BACKEND") or ("gloo" if share else "rccl")
    if backend == "gloo":
        import torch  # noqa: F401  (gloo; imported before the library binds its HIP runtime, see mrhash_amd/_runtime.py)

    from mrhash_amd import capi, hipmem, parallel, synth

    hip = capi.load_hip()
    ndev = hipmem.device_count()
    if not share and ndev < world:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} HIP device(s) visible; refusing to report n_gpus = {args.gpus}")
    hipmem.set_device(device_index)
    os.environ.setdefault("MRH_COMM_INIT_TIMEOUT_S", "150")  # ncclCommInitRank's watchdog (mrh_comm_create): past it the run goes on over the host group
    grp = Group(hip, rank, world, device_index, backend)
    backend = grp.backend
    devices = sorted(set(int(v) for v in grp.allgather([device_index])[:, 0]))
    n_gpus = 1 if share else len(devices)
    rccl = grp.exchanges and grp.dist is None
    # what RCCL itself says about the group (mrh_comm_status): the driver reads here that RCCL saw N ranks on N devices
    if rccl:
        st = grp.handle.status()
        seen = grp.allgather([st["rccl_ranks"], st["rccl_rank"], st["rccl_device"], st["async_error"]])
        rccl_info = {"rccl_ranks": st["rccl_ranks"], "ranks_seen_by_each_rank": [int(v) for v in seen[:, 0]], "rccl_rank_of_each_rank": [int(v) for v in seen[:, 1]],
                     "rccl_device_of_each_rank": [int(v) for v in seen[:, 2]], "devices": devices,
                     "async_error_after_init": st["async_error_string"], "async_error_code_of_each_rank": [int(v) for v in seen[:, 3]],
                     "async_error_after_phases": None, "rccl_version": st["rccl_version"], "library": st["library_path"]}
    else:
        rccl_info = {"rccl_ranks": None, "devices": devices, "async_error_after_init": None, "async_error_after_phases": None,
                     "reason": (grp.note or ("MRH_BENCH_SHARE_DEVICE=1: the ranks share one device, which RCCL refuses; they talk over gloo" if share
                                             else f"backend {backend} requested (MRH_BENCH_BACKEND)"))}

    # `value` is the weak-scaled N = 1 line: the SAME workload (configs[1]'s stream and parameters), rank r fusing frames
    # [r * total, (r + 1) * total) of the orbit (8 ranks x 25 frames x 1.8 degrees = once round the room), so that the driver's
    # value(N) / (N x value(1)) compares like with like — with one rank this function runs the job of bench_single
    # (tests/test_bench_gpu.py::test_one_rank_multi_line_agrees_with_the_single_gpu_line).  configs[3] as BASELINE.json words it
    # (the ScanNet stream cut into N segments, one merge + one boundary-block exchange) is `full_stream` below.
    Kc = synth.REPLICA_640
    params = capi.Params(num_sdf_blocks=args.blocks, device_id=device_index, **synth.REPLICA_PARAMS)
    chunk_log2 = 3

    # ---- frame-sharded fusion (value): this rank's own segment of the orbit
    mine = Resident(render_stream("replica", total, start=rank * total), Kc)
    eng = make_engine(hip, params, Kc)
    if rccl:
        eng.attach_comm(grp.handle)
    mine.run(eng, 0, W)
    eng.sync()
    grp.barrier()
    t0 = time.perf_counter()
    mine.run(eng, W, total)
    eng.sync()
    hipmem.synchronize()
    own = time.perf_counter() - t0  # this rank's K frames, device idle on both sides; the closing barrier is not one of the steps
    grp.barrier()
    elapsed = grp.max(own)
    sub_blocks = int(eng.stats().occupied_fine)
    sub_all = grp.allgather([sub_blocks])[:, 0]

    out = {
        "metric": "depth frames/sec integrated (640x480)",
        "value": world * K / elapsed, "unit": "frames/s", "n_gpus": n_gpus, "ranks": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "replica-room0 stand-in 640x480 (the N = 1 line's stream, weak-scaled: rank r fuses its own segment of the orbit), "
                               "single-resolution hash TSDF integrate (alloc+compact+integrate+GC per frame, replica.cfg params), frames resident in HBM; "
                               "configs[3] (scannet-scene0000 stand-in, 500 poses in N segments) is full_stream",
                   "frames_per_gpu": K, "voxel_size_m": 0.01, "truncation_m": 0.07,
                   "parallelism": f"FRAME-SHARDED: {world} ranks x {K} own frames into {world} sub-maps (backend {backend}, devices {devices})",
                   "sub_map_blocks_per_rank": [int(v) for v in sub_all]},
        # until the exchange phases below have run, `value` is the fusion alone — and says so; with them it becomes frames / (fusion +
        # sub-map merge + boundary-block exchange), one merge per run of K frames per rank
        "value_definition": "fuse only: frames / max-over-ranks time of the frame-sharded fusion (the exchange phases have not run or are not available)",
        "fuse_only_frames_per_s": world * K / elapsed, "fuse_only_ms_per_step": elapsed / K * 1e3,
        "backend": backend, "backend_note": grp.note, "rccl": rccl_info,
        "phases": None, "full_stream": None, "merge": None, "tile_sharded": None, "roofline": None, "cpu_baseline": None,
    }

    # Everything below is reported BESIDE the value.  A collective that never returns (one rank lost, a fabric problem) must
    # not take the measured value with it: after MRH_BENCH_PHASE_TIMEOUT seconds rank 0 prints the line with what it has and
    # every rank leaves.
    import threading

    def give_up():
        if rank == 0:
            out["phases_error"] = f"the exchange phases did not finish within {limit:.0f} s; value is the completed frame-sharded measurement"
            emit(out)
        os._exit(0)

    limit = float(os.environ.get("MRH_BENCH_PHASE_TIMEOUT", "420"))
    dog = threading.Timer(limit, give_up)
    dog.daemon = True
    dog.start()
    try:
        bench_multi_phases(args, grp, eng, mine, hip, params, Kc, chunk_log2, device_index, out, rccl, sub_blocks)
    except Exception as e:  # noqa: BLE001
        print(f"[bench.py] rank {rank}: exchange phases failed: {type(e).__name__}: {e}", file=sys.stderr)
        dog.cancel()
        if rank == 0:
            out["phases_error"] = f"{type(e).__name__}: {e}"[:400]
            emit(out)
        os._exit(0)  # the other ranks may be inside a collective this rank will never join: their own timers end them
    dog.cancel()
    if rccl:
        try:
            out["rccl"]["async_error_after_phases"] = grp.handle.status()["async_error_string"]
        except Exception as e:  # noqa: BLE001
            out["rccl"]["async_error_after_phases"] = f"{type(e).__name__}: {e}"[:200]
    if rank == 0:
        emit(out)
    grp.barrier()
    grp.close()
    if grp.note and grp.note.startswith("RCCL communicator not created"):
        # mrh_comm_create may have timed out: its init thread is then still inside ncclCommInitRank (include/mrhash_comm.h) and
        # RCCL's / HIP's static teardown must not run under it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def bench_multi_phases(args, grp, eng, mine, hip, params, Kc, chunk_log2, device_index, out, rccl, sub_blocks):
    from mrhash_amd import capi, parallel, synth

    rank, world = grp.rank, grp.world
    K, W = args.steps, args.warmup
    total = W + K
    merge = None
    merge_phases = halo_phases = tile_phases = None
    if grp.exchanges:
        # ---- sub-maps -> one tile-sharded map -> halo exchange (what a mesh extraction needs next)
        grp.barrier()
        t1 = time.perf_counter()
        info = parallel.merge_submaps(eng, grp.handle, chunk_log2)
        grp.barrier()
        merge_s = grp.max(time.perf_counter() - t1)
        merge_phases = eng.comm_phase_times() if rccl else None
        t2 = time.perf_counter()
        n_halo = parallel.exchange_halo(eng, grp.handle)
        grp.barrier()
        halo_s = grp.max(time.perf_counter() - t2)
        halo_phases = eng.comm_phase_times() if rccl else None
        owned = int(eng.stats().occupied_fine) - n_halo
        parallel.drop_halo(eng)
        allc = grp.allgather([sub_blocks, info["sent"], owned, n_halo])
        rec = capi.RECORD_BYTES
        merge = {"what": "sub-maps -> one tile-sharded map: all-to-all of blocks to their tile owner + weighted merge on the device "
                         "(mrh_comm_merge_submaps), then exchange of boundary blocks (mrh_comm_exchange_halo)",
                 "merge_ms": merge_s * 1e3, "halo_exchange_ms": halo_s * 1e3,
                 "blocks_sent_per_rank": [int(v) for v in allc[:, 1]], "bytes_sent_per_rank": [int(v) * rec for v in allc[:, 1]],
                 "owned_blocks_after_merge_per_rank": [int(v) for v in allc[:, 2]], "halo_blocks_taken_per_rank": [int(v) for v in allc[:, 3]],
                 "cadence": f"one merge + one halo exchange per run ({K} frames per rank)"}
        # configs[3] as stated — "frame-sharded ... with RCCL boundary-voxel all-gather": the collectives are part of the job, so they
        # are part of `value`.  Each term is a max over ranks between barriers.
        fuse_s = out["fuse_only_ms_per_step"] * K / 1e3
        if world > 1 or os.environ.get("MRH_COMM_SELF_LOOP"):
            out["value"] = world * K / (fuse_s + merge_s + halo_s)
            out["ms_per_step"] = (fuse_s + merge_s + halo_s) / K * 1e3
            out["value_definition"] = (f"frames / (frame-sharded fusion + mrh_comm_merge_submaps + mrh_comm_exchange_halo): {world} x {K} frames, one merge and one "
                                       f"boundary-block exchange per run; fusion {fuse_s * 1e3:.2f} ms + merge {merge_s * 1e3:.2f} ms + halo {halo_s * 1e3:.2f} ms "
                                       f"(fuse_only_frames_per_s is the fusion alone; tile_sharded the result-identical mode)")
        else:
            # one rank: its sub-map is the map — there is no second sub-map to merge and no neighbour to exchange with, so the job
            # is the N = 1 line's job and `value` must be that line's value.  The two calls still ran (above) and are reported in
            # `merge` / `phases` as what they cost when they move nothing; MRH_COMM_SELF_LOOP=1 counts them as at N > 1.
            out["value_definition"] = (f"one rank: frames / time of the fusion ({K} frames; the job of the N = 1 line); mrh_comm_merge_submaps "
                                       f"({merge_s * 1e3:.2f} ms) and mrh_comm_exchange_halo ({halo_s * 1e3:.2f} ms) ran with nothing to move and are not part of it")
            merge["cadence"] += "; one rank: not part of `value`"
    if rccl:
        eng.attach_comm(None)
    eng.close()
    out["merge"] = merge

    # ---- roofline of the integrate kernel on rank 0's own segment (profiled pass on a second context; the other ranks wait)
    roof = None
    if rank == 0:
        pe = make_engine(hip, params, Kc)
        roof = profiled_roofline(pe, mine, W, total, f"rank 0's segment of the frame-sharded stream ({K} frames; the N = 1 line's workload)")
        roof["cache_note"] = "per-frame working set inside the 256 MiB Infinity Cache (see the N = 1 line's roofline_hbm for the kernel outside it)"
        pe.close()
    out["roofline"] = roof
    grp.barrier()

    # ---- tile-sharded fusion of ONE stream (rank 0's segment) by all ranks
    if grp.exchanges:
        shared = mine if rank == 0 else Resident(render_stream("replica", total, start=0), Kc)
        tp = capi.Params(num_sdf_blocks=args.blocks, device_id=device_index, shard_rank=rank, shard_count=world, shard_chunk_log2=chunk_log2,
                         **synth.REPLICA_PARAMS)
        te = make_engine(hip, tp, Kc)
        if rccl:
            te.attach_comm(grp.handle)  # starve frames: the two MIN all-reduces run inside mrh_integrate
            step = None
        else:
            step = lambda e: parallel.integrate(e, grp.handle)  # noqa: E731
        shared.run(te, 0, W, integrate=step)
        te.sync()
        grp.barrier()
        t3 = time.perf_counter()
        shared.run(te, W, total, integrate=step)
        te.sync()
        grp.barrier()
        tile_elapsed = grp.max(time.perf_counter() - t3)
        tile_blocks = int(te.stats().occupied_fine)
        tile_phases = te.comm_phase_times() if rccl else None
        if rccl:
            te.attach_comm(None)
        te.close()
        out["tile_sharded"] = {"what": f"the result-identical mode: every rank sees the same {K} frames and fuses only the tiles it owns (union of the "
                                       f"{world} tables == the single-GPU map); starve frames run their MIN all-reduce; strong scaling",
                               "frames_per_s": K / tile_elapsed, "ms_per_step": tile_elapsed / K * 1e3, "owned_blocks_rank0": tile_blocks,
                               "chunk_log2": chunk_log2}

    # ---- configs[3] at the cadence the configuration implies: the 500-pose stream cut into `world` contiguous segments, every rank
    # fuses its 500 / world frames, then ONE merge + ONE boundary-block exchange.  (`value` above amortises the same exchange over
    # --steps frames per rank: with the driver's 20 steps the all-to-all of whole sub-maps dominates it.)
    full_n = int(os.environ.get("MRH_BENCH_FULL_STREAM", "500"))
    if grp.exchanges and full_n >= world:
        per = full_n // world
        Ks = synth.SCANNET
        seg = Resident(render_stream("scannet", per, start=rank * per), Ks)
        fe = make_engine(hip, capi.Params(num_sdf_blocks=args.blocks, device_id=device_index, **synth.SCANNET_PARAMS), Ks)
        if rccl:
            fe.attach_comm(grp.handle)
        seg.run(fe, 0, min(W, per))  # warm-up on the segment's first frames, then from an empty map
        fe.sync()
        fe.reset()
        grp.barrier()
        t4 = time.perf_counter()
        seg.run(fe, 0, per)
        fe.sync()
        grp.barrier()
        f_fuse = grp.max(time.perf_counter() - t4)
        f_blocks = int(fe.stats().occupied_fine)
        grp.barrier()
        t5 = time.perf_counter()
        f_info = parallel.merge_submaps(fe, grp.handle, chunk_log2)
        grp.barrier()
        f_merge = grp.max(time.perf_counter() - t5)
        t6 = time.perf_counter()
        f_halo_n = parallel.exchange_halo(fe, grp.handle)
        grp.barrier()
        f_halo = grp.max(time.perf_counter() - t6)
        allf = grp.allgather([f_blocks, f_info["sent"], f_halo_n])
        parallel.drop_halo(fe)
        if rccl:
            fe.attach_comm(None)
        fe.close()
        out["full_stream"] = {"what": f"configs[3] whole: {per * world} poses of the walk in {world} contiguous segments of {per} frames, one per rank, frame-sharded; then one "
                                      f"mrh_comm_merge_submaps + one mrh_comm_exchange_halo",
                              "cadence_frames": per, "frames": per * world, "fuse_ms": f_fuse * 1e3, "merge_ms": f_merge * 1e3, "halo_exchange_ms": f_halo * 1e3,
                              "frames_per_s": per * world / (f_fuse + f_merge + f_halo), "fuse_only_frames_per_s": per * world / f_fuse,
                              "sub_map_blocks_per_rank": [int(v) for v in allf[:, 0]], "blocks_sent_per_rank": [int(v) for v in allf[:, 1]],
                              "halo_blocks_taken_per_rank": [int(v) for v in allf[:, 2]]}
        if merge is not None:
            merge["cadence_frames"] = K
            merge["cadence_frames_full_stream"] = per

    phase_keys = ("pack_ms", "counts_ms", "collective_ms", "unpack_ms", "bytes_out", "bytes_in")
    out["phases"] = {"what": "HIP-event times on rank 0: the two launches of a frame (profiled pass over rank 0's segment), the phases of the two "
                             "exchange calls, the starve all-reduces of the tile-sharded pass",
                     "k_front_ms": roof["k_front_ms_avg"] if roof else None, "k_back_ms": roof["kernel_ms_avg"] if roof else None,
                     "merge": {k: merge_phases[k] for k in phase_keys} if merge_phases else None,
                     "halo": {k: halo_phases[k] for k in phase_keys} if halo_phases else None,
                     "starve_allreduce_ms_avg": (tile_phases["allreduce_ms_sum"] / tile_phases["allreduce_count"]) if tile_phases and tile_phases["allreduce_count"] else None,
                     "starve_allreduce_count": tile_phases["allreduce_count"] if tile_phases else None}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched the single-process way: become N ranks, one per GPU (what torch.distributed.run does for the driver: RANK /
        # LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment of N copies of this script)
        from mrhash_amd import hipmem

        ndev = hipmem.device_count()
        share = os.environ.get("MRH_BENCH_SHARE_DEVICE") == "1"  # test mode: N ranks on device 0
        if ndev < args.gpus and not share:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} HIP device(s) visible; refusing to report n_gpus = {args.gpus}")
        port = free_port()
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       MRH_RDZV_KEY=f"bench_{os.getpid()}_{port}")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rcs = [p.wait() for p in procs]
        raise SystemExit(max(abs(rc) for rc in rcs))
    if not (args.pmc_inner or args.pmc_inner_big or args.pmc_inner_mc or args.pmc_inner_lidar or args.pmc_inner_sph):
        global _RESULT_FD
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)  # from here on fd 1 is stderr for every library in the process; emit() writes the result line
    # Python's cyclic collector stays off for the run: a generation-2 pass takes milliseconds and fires after a fixed number of
    # allocations, i.e. inside whichever timed loop happens to cross it (seen as a 20x outlier of one leg in
    # tools/bench_tile_shards.py).  Nothing here builds reference cycles that matter.
    gc.collect()
    gc.disable()
    if args.gpus > 1 or args.multi:
        bench_multi(args)
    else:
        if int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit(f"bench.py --gpus 1 launched with WORLD_SIZE={os.environ['WORLD_SIZE']}")
        from mrhash_amd import hipmem

        if hipmem.device_count() < 1:
            raise SystemExit("bench.py needs a HIP device (hipGetDeviceCount reports none)")
        hipmem.set_device(0)
        bench_single(args)


if __name__ == "__main__":
    main()
