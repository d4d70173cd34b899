#!/usr/bin/env python3
"""bench.py — depth frames/s integrated at 640x480 (BASELINE.json metric) on N MI355X, with the HBM-roofline
fraction of the integrate kernel and the CPU restatement timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): "Replica room0" stand-in — analytic 6x3x4 m box room, orbit trajectory,
Replica intrinsics rescaled to 640x480, depth quantised to 1/6553.5 m, replica.cfg parameters (1 cm voxels,
7 cm truncation, GC every frame, starve every 100th).  One step = one frame through the whole
VoxelContainer::integrate chain (allocate along rays -> frustum compaction -> depth->TSDF integrate -> GC).
Frames are resident in HBM before the timed region starts (the PCIe-inclusive figure is in DESIGN.md).

N > 1: frame-sharded, weak scaling — every rank fuses its own K-frame segment of the stream into its own
sub-map, no data-path collective inside the timed region; value = N*K / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01", "bench_pmc_summary.txt")


def pmc_traffic_bytes():
    """HBM bytes per launch of the integrate kernel from the committed rocprofv3 PMC passes of this same command
    (separate --pmc runs for FETCH_SIZE and WRITE_SIZE, tools/profile_bench.sh).  Units are KiB; on gfx950
    FETCH_SIZE counts a wide coalesced read at half its bytes, so it is doubled (MI355X_MICROARCH.md, HBM)."""
    try:
        lines = open(PMC_SUMMARY).read().splitlines()
        for i, l in enumerate(lines):
            if l.startswith("mrh::k_back<true, false"):
                kv = dict(tok.split("=") for tok in lines[i + 1].split())
                return (2.0 * float(kv["FETCH_SIZE"]) + float(kv["WRITE_SIZE"])) * 1024.0
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=262144, help="SDF block pool capacity (1.6 GB at 262144)")
    ap.add_argument("--cpu-frames", type=int, default=64, help="frames of the same stream timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pcie", action="store_true", help="also time the same frames fed from host memory (mrh_upload_* per frame)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    K, W = args.steps, args.warmup

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (torch.cuda.is_available() is False)")
    # MRH_BENCH_DEVICE / MRH_BENCH_BACKEND exist so that the multi-rank code path can be exercised on a 1-GPU box
    # (two ranks sharing device 0 over gloo); the driver's runs use one GPU per rank over RCCL.
    device_index = int(os.environ.get("MRH_BENCH_DEVICE", local_rank))
    backend = os.environ.get("MRH_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)

    from mrhash_amd import capi, synth

    hip = capi.load_hip()  # no fallback: raises if the HIP library is missing
    Kc = synth.REPLICA_640
    params = capi.Params(num_sdf_blocks=args.blocks, device_id=device_index, **synth.REPLICA_PARAMS)

    # ---- synthetic stream: this rank's segment, uploaded to HBM before timing ------------------------------
    total = W + K
    scene = synth.replica_room()
    poses = synth.orbit_poses(total * world)[rank * total:(rank + 1) * total]
    frames = [synth.render(scene, Kc, t, q, depth_scaling=6553.5) for t, q in poses]
    depth_d = torch.from_numpy(np.stack([f.depth for f in frames])).cuda()
    rgb_d = torch.from_numpy(np.stack([f.rgb for f in frames])).cuda()
    torch.cuda.synchronize()
    dstride = Kc.rows * Kc.cols * 4
    rstride = Kc.rows * Kc.cols * 3

    def run(engine, lo, hi):
        for i in range(lo, hi):
            f = frames[i]
            engine.set_pose(f.R, f.t)
            engine.set_depth_device(depth_d.data_ptr() + i * dstride, Kc.rows, Kc.cols)
            engine.set_rgb_device(rgb_d.data_ptr() + i * rstride, Kc.rows, Kc.cols)
            engine.integrate()

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- pass A: the timed region (profile hooks off) -----------------------------------------------------
    eng = capi.Engine(hip, params)
    eng.set_camera(Kc.fx, Kc.fy, Kc.cx, Kc.cy, Kc.rows, Kc.cols, params.min_depth, params.max_depth)
    run(eng, 0, W)
    eng.sync()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    run(eng, W, total)
    eng.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = eng.stats()
    occupied = int(st.occupied_fine)

    # ---- optional pass C: the boundary as the reference uses it (host buffers -> mrh_upload_depth / mrh_upload_rgb each frame).
    # Runs BEFORE the profiled pass: launches that carry start / stop events switch the queue to profiling mode, which
    # slows every later dispatch of the process.
    pcie_fps = None
    if args.pcie:
        pe = capi.Engine(hip, params)
        pe.set_camera(Kc.fx, Kc.fy, Kc.cx, Kc.cy, Kc.rows, Kc.cols, params.min_depth, params.max_depth)
        for i in range(W):
            f = frames[i]
            pe.set_pose(f.R, f.t); pe.upload_depth(f.depth); pe.upload_rgb(f.rgb); pe.integrate()
        pe.sync()
        t2 = time.perf_counter()
        for i in range(W, total):
            f = frames[i]
            pe.set_pose(f.R, f.t); pe.upload_depth(f.depth); pe.upload_rgb(f.rgb); pe.integrate()
        pe.sync()
        pcie_fps = K / (time.perf_counter() - t2)
        pe.close()

    # ---- pass B: same frames, HIP events around every integrate-kernel launch + device-side U/M counters ---
    eng.reset()
    run(eng, 0, W)
    eng.sync()
    eng.set_profile(True)
    s0 = eng.stats()
    t1 = time.perf_counter()
    run(eng, W, total)
    eng.sync()
    prof_elapsed = time.perf_counter() - t1
    s1 = eng.stats()
    n_k = int(s1.n_integrate_kernel - s0.n_integrate_kernel)
    k_ms = float(s1.sum_integrate_kernel_ms - s0.sum_integrate_kernel_ms) / max(n_k, 1)
    U = (int(s1.total_updated_voxels) - int(s0.total_updated_voxels)) / max(n_k, 1)
    M = (int(s1.total_compact_blocks) - int(s0.total_compact_blocks)) / max(n_k, 1)
    img_bytes = 7 * Kc.rows * Kc.cols
    alg_bytes = 24.0 * U + 24.0 * M + img_bytes  # 12 B read + 12 B write per updated voxel
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    eng.close()

    # ---- CPU baseline: the oracle on a bounded sample of the same stream (rank 0, N == 1) ------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_frames > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ctypes
        import parity_utils as pu

        orc = pu.oracle_lib()
        orc.orc_num_threads.restype = ctypes.c_int
        cores = int(orc.orc_num_threads())
        ce = capi.Engine(orc, capi.Params(num_sdf_blocks=131072, **synth.REPLICA_PARAMS))
        ce.set_camera(Kc.fx, Kc.fy, Kc.cx, Kc.cy, Kc.rows, Kc.cols, params.min_depth, params.max_depth)
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
        ce.close()
        cpu = {"value": n_cpu / tc, "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": f"first {n_cpu} frames of the same 640x480 stream, CPU restatement (oracle/mrh_oracle.c, "
                         f"gcc -O2 -fopenmp; allocation single-threaded), time of mrh_integrate only"}

    if rank == 0:
        out = {
            "metric": "depth frames/sec integrated (640x480)",
            "value": n_gpus * K / elapsed if world == n_gpus else world * K / elapsed,
            "unit": "frames/s",
            "n_gpus": n_gpus,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "replica-room0 stand-in 640x480, single-resolution hash TSDF integrate "
                                   "(alloc+compact+integrate+GC per frame, replica.cfg params)",
                       "frames_per_gpu": K, "voxel_size_m": 0.01, "truncation_m": 0.07,
                       "parallelism": "frame-sharded sub-maps, no data-path collective" if world > 1 else "single GPU",
                       "live_blocks_end": occupied},
            "roofline": {"bound": "hbm", "kernel": "k_back (depth->TSDF integrate + GC summary + GC decision)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(),
                         "traffic_note": "bytes per launch from profiles/r01/bench_pmc_summary.txt (rocprofv3 PMC passes of this "
                                         "command, 2 x FETCH_SIZE + WRITE_SIZE); the ~85 MB working set stays in the 256 MiB "
                                         "Infinity Cache between frames",
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_avg": k_ms, "launches": n_k,
                         "updated_voxels_per_launch": U, "compact_blocks_per_launch": M,
                         "profiled_pass_ms_per_step": prof_elapsed / K * 1e3},
            "cpu_baseline": cpu,
        }
        if pcie_fps is not None:
            out["pcie_inclusive_frames_per_s"] = pcie_fps  # never `value`: inputs cross PCIe inside the timed region
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
