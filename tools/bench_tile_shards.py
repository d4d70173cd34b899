"""Per-rank frame time of the tile-sharded mode, measured on ONE GPU: for every rank r of an N-shard map the same
640x480 stream is fused by a context that owns shard r only (what rank r of an N-GPU job runs between the starve-frame
collectives).  The slowest rank bounds the N-GPU frame rate.

    python tools/bench_tile_shards.py [--stream scannet|replica] [--shards 1 2 4 8] [--steps 100] [--chunk-log2 3]
"""
import argparse
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402  (workload helpers only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stream", default="scannet")
    ap.add_argument("--shards", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--chunk-log2", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=262144)
    ap.add_argument("--only-rank", type=int, default=-1, help="time this rank only (profiling runs)")
    ap.add_argument("--keep-gc", action="store_true", help="leave Python's cyclic collector on inside the timed loop")
    ap.add_argument("--lib", default="", help="another build of libmrhash_hip.so (A/B runs)")
    args = ap.parse_args()

    import torch

    from mrhash_amd import capi, parallel, synth

    torch.cuda.set_device(0)
    if args.lib:
        capi.HIP_LIB_PATH = os.path.abspath(args.lib)
    hip = capi.load_hip()
    K, P = (synth.SCANNET, synth.SCANNET_PARAMS) if args.stream == "scannet" else (synth.REPLICA_640, synth.REPLICA_PARAMS)
    total = args.warmup + args.steps
    res = bench.Resident(bench.render_stream(args.stream, total), K)
    out = {"stream": args.stream, "steps": args.steps, "chunk_log2": args.chunk_log2, "shards": {}}
    for n in args.shards:
        per_rank = []
        for r in (range(n) if args.only_rank < 0 else [args.only_rank]):
            params = capi.Params(num_sdf_blocks=args.blocks, device_id=0, shard_rank=r, shard_count=n, shard_chunk_log2=args.chunk_log2, **P)
            e = bench.make_engine(hip, params, K)
            step = lambda eng: parallel.integrate(eng, None)  # noqa: E731  (starve frames: the two pauses without their all-reduce)
            res.run(e, 0, args.warmup, integrate=step)
            e.sync()
            gc.collect()
            if not args.keep_gc:
                gc.disable()  # a generation-2 pass of an interpreter with torch loaded takes tens of ms
            t0 = time.perf_counter()
            res.run(e, args.warmup, total, integrate=step)
            e.sync()
            dt = time.perf_counter() - t0
            gc.enable()
            per_rank.append({"rank": r, "us_per_frame": dt / args.steps * 1e6, "owned_blocks": int(e.stats().occupied_fine)})
            e.close()
        worst = max(p["us_per_frame"] for p in per_rank)
        out["shards"][str(n)] = {"slowest_rank_us_per_frame": worst, "frames_per_s_bound": 1e6 / worst, "ranks": per_rank}
        print(f"N={n}: slowest rank {worst:.1f} us/frame -> {1e6 / worst:.0f} frames/s; per rank "
              + " ".join(f"{p['us_per_frame']:.1f}" for p in per_rank), file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
