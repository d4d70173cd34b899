"""Experiment: what does it cost k_front to read the colour image straight out of pinned HOST memory (no H2D copy of it)?
Depth stays resident in HBM; only the rgb pointer moves.  usage: python tools/exp_rgb_zero_copy.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from mrhash_amd import capi, synth

torch.cuda.set_device(0)
hip = capi.load_hip()
K, P = synth.REPLICA_640, synth.REPLICA_PARAMS
n, w = 110, 10
frames = bench.render_stream("replica", n)
depth = torch.from_numpy(np.stack([f.depth for f in frames])).cuda()
rgb_dev = torch.from_numpy(np.stack([f.rgb for f in frames])).cuda()
rgb_pin = torch.from_numpy(np.stack([f.rgb for f in frames])).pin_memory()
torch.cuda.synchronize()
ds, rs = K.rows * K.cols * 4, K.rows * K.cols * 3
for label, rgb in (("rgb in HBM", rgb_dev), ("rgb in pinned host memory", rgb_pin), ("rgb in HBM", rgb_dev), ("rgb in pinned host memory", rgb_pin)):
    e = bench.make_engine(hip, capi.Params(num_sdf_blocks=262144, device_id=0, **P), K)

    def run(lo, hi):
        for i in range(lo, hi):
            f = frames[i]
            e.set_pose(f.R, f.t)
            e.set_depth_device(depth.data_ptr() + i * ds, K.rows, K.cols)
            e.set_rgb_device(rgb.data_ptr() + i * rs, K.rows, K.cols)
            e.integrate()

    run(0, w); e.sync()
    t0 = time.perf_counter(); run(w, n); e.sync(); dt = time.perf_counter() - t0
    print(f"{label}: {dt / (n - w) * 1e6:.1f} us per frame, {int(e.stats().occupied_fine)} blocks")
    e.close()
