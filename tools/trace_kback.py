#!/usr/bin/env python3
"""Tuning aid: per-block phase timestamps of the last k_back launch (build with -DMRH_TRACE, see tools/trace_kback.sh)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mrhash_amd import capi, synth
capi.HIP_LIB_PATH = os.path.join(ROOT, "mrhash_amd", "csrc", "libmrhash_trace.so")
out = os.path.join(ROOT, "gpurun_out", "trace.bin")
os.environ["MRH_TRACE_FILE"] = out
hip = capi.load_hip()
Kc = synth.REPLICA_640
params = capi.Params(num_sdf_blocks=262144, **synth.REPLICA_PARAMS)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scene = synth.replica_room()
frames = [synth.render(scene, Kc, t, q, depth_scaling=6553.5) for t, q in synth.orbit_poses(n)]
dd = torch.from_numpy(np.stack([f.depth for f in frames])).cuda()
rr = torch.from_numpy(np.stack([f.rgb for f in frames])).cuda()
e = capi.Engine(hip, params)
e.set_camera(Kc.fx, Kc.fy, Kc.cx, Kc.cy, Kc.rows, Kc.cols, params.min_depth, params.max_depth)
for i, f in enumerate(frames):
    e.set_pose(f.R, f.t)
    e.set_depth_device(dd.data_ptr() + i * Kc.rows * Kc.cols * 4, Kc.rows, Kc.cols)
    e.set_rgb_device(rr.data_ptr() + i * Kc.rows * Kc.cols * 3, Kc.rows, Kc.cols)
    e.integrate()
e.sync()
nvis = int(e.stats().compact_blocks) if hasattr(e.stats(), "compact_blocks") else None
e.close()
raw = np.fromfile(out, dtype=np.uint64).reshape(-1, 8)
# ---- k_front records (one per workgroup) -------------------------------------------------------------------
fr = raw[32768:]
fr = fr[fr[:, 0] > 0]
if len(fr):
    t0 = int(fr[:, 0].min())
    issweep = (fr[:, 7] >> np.uint64(63)) != 0
    F = (fr[:, :5].astype(np.int64) - t0) / 100.0
    print("k_front: workgroups", len(fr), "sweep", int(issweep.sum()), "span us", F[:, 4].max())
    tl = F[~issweep]
    for i, nm in enumerate(["start", "lds init done", "rays done (thread 0)", "all rays done", "probe/insert done"]):
        print(f"  tile  {nm:22s} p10 {np.percentile(tl[:, i], 10):6.2f} p50 {np.percentile(tl[:, i], 50):6.2f} p90 {np.percentile(tl[:, i], 90):6.2f} max {tl[:, i].max():6.2f}")
    print("  tile  distinct keys per tile: mean", (fr[~issweep, 7]).astype(np.int64).mean(), "max", (fr[~issweep, 7]).astype(np.int64).max())
    # placement: workgroups per CU (XCC id, SE id, CU id from HW_ID) vs finish time
    hwid = fr[~issweep, 5]
    xcc = (hwid >> np.uint64(32)) & np.uint64(0xF)
    cu = (hwid >> np.uint64(8)) & np.uint64(0xF); sh_ = (hwid >> np.uint64(12)) & np.uint64(1); se = (hwid >> np.uint64(13)) & np.uint64(0x7)
    cuid = (xcc.astype(np.int64) << 12) | (se.astype(np.int64) << 8) | (sh_.astype(np.int64) << 4) | cu.astype(np.int64)
    uniq, inv, cnt = np.unique(cuid, return_inverse=True, return_counts=True)
    print("  tile  distinct CUs", len(uniq), "tile workgroups per CU: min", cnt.min(), "max", cnt.max(), "hist", np.bincount(cnt))
    for k in np.unique(cnt):
        sel = cnt[inv] == k
        print(f"        CUs with {k} tile wgs: rays done mean {tl[sel, 3].mean():6.2f}  end mean {tl[sel, 4].mean():6.2f} max {tl[sel, 4].max():6.2f}")
    sw = fr[issweep]
    if len(sw):
        S0 = (sw[:, 0].astype(np.int64) - t0) / 100.0; S4 = (sw[:, 4].astype(np.int64) - t0) / 100.0
        print(f"  sweep start p50 {np.percentile(S0, 50):6.2f} end p50 {np.percentile(S4, 50):6.2f} end max {S4.max():6.2f}; descriptors per wg {int((sw[0, 7] & np.uint64(0xFFFFFFFF)))}")
tr = raw[:32768]
ok = tr[:, 0] > 0
tr = tr[ok]
# entries from older launches can linger beyond nvis: keep the cluster of the last launch (t0 within 1 ms of the max)
t0max = tr[:, 0].max()
tr = tr[(t0max - tr[:, 0]) < 20000]
base = tr[:, 0].min()
T = (tr[:, :7].astype(np.int64) - int(base)) / 100.0  # s_memrealtime: 100 MHz -> us
hw = tr[:, 7]
print("blocks traced", len(tr), "span us", T[:, 6].max())
names = ["start", "issued", "fill_issued", "proj_done+tile_ready", "lookup_done", "blend+stores issued", "end"]
for i in range(7):
    print(f"{names[i]:28s} p10 {np.percentile(T[:, i], 10):7.2f} p50 {np.percentile(T[:, i], 50):7.2f} p90 {np.percentile(T[:, i], 90):7.2f} max {T[:, i].max():7.2f}")
D = np.diff(T, axis=1)
for i in range(6):
    print(f"phase {i}->{i+1}: mean {D[:, i].mean():6.2f} p50 {np.percentile(D[:, i], 50):6.2f} p90 {np.percentile(D[:, i], 90):6.2f}")
print("per-block total mean", (T[:, 6] - T[:, 0]).mean())
# start-time histogram (generations)
h, edges = np.histogram(T[:, 0], bins=30)
print("start histogram:", list(zip(np.round(edges[:-1], 1), h)))
h, edges = np.histogram(T[:, 6], bins=30)
print("end histogram:", list(zip(np.round(edges[:-1], 1), h)))
upd = (hw >> np.uint64(32)) & np.uint64(0xFFFF); px = (hw >> np.uint64(48)) & np.uint64(0x7FFF); skipped = (hw >> np.uint64(63)) != 0
print('skipped blocks', int(skipped.sum()), 'of', len(hw), '; skipped dur mean', (T[skipped, 6] - T[skipped, 0]).mean() if skipped.any() else 0)
print("updated voxels/block histogram:", np.histogram(upd.astype(np.int64), bins=[0,1,64,128,256,384,511,512,513])[0], "mean", upd.mean(), "tile px mean", px.mean(), "p90", np.percentile(px.astype(np.int64),90), "no tile", int((px==0).sum()))
dur = T[:, 6] - T[:, 0]
for lo, hi in [(0,1),(1,256),(256,512),(512,513)]:
    sel = (upd >= lo) & (upd < hi)
    if sel.any(): print("upd in [%d,%d): n %d dur mean %.2f" % (lo, hi, sel.sum(), dur[sel].mean()))
cu = (hw >> np.uint64(8)) & 0xF; 

