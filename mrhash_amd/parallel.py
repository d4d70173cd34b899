"""Multi-GPU support: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  The reference is single-GPU (no NCCL/MPI/streams anywhere), so this
is new design.  Two ways to use N GPUs, and how they connect:

* tile sharding (`Params.shard_rank/shard_count/shard_chunk_log2`, `Engine.set_sharding`) — the result-identical mode.
  Every rank sees every frame but only inserts the blocks whose chunk it owns, so the union of the tables is
  bit-identical to the single-GPU map.  Exchange steps: an element-wise MIN all-reduce of the per-pixel z-buffer on
  starve frames (`integrate`), and before marching cubes an all-gather of boundary blocks (`exchange_halo`): corner
  samples reach into neighbouring blocks, so each rank needs the blocks of other ranks that touch its chunks.  Every
  rank then extracts triangles for the blocks it owns and rank 0 merges the buffers into the single-GPU canonical
  order (`gather_mesh`).
* frame sharding (bench.py --gpus N, BASELINE.json configs[3]) — every rank fuses its own segment of the stream into
  its own sub-map owning everything; no collective in the per-frame loop.  `merge_submaps` then turns the N sub-maps
  into ONE tile-sharded map: every rank sends each block to the rank that owns its tile (all-to-all), the owner folds
  the sub-maps with combineVoxel's weighted mean in rank order, and from there on the map is a tile-sharded one
  (halo exchange, marching cubes, mesh gather as above).

Blocks travel as `mrh_block_record`s (16-byte descriptor + 512 reference-layout voxels) in DEVICE memory: the
library packs them into a device buffer (`mrh_pack_blocks`), RCCL moves device buffers, the library consumes device
buffers (`mrh_unpack_blocks`).  Nothing is staged through the host on the nccl path; the gloo path (CPU tests, and
the single-GPU test box where two ranks share one device) stages through host tensors because gloo needs them.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Tuple

import numpy as np

from . import capi

P0, P1, P2 = 73856093, 19349669, 83492791
REC = capi.RECORD_BYTES


def owner_of_blocks(xyz: np.ndarray, world: int, chunk_log2: int = 3) -> np.ndarray:
    """Rank that owns each block position; mirrors `owns_block` in mrhash_amd/csrc/mrh_device.h."""
    if world <= 1:
        return np.zeros(len(xyz), dtype=np.int64)
    c = (np.asarray(xyz, dtype=np.int64) >> chunk_log2).astype(np.uint32)  # arithmetic shift, then wrap to u32
    h = (c[:, 0] * np.uint32(P0)) ^ (c[:, 1] * np.uint32(P1)) ^ (c[:, 2] * np.uint32(P2))
    h = h ^ (h >> np.uint32(15))
    return (h % np.uint32(world)).astype(np.int64)


def shard_frames(n_frames: int, rank: int, world: int) -> range:
    """Contiguous segment of a stream for frame sharding (weak scaling: n_frames per rank)."""
    return range(rank * n_frames, (rank + 1) * n_frames)


def init_process_group(backend: Optional[str] = None):
    """RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run); 127.0.0.1 rendezvous."""
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend=backend, **kw)
    return dist


class _DeviceArray:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, n: int, typestr: str = "<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _bytes_view(ptr: int, nbytes: int, on_device: bool):
    """uint8 torch tensor over `nbytes` at `ptr` (device or host memory), no copy."""
    import torch

    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device="cuda" if on_device else "cpu")
    if on_device:
        return torch.as_tensor(_DeviceArray(ptr, nbytes, "|u1"), device="cuda")
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr)))


def _force_collectives() -> bool:
    """MRH_FORCE_COLLECTIVES=1: run every collective even in a one-rank group.  A 1-GPU box cannot host two RCCL ranks
    (RCCL refuses duplicate GPUs), so this is how the nccl branches — device buffers of the library handed to RCCL and
    back — get executed there (tests/test_sharding_gpu.py)."""
    return os.environ.get("MRH_FORCE_COLLECTIVES") == "1"


def _comm_device(dist):
    import torch

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _all_gather_counts(dist, values: List[int], dev) -> np.ndarray:
    """[world, len(values)] int64: every rank's `values`."""
    import torch

    world = dist.get_world_size()
    mine = torch.tensor(values, dtype=torch.int64, device=dev)
    out = torch.empty(world * len(values), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().numpy().reshape(world, len(values))


def _unpack(engine: capi.Engine, mode: int, buf, first_byte: int, n_records: int) -> int:
    """Hands records [first_byte, first_byte + n * REC) of a torch uint8 tensor to the library where they lie."""
    if n_records == 0:
        return 0
    return engine.unpack_blocks(mode, buf.data_ptr() + first_byte, n_records, buf.is_cuda)


def integrate(engine: capi.Engine, dist=None, n_frames_invalidate: int = -1):
    """One frame on a tile-sharded context.  On starve frames the library stops twice for an element-wise MIN of the
    per-pixel z-buffer over all ranks (the only data-path collective of the fusion loop; every n-th frame, 2.4 MB
    at 640x480): RCCL all-reduce over xGMI on the GPU box, gloo in the CPU tests."""
    import torch

    pending = engine.integrate(n_frames_invalidate)
    while pending:
        ptr, n, on_device = engine.exchange_buffer()
        if dist is not None and (dist.get_world_size() > 1 or _force_collectives()):
            if on_device:
                t = torch.as_tensor(_DeviceArray(ptr, n), device="cuda")
                if dist.get_backend() == "nccl":
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    torch.cuda.synchronize()
                else:  # gloo rendezvous with GPU engines (single-GPU test box): stage through the host
                    h = t.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.MIN)
                    t.copy_(h)
                    torch.cuda.synchronize()
            else:
                arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int64)), shape=(n,))
                t = torch.from_numpy(arr)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
        pending = engine.integrate_resume()


def boundary_mask(descs: np.ndarray, chunk_log2: int) -> np.ndarray:
    """Blocks on the surface of their chunk: the only ones a neighbouring chunk's marching cubes can read
    (host mirror of the MRH_PACK_HALO predicate, used by tests)."""
    side = 1 << chunk_log2
    m = np.zeros(len(descs), dtype=bool)
    for ax in ("x", "y", "z"):
        l = descs[ax] & (side - 1)
        m |= (l == 0) | (l == side - 1)
    return m


def exchange_halo(engine: capi.Engine, dist) -> int:
    """All-gather of boundary blocks, device to device: the library selects the owned blocks on the surface of their
    chunk and packs their records into a device buffer (`mrh_pack_blocks(MRH_PACK_HALO)`), one
    `all_gather_into_tensor` moves them (padded to the largest rank's count), and each rank's segment is consumed where
    it landed (`mrh_unpack_blocks(MRH_UNPACK_HALO)`: only blocks 26-adjacent to a position this rank owns are kept).
    Terminal for fusion until `drop_halo`: the library refuses to integrate while halo blocks are present.
    Returns the number of halo blocks this rank took."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1 and not _force_collectives():
        return 0
    dev = _comm_device(dist)
    ptr, n, on_device = engine.pack_blocks(capi.PACK_HALO)
    counts = _all_gather_counts(dist, [n], dev)[:, 0]
    mx = int(counts.max())
    if mx == 0:
        return 0
    send = torch.zeros(mx * REC, dtype=torch.uint8, device=dev)
    if n:
        send[: n * REC].copy_(_bytes_view(ptr, n * REC, on_device))  # device-to-device on the nccl path
    recv = torch.empty(world * mx * REC, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    if recv.is_cuda:
        torch.cuda.synchronize()
    taken = 0
    for r in range(world):
        if r != rank:
            taken += _unpack(engine, capi.UNPACK_HALO, recv, r * mx * REC, int(counts[r]))
    return taken


def drop_halo(engine: capi.Engine) -> int:
    """Removes the blocks `exchange_halo` brought in (after the extraction that needed them); fusion may continue."""
    return engine.drop_blocks(capi.DROP_HALO)


def merge_submaps(engine: capi.Engine, dist, chunk_log2: int = 3) -> dict:
    """Frame-sharded sub-maps -> one tile-sharded map.  Every rank packs, per destination, the blocks whose tile that
    rank owns (`MRH_PACK_OWNER`), one all-to-all moves them (RCCL `all_to_all_single` with per-rank split sizes; gloo
    has no all-to-all on CPU tensors, so the test path all-gathers and selects), the local map is emptied and the N
    incoming sub-maps are folded in rank order with combineVoxel's weighted mean (`MRH_UNPACK_MERGE`,
    vhu.cuh:167-181).  TSDF values and weights of the result equal a single-GPU fusion of all frames up to the
    rounding of the running mean (while weights stay below the clamp); colours are order-dependent (50/50 blend).
    Returns {"sent": blocks sent, "received": blocks received, "bytes": payload bytes through the collective}."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    engine.set_sharding(rank, world, chunk_log2)
    if world == 1 and not _force_collectives():
        return {"sent": 0, "received": 0, "bytes": 0}
    dev = _comm_device(dist)
    parts, out_counts = [], []
    for dest in range(world):
        ptr, n, on_device = engine.pack_blocks(capi.PACK_OWNER, dest)
        out_counts.append(n)
        t = torch.empty(n * REC, dtype=torch.uint8, device=dev)
        if n:
            t.copy_(_bytes_view(ptr, n * REC, on_device))
            if t.is_cuda:  # the copy runs on torch's stream, the next pack reuses (or re-allocates) the buffer on the library's
                torch.cuda.synchronize()
        parts.append(t)
    counts = _all_gather_counts(dist, out_counts, dev)  # counts[src, dest]
    in_counts = [int(counts[src, rank]) for src in range(world)]
    send = torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=dev)
    engine.drop_blocks(capi.DROP_ALL)  # the owned blocks come back through the fold, at this rank's position in the order
    if dist.get_backend() == "nccl":
        recv = torch.empty(sum(in_counts) * REC, dtype=torch.uint8, device=dev)
        dist.all_to_all_single(recv, send, output_split_sizes=[c * REC for c in in_counts], input_split_sizes=[c * REC for c in out_counts])
        torch.cuda.synchronize()
        offsets = np.concatenate([[0], np.cumsum(in_counts)])[:-1] * REC
        segments = [(recv, int(offsets[src]), in_counts[src]) for src in range(world)]
    else:
        mx = int(counts.sum(axis=1).max())
        padded = torch.zeros(mx * REC, dtype=torch.uint8, device=dev)
        padded[: send.numel()].copy_(send)
        gathered = torch.empty(world * mx * REC, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, padded)
        segments = []
        for src in range(world):
            before = int(counts[src, :rank].sum())
            segments.append((gathered, (src * mx + before) * REC, in_counts[src]))
    for buf, first, n in segments:  # rank order: the fold is deterministic for a given world size
        _unpack(engine, capi.UNPACK_MERGE, buf, first, n)
    sent = int(sum(out_counts)) - out_counts[rank]
    return {"sent": sent, "received": int(sum(in_counts)) - in_counts[rank], "bytes": sent * REC}


def gather_mesh(engine: capi.Engine, dist) -> Optional[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]:
    """Every rank extracts the triangles of the blocks it owns; rank 0 brings the per-block runs of all ranks into the
    canonical single-GPU order (block position) and runs the mesh post-process.  The triangles stay where the library keeps
    them: the soup of the extraction is a device buffer (`mrh_get_triangles_device`), one `all_gather_into_tensor` moves it
    (padded to the largest rank's count), and `mrh_process_triangle_runs` permutes the runs on the device; only the
    per-block descriptors and counts (20 bytes a block) travel as host-side metadata.  Returns (triangles, V, F, C) on
    rank 0, None elsewhere."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _comm_device(dist)
    engine.extract_triangles(soup=False)  # the soup stays in the library's memory
    ptr, nt, on_device = engine.triangles_device()
    descs, counts = engine.triangle_blocks()
    keep = counts > 0
    nb = int(keep.sum())
    sizes = _all_gather_counts(dist, [nb, nt], dev)
    max_b, max_t = max(int(sizes[:, 0].max()), 1), max(int(sizes[:, 1].max()), 1)
    meta = np.zeros(max_b * 20, np.uint8)
    if nb:
        meta[: nb * 16] = np.frombuffer(descs[keep].tobytes(), np.uint8)
        meta[max_b * 16: max_b * 16 + nb * 4] = np.frombuffer(counts[keep].astype(np.uint32).tobytes(), np.uint8)
    meta_t = torch.from_numpy(meta).to(dev)
    all_meta = torch.empty(world * max_b * 20, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(all_meta, meta_t)
    send = torch.zeros(max_t * 72, dtype=torch.uint8, device=dev)
    if nt:
        send[: nt * 72].copy_(_bytes_view(ptr, nt * 72, on_device))  # device-to-device on the nccl path
    recv = torch.empty(world * max_t * 72, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    if rank != 0:
        return None
    if recv.is_cuda:
        torch.cuda.synchronize()
    host_meta = all_meta.cpu().numpy()
    all_d, all_c = [], []
    for r in range(world):
        b = int(sizes[r, 0])
        m = host_meta[r * max_b * 20: (r + 1) * max_b * 20]
        all_d.append(np.frombuffer(m[: 16 * b].tobytes(), dtype=capi.DESC_DTYPE))
        all_c.append(np.frombuffer(m[max_b * 16: max_b * 16 + 4 * b].tobytes(), dtype=np.uint32))
    # the ranks' soups, closed up (each segment of `recv` is padded to max_t triangles)
    total = int(sizes[:, 1].sum())
    packed = torch.empty(max(total, 1) * 72, dtype=torch.uint8, device=dev)
    off = 0
    for r in range(world):
        t = int(sizes[r, 1])
        if t:
            packed[off * 72: (off + t) * 72].copy_(recv[r * max_t * 72: (r * max_t + t) * 72])
        off += t
    if packed.is_cuda:
        torch.cuda.synchronize()
    engine.process_triangle_runs(np.concatenate(all_d), np.concatenate(all_c), packed.data_ptr(), total, packed.is_cuda)
    V, F, C = engine.extract_mesh()
    mptr, mn, mdev = engine.triangles_device()
    if mn:
        raw = _bytes_view(mptr, mn * 72, mdev)
        merged = np.frombuffer((raw.cpu() if mdev else raw).numpy().tobytes(), dtype=capi.TRI_DTYPE).reshape(mn, 3)
    else:
        merged = np.zeros((0, 3), dtype=capi.TRI_DTYPE)
    return merged, V, F, C
