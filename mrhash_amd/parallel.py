"""Multi-GPU support: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  The reference is single-GPU (no NCCL/MPI/streams anywhere), so this
is new design:

* frame sharding (bench.py --gpus N): every rank fuses its own segment of the stream into its own sub-map;
  no collective in the per-frame loop.
* tile sharding (`Params.shard_rank/shard_count/shard_chunk_log2`): every rank sees every frame but only inserts
  the blocks whose chunk it owns, so the union of the tables is bit-identical to the single-GPU map.  The one
  exchange step is before marching cubes: corner samples reach into neighbouring blocks, so each rank needs the
  blocks of other ranks that touch its chunks -> all-gather of boundary blocks (`exchange_halo`), then every rank
  extracts triangles for the blocks it owns and rank 0 merges the buffers into the single-GPU canonical order
  (`gather_mesh`).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np

from . import capi

P0, P1, P2 = 73856093, 19349669, 83492791


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


def _all_gather_bytes(dist, payload: bytes, device) -> List[bytes]:
    """Variable-length all-gather: sizes first, then padded uint8 tensors (one collective each)."""
    import torch

    world = dist.get_world_size()
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if payload:
        buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    out = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(out, buf)
    return [bytes(o[:s].cpu().numpy().tobytes()) for o, s in zip(out, sizes)]


def _device_for(dist):
    import torch

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


class _DeviceArray:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}


def integrate(engine: capi.Engine, dist=None, n_frames_invalidate: int = -1):
    """One frame on a tile-sharded context.  On starve frames the library stops twice for an element-wise MIN of the
    per-pixel z-buffer over all ranks (the only data-path collective of the fusion loop; every n-th frame, 2.4 MB
    at 640x480): RCCL all-reduce over xGMI on the GPU box, gloo in the CPU tests."""
    import ctypes
    import torch

    pending = engine.integrate(n_frames_invalidate)
    while pending:
        ptr, n, on_device = engine.exchange_buffer()
        if dist is not None and dist.get_world_size() > 1:
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
    """Blocks on the surface of their chunk: the only ones a neighbouring chunk's marching cubes can read."""
    side = 1 << chunk_log2
    m = np.zeros(len(descs), dtype=bool)
    for ax in ("x", "y", "z"):
        l = descs[ax] & (side - 1)
        m |= (l == 0) | (l == side - 1)
    return m


def exchange_halo(engine: capi.Engine, dist, chunk_log2: int = 3) -> int:
    """All-gather of boundary blocks (16-byte desc + 512 x 12-byte voxels each) and import of the ones that are
    26-adjacent to a chunk this rank owns.  Returns the number of imported halo blocks."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return 0
    descs, voxels = engine.dump_blocks()
    mine = owner_of_blocks(np.stack([descs["x"], descs["y"], descs["z"]], 1), world, chunk_log2) == rank if len(descs) else np.zeros(0, bool)
    sel = mine & boundary_mask(descs, chunk_log2) if len(descs) else mine
    payload = descs[sel].tobytes() + voxels[sel].tobytes()
    header = np.array([int(sel.sum())], dtype=np.int64).tobytes()
    parts = _all_gather_bytes(dist, header + payload, _device_for(dist))
    imported = 0
    for r, blob in enumerate(parts):
        if r == rank:
            continue
        n = int(np.frombuffer(blob[:8], dtype=np.int64)[0])
        if n == 0:
            continue
        d = np.frombuffer(blob[8: 8 + 16 * n], dtype=capi.DESC_DTYPE)
        v = np.frombuffer(blob[8 + 16 * n: 8 + 16 * n + n * 512 * 12], dtype=capi.VOXEL_DTYPE).reshape(n, 512)
        xyz = np.stack([d["x"], d["y"], d["z"]], 1).astype(np.int64)
        need = np.zeros(n, dtype=bool)
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    if dx == dy == dz == 0:
                        continue
                    need |= owner_of_blocks(xyz + np.array([dx, dy, dz]), world, chunk_log2) == rank
        if need.any():
            engine.import_blocks(d[need], v[need])
            imported += int(need.sum())
    return imported


def gather_mesh(engine: capi.Engine, dist) -> Optional[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]:
    """Every rank extracts the triangles of the blocks it owns; rank 0 merges the per-block runs by block position
    (the canonical order) and runs the CPU mesh post-process.  Returns (triangles, V, F, C) on rank 0, None elsewhere."""
    rank = dist.get_rank()
    tris = engine.extract_triangles()
    descs, counts = engine.triangle_blocks()
    keep = counts > 0
    blob = (np.array([int(keep.sum()), int(tris.shape[0])], dtype=np.int64).tobytes() + descs[keep].tobytes()
            + counts[keep].tobytes() + tris.tobytes())
    parts = _all_gather_bytes(dist, blob, _device_for(dist))
    if rank != 0:
        return None
    runs = []  # (x, y, z, triangles of that block)
    for p in parts:
        nb, nt = (int(v) for v in np.frombuffer(p[:16], dtype=np.int64))
        d = np.frombuffer(p[16: 16 + 16 * nb], dtype=capi.DESC_DTYPE)
        c = np.frombuffer(p[16 + 16 * nb: 16 + 20 * nb], dtype=np.uint32)
        t = np.frombuffer(p[16 + 20 * nb: 16 + 20 * nb + nt * 72], dtype=capi.TRI_DTYPE).reshape(nt, 3)
        off = 0
        for i in range(nb):
            runs.append((int(d["x"][i]), int(d["y"][i]), int(d["z"][i]), t[off: off + int(c[i])]))
            off += int(c[i])
    runs.sort(key=lambda r: r[:3])
    merged = np.concatenate([r[3] for r in runs]) if runs else np.zeros((0, 3), dtype=capi.TRI_DTYPE)
    engine.process_triangles(merged)
    V, F, C = engine.extract_mesh()
    return merged, V, F, C
