"""Multi-GPU support: one process per GPU.  The reference is single-GPU (no NCCL/MPI/streams anywhere), so
this is new design.  Two ways to use N GPUs, and how they connect:

* tile sharding (`Params.shard_rank/shard_count/shard_chunk_log2`, `Engine.set_sharding`) — the result-identical mode.
  Every rank sees every frame but only inserts the blocks whose chunk it owns, so the union of the tables is
  bit-identical to the single-GPU map.  Exchange steps: an element-wise MIN all-reduce of the per-pixel z-buffer on
  starve frames (`integrate`), and before marching cubes an exchange of boundary blocks (`exchange_halo`): corner
  samples reach into neighbouring blocks, so each rank needs the blocks of other ranks that touch its chunks.  Every
  rank then extracts triangles for the blocks it owns and rank 0 merges the buffers into the single-GPU canonical
  order (`gather_mesh`).
* frame sharding (bench.py --gpus N, BASELINE.json configs[3]) — every rank fuses its own segment of the stream into
  its own sub-map owning everything; no collective in the per-frame loop.  `merge_submaps` then turns the N sub-maps
  into ONE tile-sharded map: every rank sends each block to the rank that owns its tile (all-to-all), the owner folds
  the sub-maps with combineVoxel's weighted mean in rank order, and from there on the map is a tile-sharded one
  (halo exchange, marching cubes, mesh gather as above).

Every function below takes a `group`, of one of two kinds:

* a `capi.Comm` — the PRODUCT path: RCCL over xGMI behind the C ABI (include/mrhash_comm.h, csrc/mrh_comm.h).  The functions
  are then one C call each (`mrh_comm_exchange_halo`, `mrh_comm_merge_submaps`, `mrh_comm_gather_mesh`; the starve
  all-reduce runs inside `mrh_integrate`): collectives on the library's own stream and device buffers, one HIP runtime in
  the process, no torch.  `rendezvous()` creates the communicator from the launcher's environment.
* the `torch.distributed` module with a gloo group — TEST infrastructure: the same protocol over host buffers, for the CPU
  tests (oracle engines, world size 2 .. 8) and for several HIP ranks sharing the one GPU of the test box (RCCL refuses two
  ranks on one device).  Blocks are packed / unpacked through the same `mrh_pack_blocks` / `mrh_unpack_blocks` calls.
"""
from __future__ import annotations

import ctypes
import os
import time
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


def rdzv_dir() -> str:
    """Directory of the rendezvous files: MRH_RDZV_DIR if given, else a per-user directory (mode 0700, owned by this user) under
    XDG_RUNTIME_DIR or /tmp — not the world-writable /tmp itself, where another local user could pre-create a file of a
    predictable name."""
    d = os.environ.get("MRH_RDZV_DIR")
    if d:
        return d
    base = os.environ.get("XDG_RUNTIME_DIR")
    d = os.path.join(base if base and os.path.isdir(base) else "/tmp", f"mrh_rdzv_u{os.getuid()}")
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"rendezvous directory {d} is not private to uid {os.getuid()}")
    return d


def launcher_start_time() -> float:
    """Wall-clock start of the parent process (the launcher every rank of a node shares: torch.distributed.run's agent, bench.py's
    own launcher, mpirun): a rendezvous file older than that is left over from an earlier run (a crashed job, a reused key or
    pid) and must not be believed.  0.0 when /proc does not say."""
    try:
        with open(f"/proc/{os.getppid()}/stat") as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])  # field 22 (starttime), counted after "pid (comm)"
        with open("/proc/stat") as f:
            btime = next(float(ln.split()[1]) for ln in f if ln.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except Exception:  # noqa: BLE001
        return 0.0


def publish_file(path: str, payload: bytes) -> None:
    """All of `payload` or nothing at `path`: written to a fresh private file (O_EXCL, mode 0600) and renamed."""
    tmp = f"{path}.{os.getpid()}.tmp"
    try:
        os.unlink(tmp)
    except FileNotFoundError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    try:
        os.write(fd, payload)
    finally:
        os.close(fd)
    os.replace(tmp, path)


def rendezvous(lib, rank: Optional[int] = None, world: Optional[int] = None, device_id: Optional[int] = None, timeout_s: float = 300.0) -> capi.Comm:
    """RCCL communicator from the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE as torch.distributed.run, mpirun
    wrappers and bench.py's own launcher set them).  Rank 0 removes whatever an earlier run left under the name, creates the
    ncclUniqueId and publishes its 128 bytes in a file under `rdzv_dir()` named after the launcher — parent pid, MASTER_PORT,
    TORCHELASTIC_RUN_ID — private (0600, O_EXCL), written to a temporary name and renamed, so a reader sees all of it or
    nothing; the other ranks of the node poll for it and believe only a file that is theirs and not older than their launcher
    (a stale id would leave them inside ncclCommInitRank for ever: the time limit covers the poll only).  One node, as the
    bench contract says; a multi-node launcher hands the id over itself and calls capi.Comm directly."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    device_id = int(os.environ.get("LOCAL_RANK", str(rank))) if device_id is None else device_id
    key = os.environ.get("MRH_RDZV_KEY") or f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
    path = os.path.join(rdzv_dir(), f"mrh_rdzv_{key}.id")
    if rank == 0:
        try:
            os.unlink(path)  # a leftover of an earlier run with the same key
        except FileNotFoundError:
            pass
        uid = capi.Comm.unique_id(lib)
        publish_file(path, uid)
    else:
        t0 = time.time()
        not_before = launcher_start_time() - 2.0  # clock granularity of /proc
        uid = b""
        while len(uid) != capi.COMM_ID_BYTES:
            try:
                st = os.stat(path)
                if st.st_uid == os.getuid() and st.st_mtime >= not_before:
                    with open(path, "rb") as f:
                        uid = f.read()
            except FileNotFoundError:
                uid = b""
            if len(uid) != capi.COMM_ID_BYTES:
                uid = b""
                if time.time() - t0 > timeout_s:
                    raise TimeoutError(f"rendezvous: rank 0 never published {path}")
                time.sleep(0.01)
    comm = capi.Comm(lib, uid, rank, world, device_id)  # collective: returns once every rank has joined
    if rank == 0:
        try:
            os.unlink(path)  # every rank has read it
        except OSError:
            pass
    return comm


def _is_comm(group) -> bool:
    return isinstance(group, capi.Comm)


def init_process_group(backend: str = "gloo"):
    """gloo group for the test paths: RANK / WORLD_SIZE / MASTER_* from the environment, 127.0.0.1 rendezvous."""
    import torch.distributed as dist

    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group(backend=backend)
    return dist


def _host_bytes(ptr: int, nbytes: int, on_device: bool):
    """uint8 CPU torch tensor holding `nbytes` from `ptr` (device memory is read back; host memory is viewed in place)."""
    import torch

    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8)
    if on_device:
        from . import hipmem

        return torch.from_numpy(np.frombuffer(hipmem.read(ptr, nbytes), dtype=np.uint8).copy())
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr)))


def _force_collectives() -> bool:
    """MRH_FORCE_COLLECTIVES=1: run every collective of the gloo path even in a one-rank group (tests)."""
    return os.environ.get("MRH_FORCE_COLLECTIVES") == "1"


def _all_gather_counts(dist, values: List[int]) -> np.ndarray:
    """[world, len(values)] int64: every rank's `values`."""
    import torch

    world = dist.get_world_size()
    mine = torch.tensor(values, dtype=torch.int64)
    out = torch.empty(world * len(values), dtype=torch.int64)
    dist.all_gather_into_tensor(out, mine)
    return out.numpy().reshape(world, len(values))


def _unpack(engine: capi.Engine, mode: int, buf, first_byte: int, n_records: int) -> int:
    """Hands records [first_byte, first_byte + n * REC) of a CPU uint8 tensor to the library (which stages them once)."""
    if n_records == 0:
        return 0
    return engine.unpack_blocks(mode, buf.data_ptr() + first_byte, n_records, False)


def world_size(group) -> int:
    return group.world if _is_comm(group) else group.get_world_size()


def rank_of(group) -> int:
    return group.rank if _is_comm(group) else group.get_rank()


def integrate(engine: capi.Engine, group=None, n_frames_invalidate: int = -1):
    """One frame on a tile-sharded context.  On starve frames the per-pixel z-buffer is min-reduced over all ranks, twice
    (the only data-path collective of the fusion loop; every n-th frame, 2.4 MB at 640x480).  With a `capi.Comm` attached to
    the engine the library does that itself — ncclAllReduce on its own stream, no host synchronisation, `integrate` is one
    enqueue.  Over gloo the library stops (`MRH_PENDING_EXCHANGE`) and the buffer is reduced here through the host."""
    pending = engine.integrate(n_frames_invalidate)
    if not pending:
        return
    if group is None or _is_comm(group):
        raise RuntimeError("parallel.integrate: the context stopped for an exchange but no gloo group was given "
                           "(with a capi.Comm, attach it to the engine: Engine.attach_comm)")
    import torch

    dist = group
    while pending:
        ptr, n, on_device = engine.exchange_buffer()
        if dist.get_world_size() > 1 or _force_collectives():
            if on_device:
                from . import hipmem

                h = torch.from_numpy(np.frombuffer(hipmem.read(ptr, n * 8), dtype=np.int64).copy())
                dist.all_reduce(h, op=dist.ReduceOp.MIN)
                hipmem.write(ptr, h.numpy())
            else:
                arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int64)), shape=(n,))
                dist.all_reduce(torch.from_numpy(arr), op=dist.ReduceOp.MIN)
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


def exchange_halo(engine: capi.Engine, group) -> int:
    """Boundary blocks of every rank to every rank.  The library selects the owned blocks on the surface of their chunk and
    packs their records (`mrh_pack_blocks(MRH_PACK_HALO)`); they reach every other rank, which keeps the ones 26-adjacent to
    a position it owns (`mrh_unpack_blocks(MRH_UNPACK_HALO)`).  capi.Comm: `mrh_comm_exchange_halo` — grouped ncclSend /
    ncclRecv between every pair, device to device.  gloo: one padded all-gather through host tensors.
    Terminal for fusion until `drop_halo`: the library refuses to integrate while halo blocks are present.
    Returns the number of halo blocks this rank took."""
    if _is_comm(group):
        return engine.comm_exchange_halo()
    import torch

    dist = group
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1 and not _force_collectives():
        return 0
    ptr, n, on_device = engine.pack_blocks(capi.PACK_HALO)
    counts = _all_gather_counts(dist, [n])[:, 0]
    mx = int(counts.max())
    if mx == 0:
        return 0
    send = torch.zeros(mx * REC, dtype=torch.uint8)
    if n:
        send[: n * REC].copy_(_host_bytes(ptr, n * REC, on_device))
    recv = torch.empty(world * mx * REC, dtype=torch.uint8)
    dist.all_gather_into_tensor(recv, send)
    taken = 0
    for r in range(world):
        if r != rank:
            taken += _unpack(engine, capi.UNPACK_HALO, recv, r * mx * REC, int(counts[r]))
    return taken


def drop_halo(engine: capi.Engine) -> int:
    """Removes the blocks `exchange_halo` brought in (after the extraction that needed them); fusion may continue."""
    return engine.drop_blocks(capi.DROP_HALO)


def merge_submaps(engine: capi.Engine, group, chunk_log2: int = 3) -> dict:
    """Frame-sharded sub-maps -> one tile-sharded map.  Every rank packs, per destination, the blocks whose tile that
    rank owns (`MRH_PACK_OWNER`), an all-to-all moves them, the local map is emptied and the N incoming sub-maps are folded
    in rank order with combineVoxel's weighted mean (`MRH_UNPACK_MERGE`, vhu.cuh:167-181).  capi.Comm:
    `mrh_comm_merge_submaps` (grouped ncclSend / ncclRecv with the true split sizes); gloo has no all-to-all on CPU tensors,
    so the test path all-gathers and selects.  TSDF values and weights of the result equal a single-GPU fusion of all frames
    up to the rounding of the running mean (while weights stay below the clamp); colours are order-dependent (50/50 blend).
    Returns {"sent": blocks sent, "received": blocks received, "bytes": payload bytes through the collective}."""
    if _is_comm(group):
        return engine.comm_merge_submaps(chunk_log2)
    import torch

    dist = group
    world, rank = dist.get_world_size(), dist.get_rank()
    engine.set_sharding(rank, world, chunk_log2)
    if world == 1 and not _force_collectives():
        return {"sent": 0, "received": 0, "bytes": 0}
    parts, out_counts = [], []
    for dest in range(world):
        ptr, n, on_device = engine.pack_blocks(capi.PACK_OWNER, dest)
        out_counts.append(n)
        parts.append(_host_bytes(ptr, n * REC, on_device).clone())  # the next pack reuses the library's buffer
    counts = _all_gather_counts(dist, out_counts)  # counts[src, dest]
    in_counts = [int(counts[src, rank]) for src in range(world)]
    send = torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8)
    engine.drop_blocks(capi.DROP_ALL)  # the owned blocks come back through the fold, at this rank's position in the order
    mx = int(counts.sum(axis=1).max())
    padded = torch.zeros(max(mx, 1) * REC, dtype=torch.uint8)
    padded[: send.numel()].copy_(send)
    gathered = torch.empty(world * max(mx, 1) * REC, dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, padded)
    for src in range(world):  # rank order: the fold is deterministic for a given world size
        before = int(counts[src, :rank].sum())
        _unpack(engine, capi.UNPACK_MERGE, gathered, (src * max(mx, 1) + before) * REC, in_counts[src])
    sent = int(sum(out_counts)) - out_counts[rank]
    return {"sent": sent, "received": int(sum(in_counts)) - in_counts[rank], "bytes": sent * REC}


def _root_mesh(engine: capi.Engine):
    """(merged triangles, V, F, C) out of the library after a run merge."""
    V, F, C = engine.extract_mesh()
    mptr, mn, mdev = engine.triangles_device()
    if mn:
        if mdev:
            from . import hipmem

            raw = hipmem.read(mptr, mn * 72)
        else:
            raw = ctypes.string_at(mptr, mn * 72)
        merged = np.frombuffer(raw, dtype=capi.TRI_DTYPE).reshape(mn, 3).copy()
    else:
        merged = np.zeros((0, 3), dtype=capi.TRI_DTYPE)
    return merged, V, F, C


def gather_mesh(engine: capi.Engine, group) -> Optional[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]:
    """Every rank extracts the triangles of the blocks it owns; rank 0 brings the per-block runs of all ranks into the
    canonical single-GPU order (block position) and runs the mesh post-process (`mrh_process_triangle_runs`).  capi.Comm:
    `mrh_comm_gather_mesh` — the soups travel device to device, only 20 bytes of descriptor + count per block are metadata.
    gloo: padded all-gathers through host tensors.  Returns (triangles, V, F, C) on rank 0, None elsewhere."""
    if _is_comm(group):
        engine.comm_gather_mesh(0)
        return _root_mesh(engine) if group.rank == 0 else None
    import torch

    dist = group
    world, rank = dist.get_world_size(), dist.get_rank()
    engine.extract_triangles(soup=False)  # the soup stays in the library's memory
    ptr, nt, on_device = engine.triangles_device()
    descs, counts = engine.triangle_blocks()
    keep = counts > 0
    nb = int(keep.sum())
    sizes = _all_gather_counts(dist, [nb, nt])
    max_b, max_t = max(int(sizes[:, 0].max()), 1), max(int(sizes[:, 1].max()), 1)
    meta = np.zeros(max_b * 20, np.uint8)
    if nb:
        meta[: nb * 16] = np.frombuffer(descs[keep].tobytes(), np.uint8)
        meta[max_b * 16: max_b * 16 + nb * 4] = np.frombuffer(counts[keep].astype(np.uint32).tobytes(), np.uint8)
    all_meta = torch.empty(world * max_b * 20, dtype=torch.uint8)
    dist.all_gather_into_tensor(all_meta, torch.from_numpy(meta))
    send = torch.zeros(max_t * 72, dtype=torch.uint8)
    if nt:
        send[: nt * 72].copy_(_host_bytes(ptr, nt * 72, on_device))  # read back before anything else touches the soup
    recv = torch.empty(world * max_t * 72, dtype=torch.uint8)
    dist.all_gather_into_tensor(recv, send)
    if rank != 0:
        return None
    host_meta = all_meta.numpy()
    all_d, all_c = [], []
    for r in range(world):
        b = int(sizes[r, 0])
        m = host_meta[r * max_b * 20: (r + 1) * max_b * 20]
        all_d.append(np.frombuffer(m[: 16 * b].tobytes(), dtype=capi.DESC_DTYPE))
        all_c.append(np.frombuffer(m[max_b * 16: max_b * 16 + 4 * b].tobytes(), dtype=np.uint32))
    # the ranks' soups, closed up (each segment of `recv` is padded to max_t triangles)
    total = int(sizes[:, 1].sum())
    packed = torch.empty(max(total, 1) * 72, dtype=torch.uint8)
    off = 0
    for r in range(world):
        t = int(sizes[r, 1])
        if t:
            packed[off * 72: (off + t) * 72].copy_(recv[r * max_t * 72: (r * max_t + t) * 72])
        off += t
    engine.process_triangle_runs(np.concatenate(all_d), np.concatenate(all_c), packed.data_ptr(), total, False)
    return _root_mesh(engine)
