"""ctypes binding of the C ABI declared in include/mrhash_hip.h.

`Engine` drives any shared library that implements that ABI.  The product library is
`mrhash_amd/csrc/libmrhash_hip.so` (hand-written gfx950 HIP kernels); tests additionally load
the CPU oracle (`oracle/_build/libmrh_oracle.so`) through the very same class so that parity
tests run identical host code against both.  Nothing in this module falls back from one to
the other: `load_hip()` raises if the HIP library is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, replace
from typing import Optional, Tuple

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_LIB_PATH = os.path.join(_ROOT, "mrhash_amd", "csrc", "libmrhash_hip.so")

MRH_ABI_VERSION = 3

MRH_OK = 0
MRH_PENDING_EXCHANGE = 1
MRH_ERR_INVALID_ARG = -1
MRH_ERR_DEVICE = -2
MRH_ERR_NO_DEVICE = -3
MRH_ERR_CAPACITY = -4
MRH_ERR_STATE = -5
MRH_ERR_UNSUPPORTED = -6
MRH_ERR_OUT_OF_RANGE = -7

PINHOLE, SPHERICAL = 0, 1


class MrhParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("sdf_truncation", C.c_float),
        ("sdf_truncation_scale", C.c_float),
        ("integration_weight_sample", C.c_int32),
        ("integration_weight_max", C.c_int32),
        ("virtual_voxel_size", C.c_float),
        ("n_frames_invalidate_voxels", C.c_int32),
        ("voxel_extents_scale", C.c_int32),
        ("marching_cubes_threshold", C.c_float),
        ("min_weight_threshold", C.c_uint8),
        ("projective_sdf", C.c_uint8),
        ("reserved0", C.c_uint8 * 2),
        ("min_depth", C.c_float),
        ("max_depth", C.c_float),
        ("sdf_var_threshold", C.c_float),
        ("vertices_merging_threshold", C.c_float),
        ("num_sdf_blocks", C.c_uint64),
        ("hash_slots", C.c_uint64),
        ("max_triangles", C.c_uint64),
        ("device_id", C.c_int32),
        ("shard_rank", C.c_int32),
        ("shard_count", C.c_int32),
        ("shard_chunk_log2", C.c_int32),
    ]


class MrhStats(C.Structure):
    _fields_ = [
        ("frames_integrated", C.c_uint64),
        ("num_sdf_blocks", C.c_uint64),
        ("occupied_fine", C.c_uint64),
        ("occupied_coarse", C.c_uint64),
        ("free_fine", C.c_int64),
        ("free_coarse", C.c_int64),
        ("last_compact_blocks", C.c_uint64),
        ("last_updated_voxels", C.c_uint64),
        ("last_inserted_blocks", C.c_uint64),
        ("last_freed_blocks", C.c_uint64),
        ("total_updated_voxels", C.c_uint64),
        ("total_compact_blocks", C.c_uint64),
        ("last_triangles", C.c_uint64),
        ("last_integrate_kernel_ms", C.c_float),
        ("sum_integrate_kernel_ms", C.c_float),
        ("n_integrate_kernel", C.c_uint64),
        ("error_flags", C.c_uint32),
        ("reserved", C.c_uint32),
        ("hash_slots", C.c_uint64),
        ("tombstones", C.c_uint64),
        ("max_probe_length", C.c_uint32),
        ("rehash_count", C.c_uint32),
        ("last_mc_count_ms", C.c_float),
        ("last_mc_emit_ms", C.c_float),
        ("last_mc_blocks", C.c_uint64),
        ("sum_front_kernel_ms", C.c_float),
        ("reserved1", C.c_uint32),
        ("n_front_kernel", C.c_uint64),
    ]


VOXEL_DTYPE = np.dtype(
    [("sdf", "<f4"), ("sum_squared", "<f4"), ("rgb", "u1", (3,)), ("weight", "u1")], align=False
)
assert VOXEL_DTYPE.itemsize == 12
DESC_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("z", "<i4"), ("resolution", "<i4")])
RECORD_DTYPE = np.dtype([("desc", DESC_DTYPE), ("voxels", VOXEL_DTYPE, (512,))])  # mrh_block_record
RECORD_BYTES = 6160
PACK_HALO, PACK_OWNER = 0, 1
UNPACK_HALO, UNPACK_MERGE = 0, 1
DROP_HALO, DROP_FOREIGN, DROP_ALL = 0, 1, 2
TRI_DTYPE = np.dtype([("p", "<f4", (3,)), ("c", "<f4", (3,))])  # one vertex; a triangle is 3 of them
SEED_DTYPE = np.dtype([("p", "<f4", (3,)), ("scale", "<f4"), ("rgb", "u1", (3,)), ("pad", "u1")])  # mrh_splat_seed, 20 bytes
LEAF_DTYPE = np.dtype([("x0", "<i4"), ("y0", "<i4"), ("width", "<i4"), ("height", "<i4")])  # mrh_qtree_leaf
assert TRI_DTYPE.itemsize == 24

# every symbol include/mrhash_hip.h declares
ABI_SYMBOLS = (
    "mrh_create mrh_destroy mrh_reset mrh_last_error mrh_set_camera mrh_set_pose mrh_upload_depth "
    "mrh_upload_rgb mrh_set_depth_device mrh_set_rgb_device mrh_integrate mrh_integrate_resume mrh_exchange_buffer mrh_sync "
    "mrh_upload_points mrh_upload_normals mrh_set_points_device mrh_integrate_points mrh_set_scan_layout mrh_detect_scan_layout mrh_stream_out mrh_get_free_blocks "
    "mrh_splat_seeds mrh_get_qtree_leaves mrh_peek_free_blocks mrh_peek_error_flags "
    "mrh_set_sharding mrh_pack_blocks mrh_unpack_blocks mrh_drop_blocks "
    "mrh_extract_triangles mrh_extract_mesh mrh_mesh_merge_begin mrh_mesh_merge_end mrh_get_stats mrh_set_profile mrh_dump_blocks "
    "mrh_get_voxel mrh_import_blocks mrh_get_triangle_blocks mrh_get_triangles_device mrh_process_triangle_runs mrh_process_triangles mrh_selftest_division mrh_version"
).split()


# every symbol include/mrhash_comm.h declares (the HIP library only: the oracle has no communicator)
COMM_SYMBOLS = (
    "mrh_comm_unique_id mrh_comm_create mrh_comm_destroy mrh_comm_last_error mrh_comm_size mrh_comm_barrier mrh_comm_allreduce_f64 "
    "mrh_comm_allgather_bytes mrh_comm_attach mrh_comm_exchange_halo mrh_comm_merge_submaps mrh_comm_gather_mesh mrh_comm_phase_times mrh_comm_status"
).split()
COMM_ID_BYTES = 128
COMM_SUM, COMM_MAX, COMM_MIN = 0, 1, 2


class MrhCommStatus(C.Structure):
    _fields_ = [("rccl_ranks", C.c_int), ("rccl_rank", C.c_int), ("rccl_device", C.c_int), ("rccl_version", C.c_int), ("async_error", C.c_int),
                ("async_error_string", C.c_char * 64), ("library_path", C.c_char * 256)]


class MrhCommMergeInfo(C.Structure):
    _fields_ = [("blocks_sent", C.c_uint64), ("blocks_received", C.c_uint64), ("bytes_sent", C.c_uint64), ("blocks_kept", C.c_uint64)]


class MrhCommPhases(C.Structure):
    _fields_ = [("pack_ms", C.c_float), ("counts_ms", C.c_float), ("collective_ms", C.c_float), ("unpack_ms", C.c_float),
                ("bytes_out", C.c_uint64), ("bytes_in", C.c_uint64), ("allreduce_ms_sum", C.c_float), ("allreduce_count", C.c_uint32)]


class MrhError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mrh error {code}: {msg}")
        self.code = code


def _declare(lib: C.CDLL) -> C.CDLL:
    P = C.POINTER
    lib.mrh_create.argtypes = [P(MrhParams), P(C.c_void_p)]
    lib.mrh_destroy.argtypes = [C.c_void_p]
    lib.mrh_reset.argtypes = [C.c_void_p]
    lib.mrh_last_error.argtypes = [C.c_void_p]
    lib.mrh_last_error.restype = C.c_char_p
    lib.mrh_set_camera.argtypes = [C.c_void_p] + [C.c_float] * 4 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
    lib.mrh_set_pose.argtypes = [C.c_void_p, P(C.c_float), P(C.c_float)]
    lib.mrh_upload_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.mrh_upload_rgb.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.mrh_set_depth_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.mrh_set_rgb_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.mrh_integrate.argtypes = [C.c_void_p, C.c_int]
    lib.mrh_upload_points.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.mrh_upload_normals.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.mrh_set_points_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.mrh_integrate_points.argtypes = [C.c_void_p, C.c_int]
    lib.mrh_set_scan_layout.argtypes = [C.c_void_p, C.c_int]
    lib.mrh_detect_scan_layout.argtypes = [C.c_void_p, C.c_uint64]
    lib.mrh_get_free_blocks.argtypes = [C.c_void_p, P(C.c_int64), P(C.c_int64)]
    lib.mrh_peek_free_blocks.argtypes = [C.c_void_p, P(C.c_int64), P(C.c_int64), P(C.c_uint64)]
    lib.mrh_stream_out.argtypes = [C.c_void_p, P(C.c_float), C.c_float, C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_uint64)]
    lib.mrh_splat_seeds.argtypes = [C.c_void_p, C.c_float, C.c_int, P(C.c_void_p), P(C.c_uint64)]
    lib.mrh_get_qtree_leaves.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_uint64)]
    lib.mrh_peek_error_flags.argtypes = [C.c_void_p, P(C.c_uint32)]
    lib.mrh_set_sharding.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.mrh_pack_blocks.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_void_p), P(C.c_uint64), P(C.c_int)]
    lib.mrh_unpack_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int, P(C.c_uint64)]
    lib.mrh_drop_blocks.argtypes = [C.c_void_p, C.c_int, P(C.c_uint64)]
    lib.mrh_integrate_resume.argtypes = [C.c_void_p]
    lib.mrh_exchange_buffer.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_uint64), P(C.c_int)]
    lib.mrh_sync.argtypes = [C.c_void_p]
    lib.mrh_extract_triangles.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_uint64)]
    lib.mrh_extract_mesh.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_uint64), P(C.c_void_p), P(C.c_uint64), P(C.c_void_p)]
    lib.mrh_mesh_merge_begin.argtypes = [C.c_void_p]
    lib.mrh_mesh_merge_end.argtypes = [C.c_void_p, P(C.c_uint64)]
    lib.mrh_get_stats.argtypes = [C.c_void_p, P(MrhStats)]
    lib.mrh_set_profile.argtypes = [C.c_void_p, C.c_int]
    lib.mrh_dump_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_uint64)]
    lib.mrh_get_voxel.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, P(C.c_int)]
    lib.mrh_import_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.mrh_get_triangle_blocks.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_void_p), P(C.c_uint64)]
    lib.mrh_process_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.mrh_get_triangles_device.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_uint64), P(C.c_int)]
    lib.mrh_process_triangle_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
    lib.mrh_selftest_division.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, P(C.c_uint64)]
    lib.mrh_version.argtypes = []
    lib.mrh_version.restype = C.c_char_p
    for name in ABI_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("mrh_last_error", "mrh_version"):
            fn.restype = C.c_int
    if hasattr(lib, "mrh_comm_create"):  # include/mrhash_comm.h
        lib.mrh_comm_unique_id.argtypes = [C.c_void_p]
        lib.mrh_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(C.c_void_p)]
        lib.mrh_comm_destroy.argtypes = [C.c_void_p]
        lib.mrh_comm_last_error.argtypes = [C.c_void_p]
        lib.mrh_comm_last_error.restype = C.c_char_p
        lib.mrh_comm_size.argtypes = [C.c_void_p, P(C.c_int), P(C.c_int)]
        lib.mrh_comm_barrier.argtypes = [C.c_void_p]
        lib.mrh_comm_allreduce_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        lib.mrh_comm_allgather_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        lib.mrh_comm_attach.argtypes = [C.c_void_p, C.c_void_p]
        lib.mrh_comm_exchange_halo.argtypes = [C.c_void_p, P(C.c_uint64)]
        lib.mrh_comm_merge_submaps.argtypes = [C.c_void_p, C.c_int, P(MrhCommMergeInfo)]
        lib.mrh_comm_gather_mesh.argtypes = [C.c_void_p, C.c_int, P(C.c_uint64)]
        lib.mrh_comm_phase_times.argtypes = [C.c_void_p, P(MrhCommPhases)]
        lib.mrh_comm_status.argtypes = [C.c_void_p, P(MrhCommStatus)]
        for name in COMM_SYMBOLS:
            if name != "mrh_comm_last_error":
                getattr(lib, name).restype = C.c_int
    return lib


def load_library(path: str) -> C.CDLL:
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found - build it first (python -c 'import __graft_entry__ as g; g.build()')"
        )
    return _declare(C.CDLL(path))


_hip_lib: Optional[C.CDLL] = None


def load_hip() -> C.CDLL:
    """The product library. Raises (never falls back) when it has not been built."""
    global _hip_lib
    if _hip_lib is None:
        from ._runtime import torch_first

        torch_first()  # see _runtime.py: torch's HIP runtime has to initialise before this library's
        _hip_lib = load_library(HIP_LIB_PATH)
    return _hip_lib


class Comm:
    """RCCL communicator behind the C ABI (include/mrhash_comm.h): one per process / GPU.  `unique_id()` on one rank, the 128
    bytes handed to the others by the host (mrhash_amd.parallel.rendezvous), then `Comm(lib, id, rank, world, device)`."""

    def __init__(self, lib: C.CDLL, uid: bytes, rank: int, world: int, device_id: int = 0):
        self.lib, self.rank, self.world, self.device_id = lib, rank, world, device_id
        self._comm = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        rc = lib.mrh_comm_create(buf, rank, world, device_id, C.byref(self._comm))
        if rc != MRH_OK:
            raise MrhError(rc, lib.mrh_comm_last_error(None).decode())

    @staticmethod
    def unique_id(lib: C.CDLL) -> bytes:
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        rc = lib.mrh_comm_unique_id(buf)
        if rc != MRH_OK:
            raise MrhError(rc, lib.mrh_comm_last_error(None).decode())
        return bytes(buf)

    def _check(self, rc: int):
        if rc != MRH_OK:
            raise MrhError(rc, self.lib.mrh_comm_last_error(self._comm).decode())

    def barrier(self):
        self._check(self.lib.mrh_comm_barrier(self._comm))

    def allreduce(self, values, op: int = COMM_SUM) -> np.ndarray:
        a = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self._check(self.lib.mrh_comm_allreduce_f64(self._comm, a.ctypes.data, a.size, op))
        return a

    def allgather_i64(self, values) -> np.ndarray:
        """[world, len(values)] int64: every rank's `values`."""
        a = np.ascontiguousarray(np.atleast_1d(values), dtype=np.int64)
        out = np.empty((self.world, a.size), dtype=np.int64)
        self._check(self.lib.mrh_comm_allgather_bytes(self._comm, a.ctypes.data, a.nbytes, out.ctypes.data))
        return out

    def status(self) -> dict:
        """What RCCL itself says about this communicator (mrh_comm_status): ranks it sees, its rank and device, the pending
        asynchronous error, version and library path."""
        st = MrhCommStatus()
        self._check(self.lib.mrh_comm_status(self._comm, C.byref(st)))
        return {"rccl_ranks": int(st.rccl_ranks), "rccl_rank": int(st.rccl_rank), "rccl_device": int(st.rccl_device), "rccl_version": int(st.rccl_version),
                "async_error": int(st.async_error), "async_error_string": st.async_error_string.decode(), "library_path": st.library_path.decode()}

    def close(self):
        if self._comm:
            self.lib.mrh_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class Params:
    sdf_truncation: float
    sdf_truncation_scale: float = 0.0
    integration_weight_sample: int = 1
    integration_weight_max: int = 255
    virtual_voxel_size: float = 0.01
    n_frames_invalidate_voxels: int = 0
    voxel_extents_scale: int = 1
    marching_cubes_threshold: float = 1.5
    min_weight_threshold: int = 5
    projective_sdf: bool = True
    min_depth: float = 0.01
    max_depth: float = 30.0
    sdf_var_threshold: float = 0.0
    vertices_merging_threshold: float = 0.0
    num_sdf_blocks: int = 0
    hash_slots: int = 0
    max_triangles: int = 0
    device_id: int = 0
    shard_rank: int = 0
    shard_count: int = 1
    shard_chunk_log2: int = 0

    def to_c(self) -> MrhParams:
        p = MrhParams()
        p.abi_version = MRH_ABI_VERSION
        for f in (
            "sdf_truncation sdf_truncation_scale integration_weight_sample integration_weight_max "
            "virtual_voxel_size n_frames_invalidate_voxels voxel_extents_scale marching_cubes_threshold "
            "min_weight_threshold min_depth max_depth sdf_var_threshold vertices_merging_threshold "
            "num_sdf_blocks hash_slots max_triangles device_id shard_rank shard_count shard_chunk_log2"
        ).split():
            setattr(p, f, getattr(self, f))
        p.projective_sdf = 1 if self.projective_sdf else 0
        return p


class Engine:
    """One fusion context behind the C ABI (HIP product library or, in tests, the oracle)."""

    def __init__(self, lib: C.CDLL, params: Params):
        self.lib = lib
        # a private copy: set_sharding / comm_merge_submaps record the context's sharding here, and a caller that builds
        # several engines from one Params object must not find the later ones sharded like the first
        self.params = replace(params)
        self._ctx = C.c_void_p()
        cp = params.to_c()
        rc = lib.mrh_create(C.byref(cp), C.byref(self._ctx))
        if rc != MRH_OK:
            raise MrhError(rc, lib.mrh_last_error(None).decode())
        self._keep = []  # device tensors referenced by *_device setters
        self._comm_ref = None  # keeps an attached Comm alive

    # -- helpers -------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != MRH_OK:
            raise MrhError(rc, self.lib.mrh_last_error(self._ctx).decode())

    def close(self):
        if self._ctx:
            self.lib.mrh_destroy(self._ctx)
            self._ctx = C.c_void_p()
            self._comm_ref = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- inputs --------------------------------------------------------------------------------
    def set_camera(self, fx, fy, cx, cy, rows, cols, min_depth, max_depth, model=PINHOLE):
        self._check(self.lib.mrh_set_camera(self._ctx, fx, fy, cx, cy, rows, cols, min_depth, max_depth, model))

    def set_pose(self, R: np.ndarray, t: np.ndarray):
        R = np.ascontiguousarray(R, dtype=np.float32).reshape(9)
        t = np.ascontiguousarray(t, dtype=np.float32).reshape(3)
        self._check(
            self.lib.mrh_set_pose(self._ctx, R.ctypes.data_as(C.POINTER(C.c_float)), t.ctypes.data_as(C.POINTER(C.c_float)))
        )

    def upload_depth(self, depth: np.ndarray):
        if depth.ndim != 2:
            raise RuntimeError("setDepthImage|input should be a 2D numpy array")
        d = np.ascontiguousarray(depth, dtype=np.float32)
        self._check(self.lib.mrh_upload_depth(self._ctx, d.ctypes.data, d.shape[0], d.shape[1]))

    def upload_rgb(self, rgb: np.ndarray):
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise RuntimeError("setRGBImage|input should be a 3D numpy array with 3 channels")
        r = np.ascontiguousarray(rgb, dtype=np.uint8)
        self._check(self.lib.mrh_upload_rgb(self._ctx, r.ctypes.data, r.shape[0], r.shape[1]))

    def set_depth_device(self, ptr: int, rows: int, cols: int):
        self._check(self.lib.mrh_set_depth_device(self._ctx, ptr, rows, cols))

    def set_rgb_device(self, ptr: int, rows: int, cols: int):
        self._check(self.lib.mrh_set_rgb_device(self._ctx, ptr, rows, cols))

    # -- hot path ------------------------------------------------------------------------------
    def upload_points(self, xyz: np.ndarray):
        """GeoWrapper.setPointCloud: float32 [N, 3] points in the sensor frame (copied)."""
        if xyz.ndim != 2 or xyz.shape[1] != 3:
            raise RuntimeError("GeoWrapper::setPointCloud|input should be a 2D numpy array with 3 columns")
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        self._check(self.lib.mrh_upload_points(self._ctx, a.ctypes.data, a.shape[0]))

    def upload_normals(self, nxyz: np.ndarray):
        """One normal per point of the current scan, float32 [N, 3] (copied): the normal-direction SDF (projective_sdf = False)."""
        a = np.ascontiguousarray(nxyz, dtype=np.float32).reshape(-1, 3)
        self._check(self.lib.mrh_upload_normals(self._ctx, a.ctypes.data, a.shape[0]))

    def set_points_device(self, ptr: int, n: int):
        self._check(self.lib.mrh_set_points_device(self._ctx, ptr, n))

    def set_scan_layout(self, row_len: int):
        """Points per row of the caller's organised scans (0: look at host clouds, < 0: never): a hint for the beam order, never for the result."""
        self._check(self.lib.mrh_set_scan_layout(self._ctx, int(row_len)))

    def integrate_points(self, n_frames_invalidate: int = -1) -> bool:
        """One scan.  Returns True when a sharded context stopped for the starve z-buffer reduction (see integrate())."""
        rc = self.lib.mrh_integrate_points(self._ctx, n_frames_invalidate)
        if rc == MRH_PENDING_EXCHANGE:
            return True
        self._check(rc)
        return False

    def integrate(self, n_frames_invalidate: int = -1) -> bool:
        """Enqueues one frame.  Returns True when a sharded context stopped for a min-reduction over ranks
        (MRH_PENDING_EXCHANGE): reduce `exchange_buffer()` and call `integrate_resume()` until it returns False
        (mrhash_amd.parallel.integrate does this)."""
        rc = self.lib.mrh_integrate(self._ctx, n_frames_invalidate)
        if rc == MRH_PENDING_EXCHANGE:
            return True
        self._check(rc)
        return False

    def integrate_resume(self) -> bool:
        rc = self.lib.mrh_integrate_resume(self._ctx)
        if rc == MRH_PENDING_EXCHANGE:
            return True
        self._check(rc)
        return False

    def exchange_buffer(self) -> Tuple[int, int, bool]:
        """(pointer, number of int64 elements, is_device_memory) of the buffer awaiting a MIN all-reduce."""
        ptr, n, dev = C.c_void_p(), C.c_uint64(), C.c_int()
        self._check(self.lib.mrh_exchange_buffer(self._ctx, C.byref(ptr), C.byref(n), C.byref(dev)))
        return int(ptr.value), int(n.value), bool(dev.value)

    def sync(self):
        self._check(self.lib.mrh_sync(self._ctx))

    def reset(self):
        self._check(self.lib.mrh_reset(self._ctx))

    def set_profile(self, enabled: bool):
        self._check(self.lib.mrh_set_profile(self._ctx, 1 if enabled else 0))

    def stats(self) -> MrhStats:
        s = MrhStats()
        self._check(self.lib.mrh_get_stats(self._ctx, C.byref(s)))
        return s

    # -- outputs -------------------------------------------------------------------------------
    def extract_triangles(self, soup: bool = True):
        """[T, 3] structured array of vertices (p[3], c[3]) in canonical triangle order.  soup=False: marching cubes and
        the mesh post-process run, the triangle soup stays on the device and only the triangle count is returned."""
        ptr = C.c_void_p()
        n = C.c_uint64()
        if not soup:
            self._check(self.lib.mrh_extract_triangles(self._ctx, None, C.byref(n)))
            return int(n.value)
        self._check(self.lib.mrh_extract_triangles(self._ctx, C.byref(ptr), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 3), dtype=TRI_DTYPE)
        buf = (C.c_char * (n.value * 72)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=TRI_DTYPE).reshape(n.value, 3).copy()

    def extract_mesh(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(V f64[nv,3], F i32[nf,3], C f64[nv,3]) of the last extract_triangles()."""
        pv, pf, pc = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nv, nf = C.c_uint64(), C.c_uint64()
        self._check(self.lib.mrh_extract_mesh(self._ctx, C.byref(pv), C.byref(nv), C.byref(pf), C.byref(nf), C.byref(pc)))

        def arr(p, n, dt):
            if n == 0 or not p.value:
                return np.zeros((0, 3), dtype=dt)
            sz = n * 3 * np.dtype(dt).itemsize
            return np.frombuffer((C.c_char * sz).from_address(p.value), dtype=dt).reshape(n, 3).copy()

        return arr(pv, nv.value, np.float64), arr(pf, nf.value, np.int32), arr(pc, nv.value, np.float64)

    def mesh_merge_begin(self):
        """MeshExtractor::merge_mesh_: until mesh_merge_end() every extract_triangles() adds to the running mesh."""
        self._check(self.lib.mrh_mesh_merge_begin(self._ctx))

    def mesh_merge_end(self) -> int:
        n = C.c_uint64()
        self._check(self.lib.mrh_mesh_merge_end(self._ctx, C.byref(n)))
        return int(n.value)

    def free_blocks(self) -> Tuple[int, int]:
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.mrh_get_free_blocks(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def splat_seeds(self, qtree_thresh: float = 0.1, qtree_min_pixel_size: int = 1) -> np.ndarray:
        """3DGS splat seeds of the current frame (call after integrate()): quad-tree over the colour image, one seed
        per leaf whose centre lands in a voxel of weight 1.  Returns a SEED_DTYPE array in canonical (leaf) order."""
        ptr, n = C.c_void_p(), C.c_uint64()
        self._check(self.lib.mrh_splat_seeds(self._ctx, qtree_thresh, qtree_min_pixel_size, C.byref(ptr), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, SEED_DTYPE)
        return np.frombuffer((C.c_char * (n.value * 20)).from_address(ptr.value), dtype=SEED_DTYPE).copy()

    def qtree_leaves(self) -> np.ndarray:
        """Leaves of the quad-tree of the last splat_seeds() call, canonical order (LEAF_DTYPE)."""
        ptr, n = C.c_void_p(), C.c_uint64()
        self._check(self.lib.mrh_get_qtree_leaves(self._ctx, C.byref(ptr), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, LEAF_DTYPE)
        return np.frombuffer((C.c_char * (n.value * 16)).from_address(ptr.value), dtype=LEAF_DTYPE).copy()

    def peek_free_blocks(self) -> Tuple[int, int, int]:
        """(free fine, free coarse, frames behind) without waiting for the device (mrh_peek_free_blocks)."""
        a, b, n = C.c_int64(), C.c_int64(), C.c_uint64()
        self._check(self.lib.mrh_peek_free_blocks(self._ctx, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def peek_error_flags(self) -> int:
        f = C.c_uint32()
        self._check(self.lib.mrh_peek_error_flags(self._ctx, C.byref(f)))
        return int(f.value)

    # -- multi-GPU: RCCL behind the C ABI (include/mrhash_comm.h) -----------------------------------
    def attach_comm(self, comm: Optional["Comm"]):
        """While attached, a tile-sharded context runs the starve all-reduces itself (integrate() never returns True)."""
        self._check(self.lib.mrh_comm_attach(self._ctx, comm._comm if comm is not None else None))
        self._comm_ref = comm

    def comm_exchange_halo(self) -> int:
        n = C.c_uint64()
        self._check(self.lib.mrh_comm_exchange_halo(self._ctx, C.byref(n)))
        return int(n.value)

    def comm_merge_submaps(self, chunk_log2: int = 3) -> dict:
        info = MrhCommMergeInfo()
        self._check(self.lib.mrh_comm_merge_submaps(self._ctx, chunk_log2, C.byref(info)))
        self.params.shard_chunk_log2 = chunk_log2
        return {"sent": int(info.blocks_sent), "received": int(info.blocks_received), "bytes": int(info.bytes_sent), "kept": int(info.blocks_kept)}

    def comm_gather_mesh(self, root: int = 0) -> int:
        n = C.c_uint64()
        self._check(self.lib.mrh_comm_gather_mesh(self._ctx, root, C.byref(n)))
        return int(n.value)

    def comm_phase_times(self) -> dict:
        p = MrhCommPhases()
        self._check(self.lib.mrh_comm_phase_times(self._ctx, C.byref(p)))
        return {k: (float(getattr(p, k)) if k.endswith("_ms") or k.endswith("_sum") else int(getattr(p, k))) for k, _ in MrhCommPhases._fields_}

    # -- multi-GPU block exchange ----------------------------------------------------------------
    def set_sharding(self, rank: int, count: int, chunk_log2: int = 0):
        self._check(self.lib.mrh_set_sharding(self._ctx, rank, count, chunk_log2))
        self.params.shard_rank, self.params.shard_count, self.params.shard_chunk_log2 = rank, count, chunk_log2

    def pack_blocks(self, mode: int, rank_arg: int = 0) -> Tuple[int, int, bool]:
        """(pointer to mrh_block_record[n], n, is_device_memory); the buffer belongs to the context until the next pack."""
        ptr, n, dev = C.c_void_p(), C.c_uint64(), C.c_int()
        self._check(self.lib.mrh_pack_blocks(self._ctx, mode, rank_arg, C.byref(ptr), C.byref(n), C.byref(dev)))
        return int(ptr.value or 0), int(n.value), bool(dev.value)

    def unpack_blocks(self, mode: int, ptr: int, n: int, is_device: bool) -> int:
        taken = C.c_uint64()
        self._check(self.lib.mrh_unpack_blocks(self._ctx, mode, ptr, n, 1 if is_device else 0, C.byref(taken)))
        return int(taken.value)

    def drop_blocks(self, mode: int) -> int:
        n = C.c_uint64()
        self._check(self.lib.mrh_drop_blocks(self._ctx, mode, C.byref(n)))
        return int(n.value)

    def stream_out(self, center, radius: float) -> Tuple[np.ndarray, np.ndarray]:
        """Streamer device half: blocks at distance >= radius from `center` (radius < 0: all) are copied out, ordered by
        block position, and removed from the device map.  Returns (descs, voxels[n, 512])."""
        cen = (C.c_float * 3)(*[float(v) for v in center])
        n = C.c_uint64()
        self._check(self.lib.mrh_stream_out(self._ctx, cen, radius, None, None, 0, C.byref(n)))
        descs = np.zeros(n.value, dtype=DESC_DTYPE)
        vox = np.zeros((n.value, 512), dtype=VOXEL_DTYPE)
        if n.value:
            self._check(self.lib.mrh_stream_out(self._ctx, cen, radius, descs.ctypes.data, vox.ctypes.data, n.value, C.byref(n)))
        return descs, vox

    def dump_blocks(self) -> Tuple[np.ndarray, np.ndarray]:
        """Canonical dump: (descs sorted by (x,y,z), voxels[n,512]) in the reference Voxel layout."""
        n = C.c_uint64()
        self._check(self.lib.mrh_dump_blocks(self._ctx, None, None, 0, C.byref(n)))
        cap = max(int(n.value), 1)
        descs = np.zeros(cap, dtype=DESC_DTYPE)
        voxels = np.zeros((cap, 512), dtype=VOXEL_DTYPE)
        self._check(self.lib.mrh_dump_blocks(self._ctx, descs.ctypes.data, voxels.ctypes.data, cap, C.byref(n)))
        descs, voxels = descs[: n.value], voxels[: n.value]
        order = np.lexsort((descs["z"], descs["y"], descs["x"]))
        return descs[order], voxels[order]

    def import_blocks(self, descs: np.ndarray, voxels: np.ndarray):
        """Inverse of dump_blocks (restore a map, or bring halo blocks of other shards in)."""
        d = np.ascontiguousarray(descs, dtype=DESC_DTYPE)
        v = np.ascontiguousarray(voxels, dtype=VOXEL_DTYPE).reshape(len(d), 512)
        self._check(self.lib.mrh_import_blocks(self._ctx, d.ctypes.data, v.ctypes.data, len(d)))

    def triangle_blocks(self) -> Tuple[np.ndarray, np.ndarray]:
        """(descs, per-block triangle counts) of the last extract_triangles(), canonical block order."""
        pd, pc, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self.lib.mrh_get_triangle_blocks(self._ctx, C.byref(pd), C.byref(pc), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, DESC_DTYPE), np.zeros(0, np.uint32)
        d = np.frombuffer((C.c_char * (n.value * 16)).from_address(pd.value), dtype=DESC_DTYPE).copy()
        c = np.frombuffer((C.c_char * (n.value * 4)).from_address(pc.value), dtype=np.uint32).copy()
        return d, c

    def triangles_device(self) -> Tuple[int, int, bool]:
        """(pointer, number of triangles, is_device_memory) of the soup of the last extraction / run merge, where the library keeps it."""
        ptr, n, dev = C.c_void_p(), C.c_uint64(), C.c_int()
        self._check(self.lib.mrh_get_triangles_device(self._ctx, C.byref(ptr), C.byref(n), C.byref(dev)))
        return int(ptr.value or 0), int(n.value), bool(dev.value)

    def process_triangle_runs(self, descs: np.ndarray, counts: np.ndarray, tris_ptr: int, n_tris: int, is_device: bool):
        """Per-block triangle runs of all ranks (descs / counts: host arrays; triangles: one buffer, device or host) ->
        canonical order + mesh post-process (rank 0 of a sharded extraction)."""
        d = np.ascontiguousarray(descs, dtype=DESC_DTYPE)
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        self._check(self.lib.mrh_process_triangle_runs(self._ctx, d.ctypes.data, c.ctypes.data, len(d), tris_ptr, n_tris, 1 if is_device else 0))

    def process_triangles(self, tris: np.ndarray):
        t = np.ascontiguousarray(tris, dtype=TRI_DTYPE).reshape(-1, 3)
        self._check(self.lib.mrh_process_triangles(self._ctx, t.ctypes.data, len(t)))

    def selftest_division(self, samples: int = 1 << 26, seed: int = 1) -> int:
        n = C.c_uint64()
        self._check(self.lib.mrh_selftest_division(self._ctx, samples, seed, C.byref(n)))
        return int(n.value)

    def get_voxel(self, vx: int, vy: int, vz: int):
        out = np.zeros(1, dtype=VOXEL_DTYPE)
        found = C.c_int()
        self._check(self.lib.mrh_get_voxel(self._ctx, vx, vy, vz, out.ctypes.data, C.byref(found)))
        return out[0], bool(found.value)
