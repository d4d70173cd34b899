"""Device buffers for the Python drivers (bench.py, tests, examples) without a second framework in the process: plain
hipMalloc / hipMemcpy through ctypes on the HIP runtime libmrhash_hip.so is already bound to (the soname resolves to the
loaded copy), so a process that only fuses frames holds ONE HIP runtime and no torch."""
from __future__ import annotations

import ctypes as C

import numpy as np

_hip = None

H2D, D2H, D2D = 1, 2, 3


def runtime() -> C.CDLL:
    global _hip
    if _hip is None:
        from . import capi

        capi.load_hip()  # binds libamdhip64.so.7; the name below then resolves to that very copy
        try:
            _hip = C.CDLL("libamdhip64.so.7")
        except OSError:
            _hip = C.CDLL("/opt/rocm/lib/libamdhip64.so.7")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipSetDevice.argtypes = [C.c_int]
        _hip.hipGetDeviceCount.argtypes = [C.POINTER(C.c_int)]
        _hip.hipDeviceSynchronize.argtypes = []
        _hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _hip.hipGetErrorString.argtypes = [C.c_int]
        _hip.hipGetErrorString.restype = C.c_char_p
    return _hip


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {runtime().hipGetErrorString(rc).decode()}")


def device_count() -> int:
    n = C.c_int(0)
    rc = runtime().hipGetDeviceCount(C.byref(n))
    return int(n.value) if rc == 0 else 0


def set_device(index: int):
    _check(runtime().hipSetDevice(index), "hipSetDevice")


def synchronize():
    _check(runtime().hipDeviceSynchronize(), "hipDeviceSynchronize")


def mem_get_info():
    """(free, total) bytes of the current device."""
    f, t = C.c_size_t(), C.c_size_t()
    _check(runtime().hipMemGetInfo(C.byref(f), C.byref(t)), "hipMemGetInfo")
    return int(f.value), int(t.value)


class DeviceBuffer:
    """`nbytes` of device memory on the current device; freed with the object."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(runtime().hipMalloc(C.byref(p), max(self.nbytes, 1)), "hipMalloc")
        self.ptr = int(p.value)

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        if a.nbytes:
            _check(runtime().hipMemcpy(b.ptr, a.ctypes.data, a.nbytes, H2D), "hipMemcpy H2D")
        return b

    def to_numpy(self, dtype, count: int = -1, offset_bytes: int = 0) -> np.ndarray:
        dt = np.dtype(dtype)
        n = (self.nbytes - offset_bytes) // dt.itemsize if count < 0 else count
        out = np.empty(n, dtype=dt)
        if out.nbytes:
            _check(runtime().hipMemcpy(out.ctypes.data, self.ptr + offset_bytes, out.nbytes, D2H), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            runtime().hipFree(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def h2d_link_rate(sizes=(640 * 480 * 4, 640 * 480 * 3), frames: int = 300) -> dict:
    """What the host -> device link gives a frame's images at best, measured with nothing else on the device: every frame's
    buffers (`sizes`: the depth and the colour image of a 640x480 frame, 2.15 MB together) go from pinned host memory to device
    memory with hipMemcpyAsync, one stream per buffer (as mrh_upload_* sends them), `frames` frames back to back, one
    synchronisation at the end (`gbs`); and with a synchronisation after every frame (`gbs_frame_sync`).  The ceiling of any
    path that hands host images over per frame (bench.py: pcie_inclusive_frac_of_link)."""
    import time

    hip = runtime()
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipHostFree.argtypes = [C.c_void_p]
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hosts, devs, streams = [], [], []
    for n in sizes:
        h, st = C.c_void_p(), C.c_void_p()
        _check(hip.hipHostMalloc(C.byref(h), n, 0), "hipHostMalloc")
        C.memset(h, 1, n)
        _check(hip.hipStreamCreateWithFlags(C.byref(st), 1), "hipStreamCreateWithFlags")  # hipStreamNonBlocking
        hosts.append(h); devs.append(DeviceBuffer(n)); streams.append(st)

    def loop(sync_each: bool) -> float:
        t0 = time.perf_counter()
        for _ in range(frames):
            for h, d, st, n in zip(hosts, devs, streams, sizes):
                hip.hipMemcpyAsync(d.ptr, h, n, H2D, st)
            if sync_each:
                for st in streams:
                    hip.hipStreamSynchronize(st)
        for st in streams:
            hip.hipStreamSynchronize(st)
        return time.perf_counter() - t0

    loop(False)  # warm
    t_pipe = min(loop(False) for _ in range(3))
    t_sync = min(loop(True) for _ in range(2))
    for h, d, st in zip(hosts, devs, streams):
        hip.hipStreamDestroy(st); hip.hipHostFree(h); d.free()
    total = float(sum(sizes)) * frames
    return {"bytes_per_frame": int(sum(sizes)), "frames": frames, "gbs": total / t_pipe / 1e9, "us_per_frame": t_pipe / frames * 1e6,
            "gbs_frame_sync": total / t_sync / 1e9}


def read(ptr: int, nbytes: int) -> bytes:
    """`nbytes` at device pointer `ptr` as host bytes."""
    out = np.empty(nbytes, np.uint8)
    if nbytes:
        _check(runtime().hipMemcpy(out.ctypes.data, ptr, nbytes, D2H), "hipMemcpy D2H")
    return out.tobytes()


def write(ptr: int, a: np.ndarray):
    """Host array -> device memory at `ptr`."""
    a = np.ascontiguousarray(a)
    if a.nbytes:
        _check(runtime().hipMemcpy(ptr, a.ctypes.data, a.nbytes, H2D), "hipMemcpy H2D")
