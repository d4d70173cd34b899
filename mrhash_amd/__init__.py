"""mrhash_amd - MI355X-native voxel-hash TSDF fusion (drop-in for rvp-group/mrhash's GeoWrapper path).

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C ABI (include/mrhash_hip.h) + the C++ GeoWrapper host
  capi.py    ctypes binding of the C ABI (used by tests and bench.py)
  synth.py   seeded synthetic RGB-D streams standing in for Replica / ScanNet
  build.py   in-tree build driver (hipcc --offload-arch=gfx950)
"""
__version__ = "0.1.0"
