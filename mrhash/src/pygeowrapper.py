"""Drop-in for the reference's nanobind module `mrhash.src.pygeowrapper`."""
from mrhash_amd._runtime import torch_first as _torch_first

_torch_first()  # torch's HIP runtime (if torch is installed) has to initialise before this library's: mrhash_amd/_runtime.py
from mrhash_amd.pygeowrapper import GeoWrapper  # noqa: E402,F401
