"""Drop-in for the reference's nanobind module `mrhash.src.pygeowrapper`."""
from mrhash_amd.pygeowrapper import GeoWrapper  # noqa: F401
