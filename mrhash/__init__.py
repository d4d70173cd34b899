"""Import-path shim: the reference's runners do `from mrhash.src.pygeowrapper import GeoWrapper`
(mrhash/apps/rgbd_runner.py:9).  The implementation lives in mrhash_amd."""
