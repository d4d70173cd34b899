"""In-tree build of the native pieces (no JIT cache: the built .so files travel with the repo snapshot).

  libmrhash_hip.so   hand-written gfx950 HIP kernels + C ABI     (hipcc --offload-arch=gfx950)
  pygeowrapper*.so   C++ GeoWrapper host + pybind11 binding       (g++, loads libmrhash_hip.so via the C ABI)

Arithmetic-spec flags (must match oracle/Makefile): -ffp-contract=off, correctly rounded fp32 divide/sqrt,
denormals preserved.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mrhash_amd", "csrc")
HIP_LIB = os.path.join(CSRC, "libmrhash_hip.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-ldl",
]


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_hip(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "mrh_*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    if not force and _newer(HIP_LIB, srcs):
        return HIP_LIB
    cmd = [hipcc()] + [f for f in HIPCC_FLAGS if f] + ["-o", HIP_LIB, os.path.join(CSRC, "mrh_capi.hip")]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return HIP_LIB


def build_variant(name: str, extra_flags, verbose: bool = True) -> str:
    """A tuning build of the same sources under another name (libmrhash_<name>.so, e.g. with -DMRH_BACK_WAVES=6), for A/B runs
    on one box through tools/bench_with_lib.py / tools/abn.sh.  Never loaded by the product path."""
    out = os.path.join(CSRC, f"libmrhash_{name}.so")
    cmd = [hipcc()] + HIPCC_FLAGS + list(extra_flags) + ["-o", out, os.path.join(CSRC, "mrh_capi.hip")]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


def build_oracle(verbose: bool = True) -> str:
    """Test infrastructure only (see oracle/mrh_oracle.c header)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=None if verbose else subprocess.DEVNULL)
    return os.path.join(ROOT, "oracle", "_build", "libmrh_oracle.so")


def build_oracle_ref(verbose: bool = True) -> str:
    """oracle/_ref: the slice of the reference that builds from its own sources where they lie (oracle/Makefile `ref`):
    nothing happens when /root/reference is absent (the GPU box uses the prebuilt files).  Test infrastructure only."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, stdout=None if verbose else subprocess.DEVNULL)
    return os.path.join(ROOT, "oracle", "_ref")


def build_pybind(force: bool = False, verbose: bool = True) -> str:
    import pybind11

    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(ROOT, "mrhash_amd", "pygeowrapper" + ext)
    srcs = [os.path.join(CSRC, f) for f in ("geowrapper.cpp", "geowrapper.h", "pygeowrapper.cpp")]
    srcs.append(os.path.join(ROOT, "include", "mrhash_hip.h"))
    srcs.append(HIP_LIB)
    if not all(os.path.exists(s) for s in srcs):
        return ""
    if not force and _newer(out, srcs):
        return out
    cmd = [
        "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-fvisibility=hidden",
        "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"], "-I", os.path.join(ROOT, "include"),
        os.path.join(CSRC, "geowrapper.cpp"), os.path.join(CSRC, "pygeowrapper.cpp"),
        "-o", out, "-L", CSRC, "-lmrhash_hip", "-Wl,-rpath,$ORIGIN/csrc",
    ]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


def build_all(force: bool = False):
    build_hip(force)
    build_pybind(force)
    build_oracle()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "variant":  # python -m mrhash_amd.build variant <name> [-D...]
        build_variant(sys.argv[2], sys.argv[3:])
    else:
        build_all(force="--force" in sys.argv)
