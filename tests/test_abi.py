"""CPU-side checks of the drop-in boundary: the product library loads, exports every symbol the header
declares, and fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os
import re

import pytest

from mrhash_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mrhash_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mrh_[a-z_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(capi.ABI_SYMBOLS)


def test_hip_library_exports_every_declared_symbol(hip):
    for name in _declared_symbols():
        assert hasattr(hip, name), f"libmrhash_hip.so does not export {name}"
    assert hip.mrh_version().startswith(b"mrhash_hip")


def test_comm_header_and_binding_agree_and_the_library_exports_them(hip):
    """include/mrhash_comm.h (RCCL behind the C ABI): every declared symbol is bound by capi and exported by the product library.
    Nothing is called: RCCL is only opened by the first communicator call."""
    hdr = open(os.path.join(ROOT, "include", "mrhash_comm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mrh_comm_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(capi.COMM_SYMBOLS)
    for name in declared:
        assert hasattr(hip, name), f"libmrhash_hip.so does not export {name}"
    assert C.sizeof(capi.MrhCommPhases) == 40 and C.sizeof(capi.MrhCommMergeInfo) == 32  # = sizeof in C (gcc probe)


def test_oracle_exports_the_same_abi(oracle):
    for name in _declared_symbols():
        assert hasattr(oracle, name)
    assert oracle.mrh_version().startswith(b"mrh_oracle")


def test_param_struct_sizes_match_header():
    # mrh_params: 11 x 4-byte + 4 x 1-byte + 4 floats ... checked against the C layout by compiling a probe
    import subprocess, tempfile, textwrap

    src = textwrap.dedent(
        """
        #include <stdio.h>
        #include <stddef.h>
        #include "mrhash_hip.h"
        int main(void) {
          printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mrh_params), offsetof(mrh_params, num_sdf_blocks), offsetof(mrh_params, shard_count),
                 sizeof(mrh_stats), offsetof(mrh_stats, error_flags), sizeof(mrh_voxel), sizeof(mrh_triangle), sizeof(mrh_splat_seed),
                 offsetof(mrh_splat_seed, rgb), sizeof(mrh_qtree_leaf), sizeof(mrh_block_desc));
          return 0;
        }"""
    )
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "probe.c")
        open(p, "w").write(src)
        exe = os.path.join(d, "probe")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), p, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    got = [int(x) for x in out]
    want = [C.sizeof(capi.MrhParams), capi.MrhParams.num_sdf_blocks.offset, capi.MrhParams.shard_count.offset,
            C.sizeof(capi.MrhStats), capi.MrhStats.error_flags.offset, 12, 72, capi.SEED_DTYPE.itemsize, capi.SEED_DTYPE.fields["rgb"][1],
            capi.LEAF_DTYPE.itemsize, capi.DESC_DTYPE.itemsize]
    assert got == want


def test_create_rejects_bad_arguments(hip):
    p = capi.Params(sdf_truncation=0.06, virtual_voxel_size=0.02).to_c()
    ctx = C.c_void_p()
    p.abi_version = 99
    assert hip.mrh_create(C.byref(p), C.byref(ctx)) == capi.MRH_ERR_INVALID_ARG
    assert b"abi_version" in hip.mrh_last_error(None)
    assert hip.mrh_create(None, C.byref(ctx)) == capi.MRH_ERR_INVALID_ARG


def test_product_path_fails_loudly_without_a_device(hip):
    """On a box without a GPU the product library must refuse to create a context - it must not
    silently compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present; the no-device path cannot be exercised here")
    with pytest.raises(capi.MrhError) as ei:
        capi.Engine(hip, capi.Params(sdf_truncation=0.06, virtual_voxel_size=0.02, num_sdf_blocks=1024))
    assert ei.value.code == capi.MRH_ERR_NO_DEVICE
