"""Process-level HIP runtime ordering, for processes that ALSO hold PyTorch (the test-suite's gloo rendezvous, a caller's
own tensors).  The product path — libmrhash_hip.so, mrhash_amd.capi / hipmem / parallel's RCCL communicator, bench.py — never
imports torch and holds one HIP runtime (/opt/rocm/lib/libamdhip64.so.7, and the librccl.so.1 next to it).

PyTorch-ROCm wheels carry their own copy of the HIP runtime under the same soname.  When torch is already loaded, this
library's DT_NEEDED libamdhip64.so.7 resolves to torch's copy (one runtime, torch's); torch only finds its GPUs when that copy
initialises before this library touches the device (measured on the GPU box: the other order ends in torch's "No HIP GPUs
are available").  So: if torch has been imported by the time the library is loaded, let it initialise first.  A process that
wants both must import torch BEFORE capi.load_hip().  MRHASH_NO_TORCH_PRELOAD=1 skips this."""
import os
import sys

_done = False


def torch_first() -> None:
    global _done
    if _done or os.environ.get("MRHASH_NO_TORCH_PRELOAD"):
        return
    _done = True
    torch = sys.modules.get("torch")
    if torch is None:  # the product path: no torch in the process, nothing to order
        return
    try:
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
