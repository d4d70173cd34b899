"""Process-level HIP runtime ordering.

PyTorch-ROCm wheels carry their own HIP runtime next to /opt/rocm's, and torch only finds its GPUs when its copy
initialises before this library's does (measured on the GPU box: the other order ends in torch's "No HIP GPUs are
available").  When torch is installed, let it initialise first, so that device tensors and this library can share a
process in either order of use.  MRHASH_NO_TORCH_PRELOAD=1 skips this."""
import os

_done = False


def torch_first() -> None:
    global _done
    if _done or os.environ.get("MRHASH_NO_TORCH_PRELOAD"):
        return
    _done = True
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # torch is optional for the library itself
        pass
