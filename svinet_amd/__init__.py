"""svinet_amd -- MI355X-native implementation of svinet's `-link-sampling` path.

Layout:
  csrc/        hand-written HIP kernels for gfx950 + the C ABI (include/svils.h)
  host/        C++ host side (CLI, Env, Network, RNG, driver, writers)
  _svils.py    ctypes binding of the C ABI
  host_api.py  ctypes binding of the C++ host side
  sharded.py   node-block multi-GPU driver (torch.distributed / RCCL)
  ksharded.py  K-sharded multi-GPU driver (every rank a column slice of all rows)
  build.py     in-tree hipcc / g++ build
"""
import ctypes as _C
import importlib.util as _ilu
import os as _os

__version__ = "0.1.0"


def _share_hip_runtime_with_torch():
    """libsvils.so links libamdhip64.so.7; PyTorch-ROCm bundles its own copy under the same
    soname (plus its own libhsa-runtime64).  Only one copy per soname gets loaded, and if the
    system copy wins, a later `import torch` reports "No HIP GPUs are available".  When torch is
    installed, map ITS runtime first (without importing torch) so both always share one runtime,
    whatever the import order.  Without torch the system runtime under /opt/rocm is used."""
    try:
        spec = _ilu.find_spec("torch")
        if spec is None or not spec.origin:
            return
        libdir = _os.path.join(_os.path.dirname(spec.origin), "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = _os.path.join(libdir, name)
            if _os.path.exists(path):
                _C.CDLL(path, mode=_C.RTLD_GLOBAL)
    except OSError:
        pass


_share_hip_runtime_with_torch()
