"""svinet_amd -- MI355X-native implementation of svinet's `-link-sampling` path.

Layout:
  csrc/        hand-written HIP kernels for gfx950 + the C ABI (include/svils.h)
  host/        C++ host side (CLI, Env, Network, RNG, driver, writers)
  _svils.py    ctypes binding of the C ABI
  build.py     in-tree hipcc / g++ build
"""
__version__ = "0.1.0"
