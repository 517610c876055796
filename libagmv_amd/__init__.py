"""libagmv_amd -- MI355X-native hot path of the AGMV codec (libagmv drop-in for that path).

Layout
  csrc/agmv_hip.hip   hand-written gfx950 kernels + the C-ABI of include/agmv_hip.h
  csrc/*.c            host C: libagmv-compatible API (include/agmv.h), LZSS/LZ77, container,
                      BMP I/O, palette build, synthetic clip generator
  hip.py              ctypes binding of the C-ABI for tests / bench (torch = device memory only)
  build.py            in-tree build of libagmv_hip.so / libagmv.so (hipcc, gcc)

There is no CPU fallback anywhere in this package: without the built HIP library, or
without a GPU, the hot-path calls raise.
"""
from .hip import AgmvHip, HipUnavailable, lib_path, load_library  # noqa: F401

__all__ = ["AgmvHip", "HipUnavailable", "lib_path", "load_library"]
