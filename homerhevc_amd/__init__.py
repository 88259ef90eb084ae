"""homerhevc_amd - MI355X (gfx950) backend for HomerHEVC's per-block encode hot path.

The product is `libhomer_gpu.so` (hand-written HIP kernels behind the C ABI of include/homer_gpu.h);
this package is the thin Python host side: the build recipe (build.py) and the engine-per-GPU ring bench.py --gpus N runs
(engines.py).  (The ctypes mirror of the batched kernel ABI that the kernel tests use is test infrastructure: tests/gpu_abi.py.)
There is no CPU fallback: importing works anywhere, calling a kernel without the native library or
without a GPU raises.
"""
from .build import LIB_PATH, build_native  # noqa: F401

__all__ = ["LIB_PATH", "build_native"]
