"""homerhevc_amd - MI355X (gfx950) backend for HomerHEVC's per-block encode hot path.

The product is `libhomer_gpu.so` (hand-written HIP kernels behind the C ABI of include/homer_gpu.h);
this package is the thin Python host side: the build recipe (build.py), the engine-per-GPU ring bench.py --gpus N runs
(engines.py), a ctypes mirror of the reference's `low_level_funcs_t` table (lowlevel.py) and the batched / frame-level
kernel driver the round-1 kernel tests and tools/bench_callmix.py use (gpu.py).
There is no CPU fallback: importing works anywhere, calling a kernel without the native library or
without a GPU raises.
"""
from .build import LIB_PATH, build_native  # noqa: F401

__all__ = ["LIB_PATH", "build_native"]
