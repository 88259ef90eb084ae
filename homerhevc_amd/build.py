"""Build recipe for the native library (hipcc, gfx950 only, in-tree so it travels to the GPU box)."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libhomer_gpu.so")
SOURCES = ["tables.cpp", "context.cpp", "dropin.cpp", "cmdlist.cpp", "k_pixel.hip", "k_transform.hip", "k_intra.hip", "k_interp.hip", "k_loop.hip", "k_motion.hip", "k_tuchain.hip", "k_intrasearch.hip", "k_tree.hip", "k_chromasearch.hip"]
# -ffp-contract=off: the few double-precision cost terms must round exactly like the reference's x87-free SSE2 code
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip"]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG_DIR, "..", "include", "homer_gpu.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    """Compile every HIP source into homerhevc_amd/libhomer_gpu.so (cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-shared", "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH
