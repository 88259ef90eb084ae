"""Locate / build / load the three shared libraries the tests talk to (ctypes)."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libhomer_ref.so")
REF_LOCKSTEP = os.path.join(ORACLE_DIR, "_ref", "ref_lockstep")
GPU_SO = os.path.join(ROOT, "homerhevc_amd", "libhomer_gpu.so")

_cache = {}


def load_oracle():
    """CPU oracle (test infrastructure).  Built on demand: plain C, gcc only."""
    if "ora" not in _cache:
        src = os.path.join(ORACLE_DIR, "hmr_oracle.c")
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, ORACLE_SO])
        _cache["ora"] = ctypes.CDLL(ORACLE_SO)
    return _cache["ora"]


def load_ref():
    """Compiled reference + harness, or None when absent (it is never built on the GPU box)."""
    if "ref" not in _cache:
        lib = None
        if os.path.exists(REF_SO):
            lib = ctypes.CDLL(REF_SO)
            if lib.refh_open(192, 128) != 0:
                lib = None
        _cache["ref"] = lib
    return _cache["ref"]


def load_gpu():
    """The product C-ABI library.  Raises if it has not been built (never falls back to the oracle)."""
    if "gpu" not in _cache:
        if not os.path.exists(GPU_SO):
            raise RuntimeError("homerhevc_amd/libhomer_gpu.so missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        # a process that also uses torch on the GPU (tests/test_gpu_engines.py, bench.py) has to let torch bring up its HIP runtime FIRST: torch ships its own
        # libamdhip64, and once this library has initialised the system one torch finds "No HIP GPUs" (INTEGRATION.md "torch in the same process")
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
        _cache["gpu"] = ctypes.CDLL(GPU_SO)
    return _cache["gpu"]
