"""The C-ABI library loads without a GPU and exports every symbol include/homer_gpu.h declares."""
import ctypes
import os
import re

import libs


def declared_symbols():
    text = open(os.path.join(libs.ROOT, "include", "homer_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hmr_gpu_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from homerhevc_amd.build import build_native
    build_native()
    lib = ctypes.CDLL(libs.GPU_SO)
    names = declared_symbols()
    assert len(names) > 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_tables_match_oracle(oracle):
    """Host-side table construction of the product equals the oracle's (which is pinned to the reference)."""
    import numpy as np
    lib = ctypes.CDLL(libs.GPU_SO)
    oracle.ora_scan_table.restype = ctypes.POINTER(ctypes.c_uint32)
    oracle.ora_quant_table.restype = ctypes.POINTER(ctypes.c_int32)
    oracle.ora_dequant_table.restype = ctypes.POINTER(ctypes.c_int32)
    for mode in (1, 2, 3):
        for l in range(1, 6):
            n = (1 << l) ** 2
            got = np.zeros(n, np.uint32)
            assert lib.hmr_gpu_get_scan_table(mode, l, got.ctypes.data_as(ctypes.c_void_p)) == 0
            assert np.array_equal(got, np.ctypeslib.as_array(oracle.ora_scan_table(mode, l), (n,)))
    for l in range(2, 6):
        n = (1 << l) ** 2
        for lst in ((0, 1, 3) if l == 5 else range(6)):
            for rem in range(6):
                q, iq = np.zeros(n, np.int32), np.zeros(n, np.int32)
                assert lib.hmr_gpu_get_quant_table(l, lst, rem, q.ctypes.data_as(ctypes.c_void_p), iq.ctypes.data_as(ctypes.c_void_p)) == 0
                assert np.array_equal(q, np.ctypeslib.as_array(oracle.ora_quant_table(l, lst, rem), (n,)))
                assert np.array_equal(iq, np.ctypeslib.as_array(oracle.ora_dequant_table(l, lst, rem), (n,)))


def test_no_device_is_a_loud_error():
    """Without a GPU the context cannot be created and says so (no silent CPU path)."""
    import torch
    if torch.cuda.is_available():
        return
    lib = ctypes.CDLL(libs.GPU_SO)
    ctx = ctypes.c_void_p()
    rc = lib.hmr_gpu_create(ctypes.byref(ctx), 0, None)
    assert rc != 0
    lib.hmr_gpu_last_error.restype = ctypes.c_char_p
    assert lib.hmr_gpu_last_error()
