"""Pin the CPU oracle against the compiled reference's SSE4.2 symbols (build container only).

The reference holds no tests or golden vectors of its own, so this differential
sweep against oracle/_ref/libhomer_ref.so ("oracle B" flags) is what pins the
oracle; the committed fixtures in tests/golden/ are minted from the same library.
"""
import ctypes as C

import numpy as np
import pytest

import kernel_cases as kc

CASES = kc.all_cases("full")
KERNEL_NAMES = sorted({c[0] for c in CASES})


@pytest.mark.parametrize("kernel", KERNEL_NAMES)
def test_kernel_matches_reference(kernel, oracle, ref):
    n = 0
    for case in CASES:
        if case[0] != kernel:
            continue
        a = kc.run(oracle, "ora_", case)
        b = kc.run(ref, "refh_", case)
        for key in b:
            assert np.array_equal(a[key], b[key]), f"{case}: {key} differs\noracle={a[key].ravel()[:16]}\nref   ={b[key].ravel()[:16]}"
        n += 1
    assert n > 0


def test_chroma_qp_table_matches_reference(ref):
    tab = (C.c_uint8 * 58).in_dll(ref, "chroma_scale_conversion_table")
    assert list(tab) == kc.CHROMA_QP


def test_tables_match_reference(oracle, ref):
    oracle.ora_scan_table.restype = C.POINTER(C.c_uint32)
    oracle.ora_quant_table.restype = C.POINTER(C.c_int32)
    oracle.ora_dequant_table.restype = C.POINTER(C.c_int32)
    for mode in (1, 2, 3):
        for l in range(1, 6):
            n = (1 << l) ** 2
            got = np.ctypeslib.as_array(oracle.ora_scan_table(mode, l), (n,))
            exp = np.zeros(n, np.uint32)
            ref.refh_get_scan(mode, l - 1, kc.ptr(exp), n)
            assert np.array_equal(got, exp), (mode, l)
    for l in range(2, 6):
        n = (1 << l) ** 2
        for lst in ((0, 1, 3) if l == 5 else range(6)):
            for rem in range(6):
                q, iq = np.zeros(n, np.int32), np.zeros(n, np.int32)
                ref.refh_get_quant(l - 2, lst, rem, kc.ptr(q), kc.ptr(iq), n)
                assert np.array_equal(np.ctypeslib.as_array(oracle.ora_quant_table(l, lst, rem), (n,)), q), (l, lst, rem)
                assert np.array_equal(np.ctypeslib.as_array(oracle.ora_dequant_table(l, lst, rem), (n,)), iq), (l, lst, rem)
