"""The block primitives the frame encoder's CTU walk runs (homerhevc_amd/csrc/enc/enc_prims.h) against the CPU oracle and the reference-minted golden vectors, one call at a
time through hmr_gpu_prim_* (include/homer_gpu.h section 15): the same case sweep that holds the table kernels (test_gpu_parity.py), so that the walk's own
implementation of every primitive is pinned directly and not only through whole streams.  Each kernel runs in the 16-bit instantiation (what the one-lane checker build
compiles) and - where its sample operands are the worker's byte windows and the case's samples are 0 .. 255 - in the byte instantiation the device runs."""
import numpy as np
import pytest

import golden_io
import kernel_cases as kc
import libs

pytestmark = pytest.mark.gpu

PRIMS = ["sad", "ssd16b", "predict", "reconst", "modified_variance", "intra_planar", "intra_angular", "fill_reference_samples", "adi_filter", "transform", "itransform",
         "quant", "inv_quant"]
BYTE_PRIMS = {"sad", "ssd16b", "predict", "reconst", "modified_variance", "intra_planar", "intra_angular", "transform"}     # source / prediction operands live in byte windows on the device
CASES = [c for c in kc.all_cases("full") if c[0] in PRIMS]


def byte_case(case):
    kernel, p, _ = case
    if kernel not in BYTE_PRIMS or "src_range" in p or "pred_range" in p:
        return False
    return not (kernel == "sad" and p["n"] < 8)      # (the motion search's byte SAD starts at 8 x 8 blocks)


@pytest.fixture(scope="module")
def gpu():
    return libs.load_gpu()


@pytest.mark.parametrize("kernel", PRIMS)
def test_walk_primitive_matches_oracle(kernel, gpu, oracle):
    n = nb = 0
    for case in CASES:
        if case[0] != kernel:
            continue
        exp = kc.run(oracle, "ora_", case)
        for bytes_mode in ((0, 1) if byte_case(case) else (0,)):
            gpu.hmr_gpu_prim_bytes(bytes_mode)
            try:
                got = kc.run(gpu, "hmr_gpu_prim_", case)
            finally:
                gpu.hmr_gpu_prim_bytes(0)
            for key in exp:
                assert np.array_equal(got[key], exp[key]), f"{case} bytes={bytes_mode}: {key}\ngpu={got[key].ravel()[:16]}\nora={exp[key].ravel()[:16]}"
            nb += bytes_mode
        n += 1
    assert n > 0 and (nb > 0) == (kernel in BYTE_PRIMS)


def test_walk_primitives_match_reference_goldens(gpu):
    n = 0
    for case, exp in golden_io.load_table_kernel_goldens():
        if case[0] not in PRIMS:
            continue
        got = kc.run(gpu, "hmr_gpu_prim_", case)
        for key, val in exp.items():
            assert np.array_equal(got[key], val), f"{case}: {key}"
        n += 1
    assert n > 300
