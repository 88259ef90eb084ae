"""GPU parity proper: every table kernel, called through the C ABI (drop-in entries -> batched HIP kernels),
bit-exact against the CPU oracle on seeded inputs and against the golden vectors minted from the reference."""
import collections

import numpy as np
import pytest

import golden_io
import kernel_cases as kc
import libs

pytestmark = pytest.mark.gpu

CASES = kc.all_cases("full")
KERNEL_NAMES = sorted({c[0] for c in CASES})


@pytest.fixture(scope="module")
def gpu():
    return libs.load_gpu()


@pytest.mark.parametrize("kernel", KERNEL_NAMES)
def test_kernel_matches_oracle(kernel, gpu, oracle):
    n = 0
    for case in CASES:
        if case[0] != kernel:
            continue
        got = kc.run(gpu, "hmr_gpu_", case)
        exp = kc.run(oracle, "ora_", case)
        for key in exp:
            assert np.array_equal(got[key], exp[key]), f"{case}: {key}\ngpu={got[key].ravel()[:16]}\nora={exp[key].ravel()[:16]}"
        n += 1
    assert n > 0


def test_gpu_matches_reference_goldens(gpu):
    n = 0
    for case, exp in golden_io.load_table_kernel_goldens():
        got = kc.run(gpu, "hmr_gpu_", case)
        for key, val in exp.items():
            assert np.array_equal(got[key], val), f"{case}: {key}"
        n += 1
    assert n > 500
