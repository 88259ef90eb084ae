"""The CPU oracle reproduces the golden vectors minted from the compiled reference (runs anywhere)."""
import collections

import numpy as np
import pytest

import golden_io
import kernel_cases as kc

GOLDENS = golden_io.load_table_kernel_goldens()
BY_KERNEL = collections.defaultdict(list)
for _case, _exp in GOLDENS:
    BY_KERNEL[_case[0]].append((_case, _exp))


@pytest.mark.parametrize("kernel", sorted(BY_KERNEL))
def test_oracle_matches_golden(kernel, oracle):
    for case, exp in BY_KERNEL[kernel]:
        got = kc.run(oracle, "ora_", case)
        for key, val in exp.items():
            assert np.array_equal(got[key], val), f"{case}: {key}"


def test_golden_set_covers_every_table_kernel():
    assert set(BY_KERNEL) == set(kc.KERNELS)
