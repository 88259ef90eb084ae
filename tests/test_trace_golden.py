"""The CPU oracle reproduces what the reference's block drivers produced during a real encode (tests/golden/trace_200x136.npz)."""
import numpy as np

import trace_cases as tc


def test_oracle_reproduces_the_reference_trace(oracle):
    groups = tc.load()
    planes = tc.ref_planes(groups)
    n = 0
    for g in groups:
        if g["kind"] == "ref_plane":
            continue
        for i in range(g["count"]):
            got, exp = tc.oracle_outputs(oracle, g, i, planes), tc.expected(g, i)
            for k, v in exp.items():
                assert np.array_equal(np.asarray(got[k]), np.asarray(v)), (g["tag"], i, k)
            n += 1
    assert n > 5000 and {g["kind"] for g in groups} == {"inter_tu", "intra_tu", "intra_search", "mc", "me", "ref_plane"}
