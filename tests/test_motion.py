"""Motion search driver + motion compensation: oracle vs compiled reference (build container), oracle vs goldens (anywhere),
GPU (through the C ABI) vs oracle and goldens (-m gpu)."""
import json
import os

import numpy as np
import pytest

import golden_io
import libs
import motion_cases as mc


def _golden():
    with open(os.path.join(golden_io.GOLDEN, "motion.json")) as f:
        meta = json.load(f)
    data = np.load(os.path.join(golden_io.GOLDEN, "motion.npz"))
    out = []
    for i, c in enumerate(meta["cases"]):
        p = c["params"]
        for k in ("shift", "init"):
            if k in p:
                p[k] = tuple(p[k])
        out.append(((c["kernel"], p, c["seed"]), {k: data[f"c{i}_{k}"] for k in c["outputs"]}))
    return out


def _check(lib, prefix, cases_with_exp):
    n_sub = 0
    for case, exp in cases_with_exp:
        got = mc.run(lib, prefix, case)
        for k, v in exp.items():
            assert np.array_equal(got[k], v), f"{case}: {k} got {got[k].ravel()[:8]} expected {v.ravel()[:8]}"
        if case[0] == "motion_estimation" and (got["mv"][2] or got["mv"][3]):
            n_sub += 1
    return n_sub


def test_oracle_matches_reference(oracle, ref):
    cases = mc.all_cases("full")
    n_sub = _check(oracle, "ora_", [(c, mc.run(ref, "refh_", c)) for c in cases])
    assert n_sub > 50, "sub-pel refinement hardly exercised"


def test_oracle_matches_goldens(oracle):
    g = _golden()
    assert len(g) >= 200
    _check(oracle, "ora_", g)


@pytest.mark.gpu
def test_gpu_matches_oracle_and_goldens(oracle):
    gpu = libs.load_gpu()
    cases = mc.all_cases("full")
    n_sub = _check(gpu, "hmr_gpu_", [(c, mc.run(oracle, "ora_", c)) for c in cases])
    assert n_sub > 50
    _check(gpu, "hmr_gpu_", _golden())
