#!/usr/bin/env python3
"""Mint golden vectors from the COMPILED REFERENCE (build container only).

Runs every case of kernel_cases.all_cases("golden") through oracle/_ref/libhomer_ref.so
(reference sources compiled where they lie with the "oracle B" flags, see oracle/Makefile) and
stores params + seed + the reference's outputs in tests/golden/table_kernels.npz.  Inputs are not
stored: kernel_cases.py rebuilds them from the seed.  Re-run after changing kernel_cases.py:
    python tests/golden/make_golden.py
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import kernel_cases as kc  # noqa: E402
import libs  # noqa: E402


def main():
    ref = libs.load_ref()
    if ref is None:
        sys.exit("oracle/_ref/libhomer_ref.so missing: run `make -C oracle ref` in the build container")
    cases = kc.all_cases("golden")
    arrays, index = {}, []
    for i, case in enumerate(cases):
        out = kc.run(ref, "refh_", case)
        index.append({"kernel": case[0], "params": case[1], "seed": case[2], "outputs": sorted(out)})
        for k, v in out.items():
            arrays[f"c{i}_{k}"] = v
    meta = {
        "generator": "tests/golden/make_golden.py",
        "reference_flags": "gcc -O3 -fno-aggressive-loop-optimizations -msse4.2 -mssse3 (oracle B, SURVEY.md §0-11)",
        "gcc": subprocess.check_output(["gcc", "--version"], text=True).splitlines()[0],
        "numpy": np.__version__,
        "cases": index,
    }
    np.savez_compressed(os.path.join(HERE, "table_kernels.npz"), **arrays)
    with open(os.path.join(HERE, "table_kernels.json"), "w") as f:
        json.dump(meta, f, indent=0, separators=(",", ":"))
    print(len(cases), "cases written")
    # motion search driver / motion compensation (tests/motion_cases.py)
    import motion_cases as mc
    arrays, index = {}, []
    for i, case in enumerate(mc.all_cases("golden")):
        out = mc.run(ref, "refh_", case)
        index.append({"kernel": case[0], "params": case[1], "seed": case[2], "outputs": sorted(out)})
        for k, v in out.items():
            arrays[f"c{i}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "motion.npz"), **arrays)
    with open(os.path.join(HERE, "motion.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "reference_flags": meta["reference_flags"], "cases": index}, f, indent=0, separators=(",", ":"))
    print(len(index), "motion cases written")


if __name__ == "__main__":
    main()
