"""Read the committed golden fixtures (tests/golden/*.npz + *.json)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_table_kernel_goldens():
    with open(os.path.join(GOLDEN, "table_kernels.json")) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN, "table_kernels.npz"))
    out = []
    for i, c in enumerate(meta["cases"]):
        case = (c["kernel"], c["params"], c["seed"])
        out.append((case, {k: data[f"c{i}_{k}"] for k in c["outputs"]}))
    return out
