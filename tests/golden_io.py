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


def load_frame_goldens():
    """-> list of (case dict for frame_cases.run_*, expected outputs dict) rebuilt from the fixture + seeds."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    import frame_cases as fc
    import gen_yuv
    with open(os.path.join(GOLDEN, "frames_200x136.json")) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN, "frames_200x136.npz"))
    W, H = meta["width"], meta["height"]
    frames = list(gen_yuv.gen_frames(W, H, 3))
    out = []
    for c in meta["cases"]:
        fi, tag = c["frame"], c["tag"]
        y, u, v = frames[fi]
        info = {k: data[f"f{fi}_info_{k}"] for k in fc.INFO_NAMES}
        case = {"width": W, "height": H, "info": info, "pre": fc.blocky_planes(y, u, v, c["pre_seed"]),
                "orig": [p.astype(np.int16) for p in (y, u, v)], "sao_params": data[f"{tag}_sao_params"], "dbk": c["dbk"]}
        exp = {"bs_ver": data[f"{tag}_bs_ver"], "bs_hor": data[f"{tag}_bs_hor"], "stats": data[f"{tag}_stats"],
               "deblocked": [data[f"{tag}_deblocked_{n}"].astype(np.int16) for n in "yuv"],
               "sao": [data[f"{tag}_sao_{n}"].astype(np.int16) for n in "yuv"],
               "padded": [data[f"{tag}_padded_{n}"].astype(np.int16) for n in "yuv"]}
        out.append((case, exp, c))
    return out
