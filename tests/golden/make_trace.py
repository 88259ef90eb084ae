#!/usr/bin/env python3
"""Mint tests/golden/trace_200x136.npz: inputs and outputs of the reference's own block drivers during a REAL encode (200x136, I + 2 P frames):
encode_inter_cu / _chroma, encode_intra_cu, homer_loop1_motion_intra, hmr_motion_compensation_luma / _chroma.  The reference runs unmodified
(oracle/_ref/ref_swap in trace mode: the interposers call the reference's functions and log every k-th call); nothing of its source is stored,
only the vectors.  Build container only (needs oracle/_ref)."""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_yuv  # noqa: E402

KINDS = {1: "inter_tu", 2: "intra_tu", 3: "intra_search", 4: "mc", 5: "ref_plane", 6: "me"}
W, H, FRAMES, STRIDE = 200, 136, 3, 4


def parse(path):
    recs = []
    with open(path, "rb") as f:
        data = f.read()
    o = 0
    while o < len(data):
        kind, nh = struct.unpack_from("<ii", data, o); o += 8
        hdr = list(struct.unpack_from(f"<{nh}i", data, o)); o += 4 * nh
        (nd,) = struct.unpack_from("<i", data, o); o += 4
        dbl = list(struct.unpack_from(f"<{nd}d", data, o)); o += 8 * nd
        (nb,) = struct.unpack_from("<i", data, o); o += 4
        blobs = []
        for _ in range(nb):
            (cnt,) = struct.unpack_from("<i", data, o); o += 4
            blobs.append(np.frombuffer(data, np.int16, cnt, o).copy()); o += 2 * cnt
        recs.append((kind, hdr, dbl, blobs))
    return recs


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_swap")
    with tempfile.TemporaryDirectory() as td:
        clip, trace = os.path.join(td, "c.yuv"), os.path.join(td, "t.bin")
        gen_yuv.write_clip(clip, W, H, FRAMES)
        env = dict(os.environ, HOMER_TRACE=trace, HOMER_TRACE_STRIDE=str(STRIDE))
        subprocess.check_call([exe, clip, os.path.join(td, "o.265"), str(W), str(H), str(FRAMES)], env=env, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        recs = parse(trace)
    groups = {}
    for kind, hdr, dbl, blobs in recs:
        # group by (kind, block geometry) so that every group stacks into rectangular arrays
        key = (KINDS[kind], hdr[1], hdr[2]) if kind in (4, 5) else (KINDS[kind], hdr[0], hdr[0])
        groups.setdefault(key, []).append((hdr, dbl, blobs))
    out, meta = {}, {"width": W, "height": H, "frames": FRAMES, "stride": STRIDE, "groups": []}
    for (name, a, b), items in sorted(groups.items()):
        tag = f"{name}_{a}x{b}"
        out[tag + "_hdr"] = np.array([it[0] for it in items], np.int32)
        out[tag + "_dbl"] = np.array([it[1] for it in items], np.float64).reshape(len(items), -1)
        for i in range(len(items[0][2])):
            out[f"{tag}_blob{i}"] = np.stack([it[2][i] for it in items])
        meta["groups"].append({"tag": tag, "kind": name, "w": a, "h": b, "count": len(items)})
    np.savez_compressed(os.path.join(HERE, "trace_200x136.npz"), **out)
    with open(os.path.join(HERE, "trace_200x136.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(sum(g["count"] for g in meta["groups"]), "records in", len(meta["groups"]), "groups,", os.path.getsize(os.path.join(HERE, "trace_200x136.npz")), "bytes")


main()
