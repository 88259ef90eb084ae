#!/usr/bin/env python3
"""Mint the per-CTU fixtures of the frame encoder from the COMPILED REFERENCE (build container only).

oracle/_ref/ref_ctudump encodes the synthetic clip of tools/gen_yuv.py in lockstep (wpp = 1, engines = 1: the only
deterministic mode) and dumps, after every CTU's decisions, its side-info arrays, levels and pre-filter
reconstruction; the lockstep driver also writes the final reconstructed pictures, which the tests feed back as reference
pictures so that every frame is compared on the reference's own inputs.
  ctus_200x136.npz : full records, 3 frames (I P P)
  ctus_416x240.npz : per-CTU per-field CRC32, 4 frames, + the reconstructed pictures
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import encoder_cases as ec  # noqa: E402
import gen_yuv  # noqa: E402


def run(width, height, frames, full, keys=()):
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "in.yuv")
        gen_yuv.write_clip(yuv, width, height, frames)
        env = dict(os.environ, HOMER_CTUDUMP=os.path.join(tmp, "ctus.bin"))
        cmd = [os.path.join(ROOT, "oracle", "_ref", "ref_ctudump"), yuv, os.path.join(tmp, "out.265"), str(width), str(height), str(frames),
               "recon=" + os.path.join(tmp, "rec.yuv")] + list(keys)
        subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL)
        dump = open(os.path.join(tmp, "ctus.bin"), "rb").read()
        rec = open(os.path.join(tmp, "rec.yuv"), "rb").read()
        stream = open(os.path.join(tmp, "out.265"), "rb").read()
    nx = (width + 63) // 64
    nctu = nx * ((height + 63) // 64)
    out = {"width": np.int32(width), "height": np.int32(height), "frames": np.int32(frames), "stream": np.frombuffer(stream, dtype=np.uint8)}
    fsz = width * height * 3 // 2
    for f in range(frames):
        r = ec.crop_recon(dump[f * nctu * ec.REC:(f + 1) * nctu * ec.REC], width, height, nx)
        if full:
            out[f"f{f}_records"] = np.frombuffer(r, dtype=np.uint8)
        else:
            out[f"f{f}_hashes"] = np.stack([ec.field_hashes(r[n * ec.REC:(n + 1) * ec.REC]) for n in range(nctu)])
        out[f"f{f}_recon"] = np.frombuffer(rec[f * fsz:(f + 1) * fsz], dtype=np.uint8)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "ctus_200x136.npz"), **run(200, 136, 3, True))
    np.savez_compressed(os.path.join(HERE, "ctus_416x240.npz"), **run(416, 240, 4, False))
    for n in ("ctus_200x136.npz", "ctus_416x240.npz"):
        print(n, os.path.getsize(os.path.join(HERE, n)))
