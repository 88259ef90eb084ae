#!/usr/bin/env python3
"""Synthetic planar I420 clip generator.

Sinusoid base + seeded +-12 texture, global pan (3,2) px/frame and a moving
240x160 ramp box, so motion estimation, sub-pel refinement and the intra
fallback all have work to do.  `default_rng(1234)` makes clips reproducible;
the md5 of the 1080p x 8 clip is feb867ccc9a69dc281ee193445ab234f.
"""
import argparse
import hashlib
import sys

import numpy as np


def gen_frames(width, height, frames, seed=1234, cut_at=None):
    """Yield (Y, U, V) uint8 planes for `frames` frames.  cut_at = n: from frame n on the content is a different scene (new texture, mirrored and
    re-scaled base) - what the encoder's scene-change detection reacts to; None (the published clips) = no cut."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width]
    # seed 1234 is the published clip; any other seed is a different clip of the same kind: its own texture, pan, base pattern and box path (bench.py encodes
    # several of them side by side so that the sequences of a batch do not all walk the same decisions)
    pan_x, pan_y, f1, f2, f3, box_x, box_y, box_dx, box_dy = 3, 2, 53.0, 41.0, 17.0, 200, 300, 11, 7
    if seed != 1234:
        prm = np.random.default_rng(seed ^ 0x5EED)
        pan_x, pan_y = int(prm.integers(1, 6)), int(prm.integers(0, 4))
        f1, f2, f3 = 53.0 * float(prm.uniform(0.6, 1.5)), 41.0 * float(prm.uniform(0.6, 1.5)), 17.0 * float(prm.uniform(0.7, 1.6))
        box_x, box_y = int(prm.integers(0, max(width - 240, 1))), int(prm.integers(0, max(height - 160, 1)))
        box_dx, box_dy = int(prm.integers(-9, 13)), int(prm.integers(-5, 9))
    base = (128 + 60 * np.sin(xx / f1) * np.cos(yy / f2) + 40 * np.sin((xx + yy) / f3)).astype(np.float32)
    tex = rng.integers(-12, 13, size=(height + 64, width + 64)).astype(np.float32)
    base2 = tex2 = None
    if cut_at is not None:
        base2 = (120 + 70 * np.cos(xx[:, ::-1] / 23.0) * np.sin(yy / 19.0) + 30 * np.sin((2 * xx - yy) / 11.0)).astype(np.float32)
        tex2 = np.random.default_rng(seed + 1).integers(-40, 41, size=(height + 64, width + 64)).astype(np.float32)
    for n in range(frames):
        dx, dy = pan_x * n, pan_y * n
        if cut_at is not None and n >= cut_at:
            base, tex = base2, tex2
        tx, ty = dx % 64, dy % 64      # the texture window wraps after 21 frames (identical to the published definition before that)
        Y = np.roll(np.roll(base, dx, axis=1), dy, axis=0) + tex[ty:ty + height, tx:tx + width]
        bx, by = box_x + box_dx * n, box_y + box_dy * n
        if 0 <= by < height and 0 <= bx < width:
            bh, bw = min(160, height - by), min(240, width - bx)
            Y[by:by + bh, bx:bx + bw] = (200 - 0.2 * np.arange(240))[None, :bw]
        Y = np.clip(Y, 0, 255).astype(np.uint8)
        # chroma is sampled on the even luma grid (x2 = 0, 2, 4, ...)
        U = np.clip(128 + 30 * np.sin((xx[::2, ::2] + dx) / 97.0), 0, 255).astype(np.uint8)
        V = np.clip(128 + 30 * np.cos((yy[::2, ::2] + dy) / 89.0), 0, 255).astype(np.uint8)
        yield Y, U, V


def write_clip(path, width, height, frames, seed=1234, cut_at=None):
    md5 = hashlib.md5()
    with open(path, "wb") as f:
        for planes in gen_frames(width, height, frames, seed, cut_at):
            for p in planes:
                b = p.tobytes()
                md5.update(b)
                f.write(b)
    return md5.hexdigest()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("out")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args(argv)
    print(write_clip(a.out, a.width, a.height, a.frames, a.seed))


if __name__ == "__main__":
    sys.exit(main())
