#!/usr/bin/env python3
"""GPU box tool (profiling build: HENC_PROFILE=1 python -c 'import __graft_entry__ as g; g.build()'): per-CTU timestamps of the row-per-thread
schedule for the last frame of a short 1080p encode, and what other barrier placements would give with the same CTU durations.

usage: tools/lockstep_timeline.py [--frames 5] [--width 1920 --height 1080]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import encoder_cases as ec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=5)
    a = ap.parse_args()
    lib = bench.load_lib()
    W, H = (a.width + 63) // 64, (a.height + 63) // 64
    ctx, enc = C.c_void_p(), C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0
    cfg = ec.default_cfg(a.width, a.height, wpp=H)
    assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
    buf, n = C.create_string_buffer(16 << 20), C.c_long()
    for f, planes in enumerate(ec.clip_frames(a.width, a.height, a.frames)):
        assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0
    for f in range(a.frames):
        assert lib.hmr_gpu_enc_encode_source(enc, f, 0, buf, len(buf), C.byref(n), None) in (1, 2)
    tl = (C.c_ulonglong * (W * H * 4))()
    lib.hmr_gpu_enc_timeline.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.hmr_gpu_enc_timeline(enc, tl) == 0
    t = np.array(list(tl), dtype=np.float64).reshape(H, W, 4) / 100e3   # ms
    t0 = t[..., 0].min()
    dur = t[..., 3] - t[..., 1]
    first = np.where(t[..., 2] > 0, t[..., 2] - t[..., 1], dur)          # CTUs that never use the share: no constraint before their end
    step = np.add.outer(2 * np.arange(H), np.arange(W))
    nsteps = int(step.max()) + 1
    smax = np.array([dur[step == s].max() for s in range(nsteps)])
    smean = np.array([dur[step == s].mean() for s in range(nsteps)])

    def simulate(mode):
        end = np.zeros((H, W))
        step_end = np.zeros(nsteps)
        for s in range(nsteps):
            for r in range(H):
                c = s - 2 * r
                if c < 0 or c >= W:
                    continue
                ready = max(end[r, c - 1] if c else 0.0, end[r - 1, min(c + 1, W - 1)] if r else 0.0)
                barrier = step_end[:s].max() if s else 0.0
                if mode == "lockstep":
                    e = max(ready, barrier) + dur[r, c]
                elif mode == "lazy":
                    e = max(max(ready + first[r, c], barrier) + dur[r, c] - first[r, c], ready + dur[r, c])
                else:
                    e = ready + dur[r, c]
                end[r, c] = e
            step_end[s] = max(end[r, s - 2 * r] for r in range(H) if 0 <= s - 2 * r < W)
        return float(end.max())

    out = {"frame_ms_measured": round(float(t[..., 3].max() - t0), 1), "ctu_ms_mean": round(float(dur.mean()), 2), "ctu_ms_p50_p90_max": [round(float(np.percentile(dur, q)), 2) for q in (50, 90, 100)],
           "sum_of_step_max_ms": round(float(smax.sum()), 1), "sum_of_step_mean_ms": round(float(smean.sum()), 1),
           "ctus_using_the_share": int((t[..., 2] > 0).sum()), "first_use_fraction_of_ctu_mean": round(float((first / dur)[t[..., 2] > 0].mean()), 3),
           "model_lockstep_ms": round(simulate("lockstep"), 1), "model_lazy_barrier_ms": round(simulate("lazy"), 1), "model_no_barrier_ms": round(simulate("free"), 1),
           "row_mean_ctu_ms": [round(float(x), 2) for x in dur.mean(axis=1)]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
