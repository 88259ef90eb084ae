#!/usr/bin/env python3
"""GPU box tool: the per-phase / per-primitive device timers of the profiling build (HENC_PROFILE) under BATCH load - N sequences per launch, so that the
workers share their CUs as they do in the bench - summed over all sequences' P frames.  HENC_LDS_BYTES=100000 / 70000 keeps a CU to one / two workers:
comparing the ticks per call between such runs shows which primitives slow down when workers share a CU.

usage: HOMER_GPU_LIB=build/variants/prof/libhomer_gpu.so tools/batch_profile.py --sequences 128 --frames 4 [--out file.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import encoder_cases as ec  # noqa: E402
from enc_profile import PHASES, PRIMS, NCOL  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sequences", type=int, default=128)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lib = C.CDLL(os.environ.get("HOMER_GPU_LIB") or os.path.join(ROOT, "homerhevc_amd", "libhomer_gpu.so"))
    lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.hmr_gpu_last_error.restype = C.c_char_p
    S, w, h = a.sequences, a.width, a.height
    ny = (h + 63) // 64
    seeds = [1234, 1, 2, 3, 4, 5, 6, 7]
    clips = {sd: ec.clip_frames(w, h, a.frames, seed=sd) for sd in seeds}
    encs, bufs = [], []
    for i in range(S):
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0, lib.hmr_gpu_last_error()
        cfg = ec.default_cfg(w, h, wpp=ny)
        assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        for f, planes in enumerate(clips[seeds[i % len(seeds)]]):
            assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0, lib.hmr_gpu_last_error()
        encs.append(enc)
        bufs.append(C.create_string_buffer(4 << 20))
    e_arr = (C.c_void_p * S)(*encs)
    ptrs = (C.c_char_p * S)(*[C.cast(b, C.c_char_p) for b in bufs])
    caps = (C.c_long * S)(*[len(b) for b in bufs])
    got = (C.c_long * S)()
    prof = (C.c_ulonglong * (ny * NCOL))()
    report = {"width": w, "height": h, "sequences": S, "lds_bytes_env": os.environ.get("HENC_LDS_BYTES"), "frames": []}
    for f in range(a.frames):
        t0 = time.perf_counter()
        assert lib.hmr_gpu_enc_encode_batch(e_arr, S, (C.c_int * S)(*([f] * S)), None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        dt = time.perf_counter() - t0
        raw = np.zeros((ny, NCOL))
        for x in encs:
            lib.hmr_gpu_enc_profile(x, prof, 1)
            raw += np.array(list(prof), dtype=np.float64).reshape(ny, NCOL)
        tot = raw[:, 11].sum()
        entry = {"frame": f, "step_s": round(dt, 3), "frames_per_s": round(S / dt, 1), "worker_ticks_total": tot}
        if tot > 0:
            entry["phase_share_of_total"] = {PHASES[k]: round(float(raw[:, k].sum() / tot), 4) for k in range(11)}
            entry["ticks_per_ctu"] = round(float(tot / (S * ny * ((w + 63) // 64))), 0)
            entry["primitives"] = {PRIMS[k]: {"share_of_total": round(float(raw[:, 12 + k].sum() / tot), 4), "calls": int(raw[:, 12 + len(PRIMS) + k].sum()),
                                              "ticks_per_call": round(float(raw[:, 12 + k].sum() / max(raw[:, 12 + len(PRIMS) + k].sum(), 1)), 1)} for k in range(len(PRIMS))}
        report["frames"].append(entry)
        print(json.dumps(entry))
    if a.out:
        with open(a.out, "w") as fo:
            json.dump(report, fo, indent=1)


if __name__ == "__main__":
    main()
