#!/usr/bin/env python3
"""GPU box tool: run the device CTU encoder on the synthetic clip and report frame times; with a profiling build of the
library (HENC_PROFILE=1 python -c 'import __graft_entry__ as g; g.build()') also the per-phase device timers.

usage: tools/enc_profile.py --width 416 --height 240 --frames 4 [--fixture ctus_416x240]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import encoder_cases as ec  # noqa: E402

PHASES = ["setup", "merge", "me_int", "me_sub(+int)", "pred_inter", "enc_inter", "intra_search", "intra_luma(+search)", "intra_chroma", "consolidate", "wait", "total"]
PRIMS = ["sad", "ssd", "blk(copy/predict/reconst)", "fill_refs", "adi_filter", "intra_pred", "interp", "tr_fwd", "tr_inv", "quant", "dequant", "candidates(merge/amvp)", "sync copies", "info-buffer copies", "ctu begin/end", "waiting for helpers"]
NCOL = 12 + 2 * len(PRIMS) + 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--fixture", default=None, help="teacher-force the reference pictures of this fixture and check the records")
    ap.add_argument("--out", default=None)
    ap.add_argument("--wpp", type=int, default=1, help="wfpp_num_threads (CTU rows = the row-per-thread schedule)")
    a = ap.parse_args()
    lib = C.CDLL(os.environ.get("HOMER_GPU_LIB") or os.path.join(ROOT, "homerhevc_amd", "libhomer_gpu.so"))     # (HOMER_GPU_LIB: a profiling build kept beside the product library)
    lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_frame_ctus.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int] + [C.c_char_p] * 3 + [C.c_double, C.c_char_p]
    lib.hmr_gpu_enc_last_ctu_ms.restype = C.c_float
    lib.hmr_gpu_enc_last_ctu_ms.argtypes = [C.c_void_p]
    lib.hmr_gpu_enc_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.hmr_gpu_last_error.restype = C.c_char_p
    ctx = C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0, lib.hmr_gpu_last_error()
    fx = ec.load_fixture(a.fixture) if a.fixture else None
    w, h = (int(fx["width"]), int(fx["height"])) if fx is not None else (a.width, a.height)
    frames = min(a.frames, int(fx["frames"])) if fx is not None else a.frames
    nx, ny = (w + 63) // 64, (h + 63) // 64
    cfg = ec.default_cfg(w, h, wpp=a.wpp)
    enc = C.c_void_p()
    assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
    recs = C.create_string_buffer(ec.REC * nx * ny)
    prof = (C.c_ulonglong * (ny * NCOL))()
    ysz = w * h
    report = {"width": w, "height": h, "frames": []}
    for f, planes in enumerate(ec.clip_frames(w, h, frames)):
        refs = [None, None, None]
        if fx is not None and f:
            r = fx[f"f{f - 1}_recon"].tobytes()
            refs = [r[:ysz], r[ysz:ysz + ysz // 4], r[ysz + ysz // 4:]]
        st = lib.hmr_gpu_enc_frame_ctus(enc, *planes, 0, *refs, -1.0, recs)
        assert st > 0, lib.hmr_gpu_last_error()
        ms = lib.hmr_gpu_enc_last_ctu_ms(enc)
        lib.hmr_gpu_enc_profile(enc, prof, 1)
        raw = np.array(list(prof), dtype=np.float64).reshape(ny, NCOL)
        p = raw[:, :12] / 100e6 * 1e3   # nominal 100 MHz -> "ms" (the tick is the shader clock on this part: use the shares)
        entry = {"frame": f, "slice": st, "ctu_kernel_ms": round(ms, 3), "ms_per_ctu_step": round(ms / (nx + 2 * (ny - 1)), 3)}
        if p.sum() > 0:
            entry["phase_ms_sum_over_rows"] = {PHASES[k]: round(float(p[:, k].sum()), 2) for k in range(12)}
            entry["busiest_row_ms"] = {PHASES[k]: round(float(p[:, k].max()), 2) for k in range(12)}
        if raw[:, 12:].sum() > 0:
            tot = raw[:, 11].sum()
            entry["primitives"] = {PRIMS[k]: {"share_of_total": round(float(raw[:, 12 + k].sum() / tot), 4), "calls": int(raw[:, 12 + len(PRIMS) + k].sum()),
                                              "ticks_per_call": round(float(raw[:, 12 + k].sum() / max(raw[:, 12 + len(PRIMS) + k].sum(), 1)), 1)} for k in range(len(PRIMS))}
            entry["phase_share_of_total"] = {PHASES[k]: round(float(raw[:, k].sum() / tot), 4) for k in range(11)}
        if fx is not None:
            bad = ec.check_frame_against_fixture(fx, f, recs.raw, w, h)
            entry["mismatches"] = len(bad)
        report["frames"].append(entry)
        print(json.dumps(entry))
    if a.out:
        with open(a.out, "w") as fo:
            json.dump(report, fo, indent=1)


if __name__ == "__main__":
    main()
