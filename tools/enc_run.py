#!/usr/bin/env python3
"""GPU box tool: encode the synthetic clip with the device encoder (hmr_gpu_enc_encode) and print per-frame schedule statistics and times.
Meant to be run bare or under rocprofv3 (`rocprofv3 --kernel-trace --stats -- python3 tools/enc_run.py ...`).

usage: tools/enc_run.py --width 1920 --height 1080 --frames 8 [key=value ...]
"""
import argparse
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import encoder_cases as ec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("keys", nargs="*")
    a = ap.parse_args()
    keys = dict(k.split("=") for k in a.keys)
    lib = C.CDLL(os.path.join(ROOT, "homerhevc_amd", "libhomer_gpu.so"))
    lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_source.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_long, C.POINTER(C.c_long), C.c_char_p]
    lib.hmr_gpu_enc_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.hmr_gpu_last_error.restype = C.c_char_p
    ctx, enc = C.c_void_p(), C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0, lib.hmr_gpu_last_error()
    cfg = ec.default_cfg(a.width, a.height, **keys)
    assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
    for f, planes in enumerate(ec.clip_frames(a.width, a.height, a.frames)):
        assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0, lib.hmr_gpu_last_error()
    buf = C.create_string_buffer(8 << 20)
    nbytes = C.c_long()
    md5 = hashlib.md5()
    t0 = time.time()
    for f in range(a.frames):
        st = lib.hmr_gpu_enc_encode_source(enc, f, 0, buf, len(buf), C.byref(nbytes), None)
        assert st in (1, 2), lib.hmr_gpu_last_error()
        md5.update(C.string_at(buf, nbytes.value))
        p, n, ms, tot = C.c_int(), C.c_int(), C.c_float(), C.c_float()
        lib.hmr_gpu_enc_last_stats(enc, C.byref(p), C.byref(n), C.byref(ms), C.byref(tot))
        print(f"frame {f}: slice {st} {nbytes.value} bytes, {p.value} passes, {n.value} CTU encodes, CTU passes {ms.value:.1f} ms, frame {tot.value:.1f} ms")
    dt = time.time() - t0
    if hasattr(lib, "hmr_gpu_enc_post_profile"):
        prof = (C.c_ulonglong * 16)()
        lib.hmr_gpu_enc_post_profile.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
        lib.hmr_gpu_enc_post_profile(enc, prof)
        names = ["D task", "P load", "P stats", "P candidates", "P decide+sao syntax", "P ctu syntax", "P apply+pad", "scan", "D count", "P count"]
        if any(prof):
            nd, npp = max(prof[8], 1), max(prof[9], 1)
            print(f"entropy coder: encode_residual {prof[10] / max(prof[11], 1):.0f} ticks per call ({prof[13] / max(prof[11], 1):.0f} of them its gather), {prof[11] / npp:.1f} calls and {prof[12] / npp:.0f} context-coded bins per CTU; residual share of the CTU syntax {prof[10] / max(prof[5], 1):.2f}")
            print("post-decision stage (100 MHz ticks -> us per task): " + ", ".join(f"{names[k]} {prof[k] / 100.0 / (nd if k == 0 else npp):.1f}" for k in range(8)) + f"; {prof[8]} D, {prof[9]} P tasks")
    print(f"{a.frames} frames in {dt:.2f} s = {a.frames / dt:.2f} fps; stream md5 {md5.hexdigest()}")


if __name__ == "__main__":
    main()
