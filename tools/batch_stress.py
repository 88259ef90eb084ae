#!/usr/bin/env python3
"""GPU box tool: repeat a mixed batch (hmr_gpu_enc_encode_batch) and compare every access unit with the one the same sequence produces alone.
usage: tools/batch_stress.py [iterations]"""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import encoder_cases as ec  # noqa: E402

GOLD = json.load(open(os.path.join(ec.GOLDEN, "streams.json")))
CASES = ["416x240_wpp_rows", "832x480_wpp_rows", "416x240_scene_cut_wpp_rows", "328x264_wpp3"]


def make(lib, case):
    g = GOLD[case]
    keys = dict(g["keys"])
    cut_at = keys.pop("cut_at", None)
    ctx, enc = C.c_void_p(), C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0
    cfg = ec.default_cfg(g["width"], g["height"], **keys)
    assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
    for f, planes in enumerate(ec.clip_frames(g["width"], g["height"], g["frames"], cut_at)):
        assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0
    return enc, g["frames"]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    lib = bench.load_lib()
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    buf, n = C.create_string_buffer(1 << 20), C.c_long()
    alone = []
    for case in CASES:
        enc, frames = make(lib, case)
        aus = []
        for f in range(frames):
            assert lib.hmr_gpu_enc_encode_source(enc, f, 0, buf, len(buf), C.byref(n), None) in (1, 2)
            aus.append(hashlib.md5(C.string_at(buf, n.value)).hexdigest())
        lib.hmr_gpu_enc_destroy(enc)
        alone.append(aus)
    bad = 0
    for it in range(iters):
        made = [make(lib, case) for case in CASES]
        encs, frames = [m[0] for m in made], [m[1] for m in made]
        bufs = [C.create_string_buffer(1 << 20) for _ in CASES]
        for f in range(max(frames)):
            live = [i for i in range(len(CASES)) if f < frames[i]]
            k = len(live)
            e_arr = (C.c_void_p * k)(*[encs[i] for i in live])
            ptrs = (C.c_char_p * k)(*[C.cast(bufs[i], C.c_char_p) for i in live])
            caps = (C.c_long * k)(*[len(bufs[i]) for i in live])
            got = (C.c_long * k)()
            assert lib.hmr_gpu_enc_encode_batch(e_arr, k, (C.c_int * k)(*([f] * k)), None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
            for j, i in enumerate(live):
                if hashlib.md5(C.string_at(bufs[i], got[j])).hexdigest() != alone[i][f]:
                    print(f"iteration {it}: sequence {i} ({CASES[i]}) frame {f} differs ({got[j]} bytes)", flush=True)
                    bad += 1
        for e in encs:
            lib.hmr_gpu_enc_destroy(e)
    print(f"{iters} iterations, {bad} differing access units")


if __name__ == "__main__":
    main()
