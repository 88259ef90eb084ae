#!/usr/bin/env python3
"""GPU box tool: N copies of one fixture sequence through hmr_gpu_enc_encode_batch (chained when N exceeds the number of groups), each against the fixture's md5.
usage: tools/batch_copies.py case N [case N ...]   (one after the other in the same process)"""
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


def main():
    lib = bench.load_lib()
    for k in range(1, len(sys.argv) - 1, 2):
        run(lib, sys.argv[k], int(sys.argv[k + 1]))


def run(lib, case, n):
    g = GOLD[case]
    keys = dict(g["keys"])
    cut_at = keys.pop("cut_at", None)
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    frames = ec.clip_frames(g["width"], g["height"], g["frames"], cut_at)
    encs, bufs = [], []
    for _ in range(n):
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0
        cfg = ec.default_cfg(g["width"], g["height"], **keys)
        assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        for f, planes in enumerate(frames):
            assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0
        encs.append(enc)
        bufs.append(C.create_string_buffer(4 << 20))
    e_arr = (C.c_void_p * n)(*encs)
    ptrs = (C.c_char_p * n)(*[C.cast(b, C.c_char_p) for b in bufs])
    caps = (C.c_long * n)(*[len(b) for b in bufs])
    got = (C.c_long * n)()
    md5 = [hashlib.md5() for _ in range(n)]
    per_frame = []
    for f in range(g["frames"]):
        if n == 1:      # one sequence: the frame-by-frame entry (either schedule)
            nb = C.c_long()
            assert lib.hmr_gpu_enc_encode_source(encs[0], f, 0, bufs[0], len(bufs[0]), C.byref(nb), None) in (1, 2), lib.hmr_gpu_last_error()
            got[0] = nb.value
        else:
            assert lib.hmr_gpu_enc_encode_batch(e_arr, n, (C.c_int * n)(*([f] * n)), None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        aus = [hashlib.md5(C.string_at(bufs[i], got[i])).hexdigest() for i in range(n)]
        per_frame.append(aus)
        for i in range(n):
            md5[i].update(C.string_at(bufs[i], got[i]))
    wrong = [i for i in range(n) if md5[i].hexdigest() != g["stream_md5"]]
    print(f"{case} x {n}: {len(wrong)} wrong streams {wrong[:20]}")
    for f, aus in enumerate(per_frame):
        ref = max(set(aus), key=aus.count)
        odd = [i for i in range(n) if aus[i] != ref]
        if odd:
            print(f"  frame {f}: {len(odd)} sequences differ from the majority: {odd[:20]}")
    for enc in encs:
        lib.hmr_gpu_enc_destroy(enc)


if __name__ == "__main__":
    main()
