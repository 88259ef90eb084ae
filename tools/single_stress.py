#!/usr/bin/env python3
"""GPU box tool: create / encode / destroy one fixture sequence over and over in one process, each time against the fixture's md5.
usage: tools/single_stress.py case iterations [other_case_to_interleave]"""
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


def run(lib, case, buf, n):
    g = GOLD[case]
    keys = dict(g["keys"])
    cut_at = keys.pop("cut_at", None)
    ctx, enc = C.c_void_p(), C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0
    cfg = ec.default_cfg(g["width"], g["height"], **keys)
    assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
    md5, sizes = hashlib.md5(), []
    for f, planes in enumerate(ec.clip_frames(g["width"], g["height"], g["frames"], cut_at)):
        assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0
    for f in range(g["frames"]):
        assert lib.hmr_gpu_enc_encode_source(enc, f, 0, buf, len(buf), C.byref(n), None) in (1, 2)
        md5.update(C.string_at(buf, n.value))
        sizes.append(n.value)
    lib.hmr_gpu_enc_destroy(enc)
    return md5.hexdigest() == g["stream_md5"], sizes


def main():
    case, iters = sys.argv[1], int(sys.argv[2])
    other = sys.argv[3] if len(sys.argv) > 3 else None
    lib = bench.load_lib()
    buf, n = C.create_string_buffer(4 << 20), C.c_long()
    bad = 0
    for it in range(iters):
        if other:
            run(lib, other, buf, n)
        ok, sizes = run(lib, case, buf, n)
        if not ok:
            bad += 1
            print(f"iteration {it}: {case} differs, access unit sizes {sizes}", flush=True)
    print(f"{case}: {iters} iterations, {bad} wrong")


if __name__ == "__main__":
    main()
