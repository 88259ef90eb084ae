#!/usr/bin/env python3
"""GPU box tool: ONE sequence through hmr_gpu_enc_encode_batch (the batch kernel with a single group) against the reference's per-frame digests
(tests/golden/bench_md5.json).  usage: tools/batch_one.py [workload] [frames] [sequences]"""
import ctypes as C
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import encoder_cases as ec  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2-416x240-encode"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1
width, height, keys = bench.WORKLOADS[workload]
gold = bench.REFERENCE_MD5[workload]["cumulative_md5"]
lib = bench.load_lib()
lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
clip = ec.clip_frames(width, height, frames)
encs, bufs = [], []
for _ in range(S):
    ctx, enc = C.c_void_p(), C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0
    cfg = ec.default_cfg(width, height, **keys)
    assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
    for f, planes in enumerate(clip):
        assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0
    encs.append(enc)
    bufs.append(C.create_string_buffer(4 << 20))
e_arr = (C.c_void_p * S)(*encs)
ptrs = (C.c_char_p * S)(*[C.cast(b, C.c_char_p) for b in bufs])
caps = (C.c_long * S)(*[len(b) for b in bufs])
got = (C.c_long * S)()
md5 = [hashlib.md5() for _ in range(S)]
for f in range(frames):
    assert lib.hmr_gpu_enc_encode_batch(e_arr, S, (C.c_int * S)(*([f] * S)), None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
    bad = []
    for i in range(S):
        md5[i].update(C.string_at(bufs[i], got[i]))
        if md5[i].hexdigest() != gold[f]:
            bad.append(i)
    print(f"frame {f}: {got[0]} bytes, sequences off the reference: {bad[:10]}{' ...' if len(bad) > 10 else ''} ({len(bad)} of {S})", flush=True)
    if bad:
        break
