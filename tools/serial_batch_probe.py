#!/usr/bin/env python3
"""GPU box: S sequences of the 1080p cfg-2 clip in the reference's single-thread order (wfpp_num_threads = 1) as ONE batch (hmr_gpu_enc_create_serial_pool + the batch call):
every stream against the reference's digests (tests/golden/bench_md5.json: cfg2-1080p-encode-single-thread-order), frames/s of the P frames.
usage: tools/serial_batch_probe.py [sequences [frames [WxH]]]"""
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import encoder_cases as ec  # noqa: E402
import libs  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w, h = (int(v) for v in sys.argv[3].split("x")) if len(sys.argv) > 3 else (1920, 1080)
lib = libs.load_gpu()
lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
lib.hmr_gpu_enc_create_serial_pool.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
lib.hmr_gpu_last_error.restype = C.c_char_p
clip = list(ec.clip_frames(w, h, frames))
encs = []
for i in range(S):
    ctx, enc = C.c_void_p(), C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0
    cfg = ec.default_cfg(w, h)
    assert lib.hmr_gpu_enc_create_serial_pool(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
    for f, planes in enumerate(clip):
        assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0
    encs.append(enc)
bufs = [C.create_string_buffer(2 << 20) for _ in range(S)]
e_arr = (C.c_void_p * S)(*encs)
ptrs = (C.c_char_p * S)(*[C.cast(b, C.c_char_p) for b in bufs])
caps = (C.c_long * S)(*[len(b) for b in bufs])
got = (C.c_long * S)()
md5 = [hashlib.md5() for _ in range(S)]
cum, times = [], []
for f in range(frames):
    t0 = time.perf_counter()
    assert lib.hmr_gpu_enc_encode_batch(e_arr, S, (C.c_int * S)(*([f] * S)), None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
    times.append(time.perf_counter() - t0)
    for i in range(S):
        md5[i].update(bufs[i].raw[:got[i]])
    cum.append(md5[0].hexdigest())
same = len({m.hexdigest() for m in md5}) == 1
ref = None
if (w, h) == (1920, 1080):
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_md5.json")))["cfg2-1080p-encode-single-thread-order"]["cumulative_md5"][:frames]
print(json.dumps({"sequences": S, "frames": frames, "all_sequences_identical": same, "matches_reference": (cum == ref) if ref else None, "stream_md5": cum[-1],
                  "s_per_step": [round(t, 3) for t in times], "frames_per_s_P": round(S * (frames - 1) / sum(times[1:]), 2) if frames > 1 else None}))
