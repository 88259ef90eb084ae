#!/usr/bin/env python3
"""GPU box tool: ONE sequence encoded by its E engine objects in chains of overlapping frames (hmr_gpu_enc_encode_chain, include/homer_gpu.h section 12c) - frames/s of
the chains after the first (which holds the I frame and the allocations), the CTU launch's duration per chain, every access unit checked against the compiled
reference's digests (tests/golden/bench_md5.json).  usage: tools/chain_bench.py [workload[:chain[:objects per engine]] ...]   (cfg2-1080p-encode-engines2/4/8, cfg2-2160p-encode-engines2/4/8)"""
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

GOLD = json.load(open(os.path.join(ec.GOLDEN, "bench_md5.json")))


def run(lib, workload, chain=None, frames=None, sets=1, quiet=False):
    g = GOLD[workload]
    w, h, keys = g["width"], g["height"], dict(g["keys"])
    frames = min(frames or g["frames"], g["frames"])
    E = keys["engines"]
    chain = chain or E * sets
    lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.hmr_gpu_enc_create_engine.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.c_int, C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_create_engine_twin.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_chain.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.hmr_gpu_enc_destroy.argtypes = [C.c_void_p]
    lib.hmr_gpu_destroy.argtypes = [C.c_void_p]
    lib.hmr_gpu_last_error.restype = C.c_char_p
    lib.hmr_gpu_enc_last_ctu_ms.restype = C.c_float
    lib.hmr_gpu_enc_last_ctu_ms.argtypes = [C.c_void_p]
    cfg = ec.default_cfg(w, h, **keys)
    ctxs, encs = [], []
    for k in range(E * sets):          # object k: engine k % E; the sets beyond the first are twins (the engine's persistent state is shared)
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0, lib.hmr_gpu_last_error()
        if k < E:
            assert lib.hmr_gpu_enc_create_engine(ctx, C.byref(cfg), k, C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        else:
            assert lib.hmr_gpu_enc_create_engine_twin(ctx, encs[k % E], C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        ctxs.append(ctx)
        encs.append(enc)
    obj_of, slot_of, used = {}, {}, [0] * len(encs)
    for f in range(frames):
        k = ((f % chain) // E) * E + f % E
        obj_of[f], slot_of[f] = k, used[k]
        used[k] += 1
    for f, planes in enumerate(ec.clip_frames(w, h, frames)):
        assert lib.hmr_gpu_enc_load_source(encs[obj_of[f]], slot_of[f], *planes) == 0, lib.hmr_gpu_last_error()
    bufs = [C.create_string_buffer(8 << 20) for _ in range(chain)]
    units, walls, kernel_ms = [], [], []
    for first in range(0, frames, chain):
        fs = list(range(first, min(first + chain, frames)))
        n = len(fs)
        e_arr = (C.c_void_p * n)(*[encs[obj_of[f]] for f in fs])
        slots = (C.c_int * n)(*[slot_of[f] for f in fs])
        ptrs = (C.c_char_p * n)(*[C.cast(bufs[i], C.c_char_p) for i in range(n)])
        caps = (C.c_long * n)(*[len(bufs[i]) for i in range(n)])
        got = (C.c_long * n)()
        prev = encs[obj_of[first - 1]] if first else None
        t0 = time.perf_counter()
        assert lib.hmr_gpu_enc_encode_chain(e_arr, n, prev, slots, None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        walls.append((n, time.perf_counter() - t0))
        kernel_ms.append(round(lib.hmr_gpu_enc_last_ctu_ms(encs[obj_of[fs[0]]]), 1))
        units += [bufs[i].raw[:got[i]] for i in range(n)]
    for enc in reversed(encs):
        lib.hmr_gpu_enc_destroy(enc)
    for ctx in ctxs:
        lib.hmr_gpu_destroy(ctx)
    md5, ok = hashlib.md5(), True
    for f, u in enumerate(units):
        md5.update(u)
        ok = ok and md5.hexdigest() == g["cumulative_md5"][f]
    timed = walls[1:]
    fps = sum(n for n, _ in timed) / sum(t for _, t in timed) if timed else 0.0
    full = [(n, t) for n, t in timed if n == chain]       # (the last chain of a clip is usually a short one: its engines idle at the end like the first chain's at the start)
    fps_full = sum(n for n, _ in full) / sum(t for _, t in full) if full else 0.0
    out = {"workload": workload, "engines": E, "objects_per_engine": sets, "chain": chain, "frames": frames, "frames_per_s_full_chains": round(fps_full, 2),
           "frames_per_s_after_first_chain": round(fps, 2), "stream_matches_reference": ok,
           "wall_ms_per_chain": [round(t * 1e3, 1) for _, t in walls], "ctu_launch_ms_per_chain": kernel_ms}
    if not quiet:
        print(json.dumps(out))
    return out


if __name__ == "__main__":
    lib = libs.load_gpu()
    for wl in sys.argv[1:] or ["cfg2-1080p-encode-engines4"]:
        name, _, rest = wl.partition(":")          # workload[:chain[:sets]]
        ch, _, st = rest.partition(":")
        run(lib, name, int(ch) if ch else None, sets=int(st) if st else 1)
