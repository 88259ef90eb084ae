#!/usr/bin/env python3
"""Build-container tool: random configurations of the frame encoder - checker build (oracle/libenc_cpu.so: the device's source compiled for one lane) against the
compiled reference (oracle/_ref/ref_lockstep, or ref_ctudump under HOMER_TURNSTILE for several WPP threads / engines) - beyond the fixed cases of
tests/golden/streams.json: other picture sizes, clip seeds, QPs, rate-control targets, RD modes, transform depths, thread and engine counts.

usage: tools/encoder_fuzz.py [--cases N] [--seed S] [--max-ctus M]        (prints one line per case; exit code = number of differing cases)
       tools/encoder_fuzz.py WxHxFRAMES:clipseed[:key=value,...] ...      (explicit cases)
       tools/encoder_fuzz.py --gpu ...                                     (GPU box: the device encoder, hmr_gpu_enc_encode, instead of the checker build; the
                                                                            compiled reference travels there as oracle/_ref/)
       ... --gpu --batch K                                                  (K cases per hmr_gpu_enc_encode_batch call: one launch for all their CTU stages)
       ... --gpu --engines-only --chain-sets M                              (several engines through hmr_gpu_enc_encode_chain, M objects per engine)
       ... --threads-only / --extra-keys / --max-cols C --max-rows R        (several WPP threads only; also draw me= and cqo=; larger CTU grids)
A case that differs and had evaluations on a stale prediction window (quirk Q12, include/homer_gpu.h: hmr_gpu_enc_stale_predictions) is marked; --tolerate-q12 keeps it out
of the exit code.  Runs of the round: profiles/r04_encoder_fuzz.md."""
import argparse
import ctypes as C
import hashlib
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import encoder_cases as ec  # noqa: E402
import gen_yuv  # noqa: E402

CPU_SO = os.path.join(ROOT, "oracle", "libenc_cpu.so")
DECODE = False      # --decode: the reference's stream also goes through oracle/hevcdec, decoded pictures against the reference's own reconstruction
LAST_RECON = None
STALE = 0      # evaluations on a stale prediction window (quirk Q12) in the case just encoded; -1: not counted


def reference(width, height, frames, clip_seed, keys):
    keys = dict(keys)
    cut_at = keys.pop("cut_at", None)
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "in.yuv")
        gen_yuv.write_clip(yuv, width, height, frames, clip_seed, cut_at)
        turnstile = int(keys.get("wpp", 1)) > 1 or int(keys.get("engines", 1)) > 1
        cmd = [os.path.join(ROOT, "oracle", "_ref", "ref_ctudump" if turnstile else "ref_lockstep"), yuv, os.path.join(tmp, "out.265"), str(width), str(height), str(frames)]
        cmd += [f"{k}={v}" for k, v in keys.items()]
        global LAST_RECON
        LAST_RECON = None
        if DECODE:
            cmd.append("recon=" + os.path.join(tmp, "rec.yuv"))
        try:
            subprocess.run(cmd, check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, HOMER_TURNSTILE="1") if turnstile else None)
        except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as e:      # (the reference has configurations it deadlocks or crashes on)
            return type(e).__name__
        if DECODE:
            LAST_RECON = open(os.path.join(tmp, "rec.yuv"), "rb").read()
        return open(os.path.join(tmp, "out.265"), "rb").read()


def decoder_verdict(stream, recon, width, height, keys, stale):
    """the reference's stream through oracle/hevcdec (a decoder written from the standard, test infrastructure): 'identical', or which documented drift of the reference
    explains the difference (R1 rate control, R2 quirk Q12), or 'UNEXPLAINED'"""
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "in.265"), "wb").write(stream)
        r = subprocess.run([os.path.join(ROOT, "oracle", "hevcdec"), os.path.join(tmp, "in.265"), os.path.join(tmp, "dec.yuv")], capture_output=True, text=True)
        if r.returncode:
            return "UNEXPLAINED: hevcdec " + r.stderr.strip()
        dec = open(os.path.join(tmp, "dec.yuv"), "rb").read()
    if dec == recon:
        return "identical"
    fsz = width * height * 3 // 2
    first = next((i for i in range(len(recon) // fsz) if dec[i * fsz:(i + 1) * fsz] != recon[i * fsz:(i + 1) * fsz]), -1)
    if int(keys.get("bitrate_mode", 0)):
        return f"R1 (rate control: deblocking QP) from picture {first}"
    if stale > 0:
        return f"R2 (quirk Q12) from picture {first}"
    return f"UNEXPLAINED: pictures differ from {first}"



def checker(lib, width, height, frames, clip_seed, keys):
    keys = dict(keys)
    cut_at = keys.pop("cut_at", None)
    image_type = 3 if int(keys.pop("force_intra", 0)) else 0
    cfg = ec.default_cfg(width, height, **keys)
    enc = lib.henc_cpu_create(C.byref(cfg))
    if not enc:
        return None
    buf = C.create_string_buffer(8 << 20)
    rec = C.create_string_buffer(width * height * 3 // 2)
    units = []
    global STALE
    STALE = 0
    for planes in ec.clip_frames(width, height, frames, cut_at, clip_seed):
        n = lib.henc_cpu_encode_frame(enc, *planes, image_type, buf, len(buf), rec)
        assert n > 0
        units.append(buf.raw[:n])
        STALE += lib.henc_cpu_stale_predictions(enc)
    lib.henc_cpu_destroy(enc)
    return units


def device(lib, ctx, width, height, frames, clip_seed, keys):
    keys = dict(keys)
    cut_at = keys.pop("cut_at", None)
    image_type = 3 if int(keys.pop("force_intra", 0)) else 0
    cfg = ec.default_cfg(width, height, **keys)
    enc = C.c_void_p()
    if lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) != 0:
        return None
    buf = C.create_string_buffer(8 << 20)
    nbytes = C.c_long()
    units = []
    for planes in ec.clip_frames(width, height, frames, cut_at, clip_seed):
        st = lib.hmr_gpu_enc_encode(enc, *planes, image_type, buf, len(buf), C.byref(nbytes), None)
        assert st in (1, 2), lib.hmr_gpu_last_error()
        units.append(buf.raw[:nbytes.value])
    global STALE
    tot = C.c_long()
    lib.hmr_gpu_enc_stale_predictions(enc, None, C.byref(tot))
    STALE = tot.value if cfg.wfpp_num_threads > 1 else -1
    lib.hmr_gpu_enc_destroy(enc)
    return units


SERIAL_POOL = False      # --serial-batch: one-thread cases as batches through hmr_gpu_enc_create_serial_pool


def device_batch(lib, group):
    """the cases of a group advance together: one hmr_gpu_enc_encode_batch call (ONE launch for all their CTU stages) per frame step"""
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.hmr_gpu_destroy.argtypes = [C.c_void_p]
    encs, ctxs, types, nfr = [], [], [], []
    for w, h, frames, clip_seed, keys in group:
        keys = dict(keys)
        cut_at = keys.pop("cut_at", None)
        types.append(3 if int(keys.pop("force_intra", 0)) else 0)
        cfg = ec.default_cfg(w, h, **keys)
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0, lib.hmr_gpu_last_error()
        assert (lib.hmr_gpu_enc_create_serial_pool if SERIAL_POOL else lib.hmr_gpu_enc_create)(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        for f, planes in enumerate(ec.clip_frames(w, h, frames, cut_at, clip_seed)):
            assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0, lib.hmr_gpu_last_error()
        encs.append(enc); ctxs.append(ctx); nfr.append(frames)
    bufs = [C.create_string_buffer(8 << 20) for _ in group]
    out = [[] for _ in group]
    for f in range(max(nfr)):
        live = [i for i in range(len(group)) if f < nfr[i]]
        n = len(live)
        e_arr = (C.c_void_p * n)(*[encs[i] for i in live])
        slots = (C.c_int * n)(*([f] * n))
        its = (C.c_int * n)(*[types[i] for i in live])
        ptrs = (C.c_char_p * n)(*[C.cast(bufs[i], C.c_char_p) for i in live])
        caps = (C.c_long * n)(*[len(bufs[i]) for i in live])
        got = (C.c_long * n)()
        assert lib.hmr_gpu_enc_encode_batch(e_arr, n, slots, its, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        for k, i in enumerate(live):
            out[i].append(bufs[i].raw[:got[k]])
    stale = []
    for i in range(len(group)):
        tot = C.c_long()
        lib.hmr_gpu_enc_stale_predictions(encs[i], None, C.byref(tot))
        stale.append(tot.value)
        lib.hmr_gpu_enc_destroy(encs[i])
        lib.hmr_gpu_destroy(ctxs[i])
    return out, stale


def device_chain(lib, w, h, frames, clip_seed, keys, sets):
    """a case with several engines through hmr_gpu_enc_encode_chain: the engine objects (and `sets` - 1 twins of each) encode chains of E x sets overlapping frames"""
    keys = dict(keys)
    cut_at = keys.pop("cut_at", None)
    if int(keys.pop("force_intra", 0)):
        return None
    E = int(keys["engines"])
    chain = E * sets
    lib.hmr_gpu_enc_create_engine.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.c_int, C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_create_engine_twin.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_chain.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.hmr_gpu_destroy.argtypes = [C.c_void_p]
    cfg = ec.default_cfg(w, h, **keys)
    ctxs, encs = [], []
    for k in range(E * sets):
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0, lib.hmr_gpu_last_error()
        rc = lib.hmr_gpu_enc_create_engine(ctx, C.byref(cfg), k, C.byref(enc)) if k < E else lib.hmr_gpu_enc_create_engine_twin(ctx, encs[k % E], C.byref(enc))
        if rc != 0:
            for x in reversed(encs):
                lib.hmr_gpu_enc_destroy(x)
            return None
        ctxs.append(ctx); encs.append(enc)
    obj_of, slot_of, used = {}, {}, [0] * len(encs)
    for f in range(frames):
        k = ((f % chain) // E) * E + f % E
        obj_of[f], slot_of[f] = k, used[k]
        used[k] += 1
    for f, planes in enumerate(ec.clip_frames(w, h, frames, cut_at, clip_seed)):
        assert lib.hmr_gpu_enc_load_source(encs[obj_of[f]], slot_of[f], *planes) == 0, lib.hmr_gpu_last_error()
    bufs = [C.create_string_buffer(8 << 20) for _ in range(chain)]
    units = []
    for first in range(0, frames, chain):
        fs = list(range(first, min(first + chain, frames)))
        n = len(fs)
        e_arr = (C.c_void_p * n)(*[encs[obj_of[f]] for f in fs])
        slots = (C.c_int * n)(*[slot_of[f] for f in fs])
        ptrs = (C.c_char_p * n)(*[C.cast(bufs[i], C.c_char_p) for i in range(n)])
        caps = (C.c_long * n)(*[len(bufs[i]) for i in range(n)])
        got = (C.c_long * n)()
        prev = encs[obj_of[first - 1]] if first else None
        rc = lib.hmr_gpu_enc_encode_chain(e_arr, n, prev, slots, None, ptrs, caps, got)
        if rc != 0:
            units = lib.hmr_gpu_last_error().decode()
            break
        units += [bufs[i].raw[:got[i]] for i in range(n)]
    global STALE
    STALE = 0
    for x in encs:
        tot = C.c_long()
        lib.hmr_gpu_enc_stale_predictions(x, None, C.byref(tot))
        STALE += tot.value
    for x in reversed(encs):
        lib.hmr_gpu_enc_destroy(x)
    for x in ctxs:
        lib.hmr_gpu_destroy(x)
    return units


def load_checker():
    lib = C.CDLL(CPU_SO)
    lib.henc_cpu_create.restype = C.c_void_p
    lib.henc_cpu_create.argtypes = [C.POINTER(ec.EncCfg)]
    lib.henc_cpu_encode_frame.restype = C.c_long
    lib.henc_cpu_encode_frame.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int, C.c_char_p, C.c_long, C.c_char_p]
    lib.henc_cpu_destroy.argtypes = [C.c_void_p]
    lib.henc_cpu_stale_predictions.restype = C.c_long
    lib.henc_cpu_stale_predictions.argtypes = [C.c_void_p]
    return lib


def random_case(rng, max_ctus, gpu=False, max_cols=14, max_rows=9, threads_only=False, engines_only=False, extra_keys=False, combos=False):
    while True:
        wc, hc = rng.randint(2, max_cols), rng.randint(1, max_rows)
        if wc * hc > max_ctus:
            continue
        # multiples of 8 (the minimum CU); the last CTU column / row partly outside the picture most of the time
        width = wc * 64 - 8 * rng.randint(0, 7)
        height = hc * 64 - 8 * rng.randint(0, 7)
        if width < 72 or height < 16 or (wc == 2 and hc > 1) or (wc == 3 and hc >= 4):        # (two columns x several rows: the reference crashes; 3 x 4 and taller: refused, enc_host.h)
            continue
        keys = {}
        sao = rng.random() < 0.75
        if sao and wc <= 5 and hc >= max(wc, 4):        # (a refused corner of the lagged filter pipeline, enc_host.h)
            sao = False
        if not sao:
            keys["sao"] = 0
        mode = "engines" if engines_only else rng.choice(["fixed", "fixed", "rc", "rdfull", "engines"])
        # --combos (round 6): rate control / RD_FULL with ONE thread (the pool's raster schedule) and with several engines
        combo = rng.choice(["rc1", "rd1", "rc_eng"]) if combos else None      # ("rd_eng": RD_FULL with several engines is refused, enc_host.h)
        if combo:
            mode = "rc" if combo.startswith("rc") else "rdfull"
        if engines_only and not (wc >= 9 or (wc >= 3 and hc <= 4)):
            continue
        wpp = 1
        if rng.random() < 0.6 or mode == "rdfull" or threads_only or engines_only:
            wpp = hc if hc <= 32 else 32
            if hc > 2 and rng.random() < 0.25 and mode != "rdfull":
                n = rng.randint(2, hc - 1)
                if 2 * n >= wc:
                    wpp = n
        if combo in ("rc1", "rd1"):
            wpp = 1
        elif combo:
            wpp = hc if hc <= 32 else 32
            if wpp < 2 or not (wc >= 9 or (wc >= 3 and hc <= 4 and combo == "rc_eng")):      # (RD_FULL with several engines: nine or more CTU columns, enc_host.h)
                continue
        if (mode == "rdfull" or threads_only or engines_only) and wpp < 2 and not combo:
            continue
        if gpu and mode in ("rc", "engines") and wpp < 2 and not combo:       # (several engines need a thread per CTU row on the device)
            continue
        if wpp > 1:
            keys["wpp"] = wpp
        keys["qp"] = rng.choice([12, 17, 22, 27, 30, 32, 32, 35, 38, 42, 47, 51])
        keys["perf"] = rng.choice([0, 1, 2, 2, 3])
        if rng.random() < 0.3:
            keys["sign_hiding"] = 0
        if rng.random() < 0.4:
            keys["intra_tr"] = rng.choice([1, 3, 4])
        if rng.random() < 0.25:
            keys["inter_tr"] = rng.choice([2, 3, 4])
        if extra_keys and rng.random() < 0.3:
            keys["me"] = rng.choice([0, 1])              # motion_estimation_precision: whole / half samples
        if extra_keys and rng.random() < 0.3:
            keys["cqo"] = rng.choice([-4, -1, 0, 1, 5])  # chroma_qp_offset
        frames = rng.randint(2, 5)
        if mode == "rc":
            keys["bitrate_mode"] = rng.choice([1, 2])
            keys["bitrate"] = max(100, int(width * height * rng.choice([1.5, 3, 6, 12]) / 1000))
            frames = rng.randint(4, 8)
        elif mode == "rdfull":
            keys["rd"] = 1
            if combo:
                frames = rng.randint(3, 6)
        elif mode == "engines" and (wc >= 9 or (wc >= 3 and hc <= 4)):      # (narrower: the reference's engines deadlock, enc_host.h)
            keys["engines"] = rng.choice([2, 3, 4])
            frames = rng.randint(5, 14 if engines_only else 10)
        elif rng.random() < 0.25:
            keys["rd"] = 0
        if combo in ("rc_eng", "rd_eng"):
            keys["engines"] = rng.choice([2, 3, 4])
            frames = rng.randint(6, 12)
        if rng.random() < 0.2:
            keys["force_intra"] = 1
        if rng.random() < 0.15 and mode != "engines":
            keys["intra_period"] = rng.choice([2, 3])
        elif rng.random() < 0.12 and mode != "rc":         # a new scene inside the clip (the in-frame scene-change detection, hmr_motion_inter.c:3791)
            frames += rng.randint(2, 4)
            keys["cut_at"] = rng.randint(2, frames - 2)
        return width, height, frames, rng.randint(1, 10 ** 6), keys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-ctus", type=int, default=60)
    ap.add_argument("--max-cols", type=int, default=14, help="CTU columns of the largest picture")
    ap.add_argument("--max-rows", type=int, default=9)
    ap.add_argument("--threads-only", action="store_true", help="only cases with several WPP threads (with HENC_WIPE_WORK=<byte> in the environment the checker build then "
                    "keeps of a thread's working memory only what the device's row state carries: oracle/enc_cpu.cpp frame_ctus_lockstep)")
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--chain-sets", type=int, default=0, help="with --gpu: cases with several engines go through hmr_gpu_enc_encode_chain with this many objects per engine (0: hmr_gpu_enc_encode)")
    ap.add_argument("--extra-keys", action="store_true", help="also draw motion_estimation_precision (me) and chroma_qp_offset (cqo)")
    ap.add_argument("--combos", action="store_true", help="only rate control / RD_FULL with one WPP thread or with several engines (accepted since round 6)")
    ap.add_argument("--engines-only", action="store_true", help="only cases with several engines and several WPP threads")
    ap.add_argument("--batch", type=int, default=1, help="with --gpu: this many cases per hmr_gpu_enc_encode_batch call (cases the batch call does not take are left out)")
    ap.add_argument("--tolerate-q12", action="store_true", help="do not count a differing case that had evaluations on a stale prediction window (the documented exception) in the exit code")
    ap.add_argument("--serial-batch", action="store_true", help="with --gpu --batch N: only cases with one WPP thread and one engine, created with hmr_gpu_enc_create_serial_pool and encoded N per batch call "
                                                                "(the reference's single-thread order as a batch schedule)")
    ap.add_argument("--decode", action="store_true", help="also decode the reference's stream with oracle/hevcdec and compare the pictures with the reference's own reconstruction")
    ap.add_argument("specs", nargs="*")
    a = ap.parse_args()
    global DECODE, SERIAL_POOL
    DECODE = a.decode
    SERIAL_POOL = a.serial_batch
    if DECODE:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "hevcdec")])
    dec_counts = {}
    ctx = None
    if a.gpu:
        import libs
        lib = libs.load_gpu()
        lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
        lib.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
        lib.hmr_gpu_enc_encode.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int, C.c_char_p, C.c_long, C.POINTER(C.c_long), C.c_char_p]
        lib.hmr_gpu_enc_destroy.argtypes = [C.c_void_p]
        lib.hmr_gpu_enc_stale_predictions.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        lib.hmr_gpu_last_error.restype = C.c_char_p
        ctx = C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0, lib.hmr_gpu_last_error()
    else:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), CPU_SO])
        lib = load_checker()
    cases = []
    for spec in a.specs:
        parts = spec.split(":")
        w, h, frames = (int(v) for v in parts[0].split("x"))
        keys = dict(kv.split("=") for kv in parts[2].split(",")) if len(parts) > 2 and parts[2] else {}
        cases.append((w, h, frames, int(parts[1]) if len(parts) > 1 and parts[1] else 1234, {k: int(v) for k, v in keys.items()}))
    if not cases:
        rng = random.Random(a.seed)
        cases = [random_case(rng, a.max_ctus, a.gpu, a.max_cols, a.max_rows, a.threads_only, a.engines_only, a.extra_keys, a.combos) for _ in range(a.cases)]
    bad = q12_bad = 0
    batched = {}
    if a.gpu and a.batch > 1:       # groups of cases the batch call takes (one thread per CTU row or the in-between counts, one engine) share their launches
        if a.serial_batch:
            lib.hmr_gpu_enc_create_serial_pool.argtypes = lib.hmr_gpu_enc_create.argtypes
            for c in cases:      # (every drawn case becomes a one-thread, one-engine case; RD_FULL under rate control is refused by every entry)
                c[4].pop("wpp", None); c[4].pop("engines", None)
                if int(c[4].get("rd", 0)) == 1 and int(c[4].get("bitrate_mode", 0)):
                    c[4].pop("bitrate_mode"); c[4].pop("bitrate", None)
        ok_cases = [c for c in cases if a.serial_batch or (int(c[4].get("wpp", 1)) > 1 and int(c[4].get("engines", 1)) == 1)]
        for k in range(0, len(ok_cases), a.batch):
            group = ok_cases[k:k + a.batch]
            units, stale = device_batch(lib, group)
            for c, u, st in zip(group, units, stale):
                batched[id(c)] = (u, st)
        cases = ok_cases
    global STALE
    varying = 0
    for case in cases:
        w, h, frames, clip_seed, keys = case
        spec = f"{w}x{h}x{frames}:{clip_seed}:" + ",".join(f"{k}={v}" for k, v in keys.items())
        if id(case) in batched:
            mine, STALE = batched[id(case)]
        elif a.gpu and a.chain_sets and int(keys.get("engines", 1)) > 1 and int(keys.get("wpp", 1)) > 1 and not int(keys.get("force_intra", 0)):
            mine = device_chain(lib, w, h, frames, clip_seed, keys, a.chain_sets)
            if isinstance(mine, str):
                print(spec, "CHAIN CALL REFUSED:", mine, flush=True)
                continue
        else:
            mine = device(lib, ctx, w, h, frames, clip_seed, keys) if a.gpu else checker(lib, w, h, frames, clip_seed, keys)
        if mine is None:
            print(spec, "REFUSED by the encoder" + (": " + lib.hmr_gpu_last_error().decode() if a.gpu else ""), flush=True)
            continue
        ref = reference(w, h, frames, clip_seed, keys)
        if isinstance(ref, str):
            print(spec, "THE REFERENCE FAILED:", ref, flush=True)
            bad += 1
            continue
        units, mine = mine, b"".join(mine)
        ok = ref == mine
        if not ok and int(keys.get("engines", 1)) > 1:
            # the reference's own stream is not the same on every host with several engines and IDR pictures (seen on the 256-core GPU box: one VPS + SPS pair fewer
            # than in the build container, where it gave the device's bytes): run it again before calling a difference
            again = [reference(w, h, frames, clip_seed, keys) for _ in range(2)]
            if any(isinstance(r, bytes) and r == mine for r in again) or any(isinstance(r, bytes) and r != ref for r in again):
                varies = sum(1 for r in [ref] + again if isinstance(r, bytes) and r == mine)
                print(spec, len(ref), "bytes", hashlib.md5(ref).hexdigest(), f"THE REFERENCE VARIES from run to run ({varies} of 3 runs gave the encoder's {len(mine)} bytes)", flush=True)
                varying += 1
                continue
        bad += not ok
        stale = STALE
        q12 = f" [{stale} evaluations on a stale prediction window: quirk Q12]" if stale > 0 else ""
        dv = ""
        if DECODE and LAST_RECON is not None:
            verdict = decoder_verdict(ref, LAST_RECON, w, h, keys, stale)
            dec_counts[verdict.split(" ")[0].rstrip(":")] = dec_counts.get(verdict.split(" ")[0].rstrip(":"), 0) + 1
            dv = "; decoded: " + verdict
            bad += verdict.startswith("UNEXPLAINED")
        print(spec, len(ref), "bytes", hashlib.md5(ref).hexdigest(), (("IDENTICAL" + q12) if ok else f"DIFFERENT (mine: {len(mine)} bytes){q12}") + dv, flush=True)
        if not ok and stale > 0:
            q12_bad += 1
        if not ok:          # which access unit: the reference's stream cut at the lengths of mine
            o = 0
            for f, u in enumerate(units):
                print(f"    frame {f}: {len(u)} bytes", "same" if ref[o:o + len(u)] == u else "DIFFERS from the reference's bytes at this offset", flush=True)
                o += len(u)
            if a.gpu and os.path.exists(CPU_SO):        # the checker build on the same case (it travels to the GPU box as test infrastructure)
                cl = load_checker()
                cu = checker(cl, w, h, frames, clip_seed, keys)
                print("    checker build:", "identical to the reference" if cu is not None and b"".join(cu) == ref else "differs too", [len(u) for u in cu or []], flush=True)
    if DECODE:
        print("decoder-side check of the reference's streams:", dec_counts)
    if q12_bad:
        print(f"{q12_bad} of the {bad} differing cases had evaluations on a stale prediction window (include/homer_gpu.h: hmr_gpu_enc_stale_predictions)")
    return bad - q12_bad if a.tolerate_q12 else bad


if __name__ == "__main__":
    sys.exit(main())
