#!/usr/bin/env python3
"""Build-container tool: encode the synthetic clip with the compiled reference (oracle/_ref/ref_lockstep) and with the checker
build of the frame encoder (oracle/libenc_cpu.so, free running: its own pictures, filters, SAO decision and entropy coding) and
compare the .265 streams and the reconstructed pictures frame by frame.

usage: tools/stream_diff.py --width 416 --height 240 --frames 4 [key=value ...]
"""
import argparse
import ctypes as C
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import encoder_cases as ec  # noqa: E402
import gen_yuv  # noqa: E402


def split_nals(b):
    """Annex-B byte string -> list of NAL payloads (start codes removed)"""
    out, i, start = [], 0, None
    while i + 3 <= len(b):
        if b[i:i + 3] == b"\x00\x00\x01":
            if start is not None:
                end = i - 1 if i > 0 and b[i - 1] == 0 else i
                out.append(b[start:end])
            start = i + 3
            i += 3
        else:
            i += 1
    if start is not None:
        out.append(b[start:])
    return out


def encode_cpu(width, height, frames, keys, force_intra=False, sched=0, row_guess=0):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libenc_cpu.so"))
    lib.henc_cpu_create.restype = C.c_void_p
    lib.henc_cpu_create.argtypes = [C.POINTER(ec.EncCfg)]
    lib.henc_cpu_encode_frame.restype = C.c_long
    lib.henc_cpu_encode_frame.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int, C.c_char_p, C.c_long, C.c_char_p]
    cfg = ec.default_cfg(width, height, **keys)
    h = lib.henc_cpu_create(C.byref(cfg))
    assert h
    lib.henc_cpu_set_sched.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.henc_cpu_set_sched(h, sched, row_guess)
    lib.henc_cpu_sched_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    buf = C.create_string_buffer(8 << 20)
    rec = C.create_string_buffer(width * height * 3 // 2)
    streams, recons = [], []
    for planes in ec.clip_frames(width, height, frames):
        n = lib.henc_cpu_encode_frame(h, *planes, 3 if force_intra else 0, buf, len(buf), rec)
        assert n > 0
        streams.append(buf.raw[:n])
        if sched:
            st = (C.c_int * 3)()
            lib.henc_cpu_sched_stats(h, st, 1)
            print(f"  schedule: {st[0]} passes, {st[1]} CTU encodes, {st[2]} CTUs failed the first check")
        recons.append(rec.raw)
    return streams, recons


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--force-intra", action="store_true")
    ap.add_argument("--sched", type=int, default=0, help="1 = the device's row-parallel schedule (guesses + verification), emulated")
    ap.add_argument("--row-guess", type=int, default=0)
    ap.add_argument("keys", nargs="*")
    a = ap.parse_args()
    keys = dict(k.split("=") for k in a.keys)
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "in.yuv")
        gen_yuv.write_clip(yuv, a.width, a.height, a.frames)
        cmd = [os.path.join(ROOT, "oracle", "_ref", "ref_lockstep"), yuv, os.path.join(tmp, "out.265"), str(a.width), str(a.height), str(a.frames),
               "recon=" + os.path.join(tmp, "rec.yuv")] + [f"{k}={v}" for k, v in keys.items()] + (["force_intra=1"] if a.force_intra else [])
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        ref_stream = open(os.path.join(tmp, "out.265"), "rb").read()
        ref_rec = open(os.path.join(tmp, "rec.yuv"), "rb").read()
    streams, recons = encode_cpu(a.width, a.height, a.frames, keys, a.force_intra, a.sched, a.row_guess)
    mine = b"".join(streams)
    fsz = a.width * a.height * 3 // 2
    ok = True
    for f in range(a.frames):
        same = recons[f] == ref_rec[f * fsz:(f + 1) * fsz]
        ok &= same
        if not same:
            import numpy as np
            d = np.flatnonzero(np.frombuffer(recons[f], np.uint8) != np.frombuffer(ref_rec[f * fsz:(f + 1) * fsz], np.uint8))
            print(f"frame {f}: reconstruction differs at {len(d)} samples, first {d[:5]}")
        else:
            print(f"frame {f}: reconstruction identical, {len(streams[f])} stream bytes")
    rn, mn = split_nals(ref_stream), split_nals(mine)
    print("NAL sizes ref :", [len(x) for x in rn])
    print("NAL sizes mine:", [len(x) for x in mn])
    for k, (x, y) in enumerate(zip(rn, mn)):
        if x != y:
            first = next((i for i in range(min(len(x), len(y))) if x[i] != y[i]), min(len(x), len(y)))
            print(f"NAL {k} (type {(x[0] >> 1) & 63}) differs at byte {first}: ref {x[first:first + 8].hex()} mine {y[first:first + 8].hex()}")
            ok = False
            break
    print("stream md5 ref ", hashlib.md5(ref_stream).hexdigest())
    print("stream md5 mine", hashlib.md5(mine).hexdigest())
    ok &= ref_stream == mine
    print("IDENTICAL" if ok else "DIFFERENT")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
