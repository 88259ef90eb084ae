#!/usr/bin/env python3
"""Build container: every configuration of tests/golden/make_stream_golden.py (or a subset by name part) through the compiled reference encoder with its reconstruction dumped,
the stream through oracle/hevcdec, decoded pictures against the reference's reconstruction.    usage: tools/decoder_sweep.py [name-part ...] [--max-pixels N]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_yuv  # noqa: E402
import make_stream_golden as msg  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
max_pixels = int(sys.argv[sys.argv.index("--max-pixels") + 1]) if "--max-pixels" in sys.argv else 1 << 62
args = [a for a in args if not a.isdigit()]
bad = 0
for name, w, h, frames, keys in msg.CASES:
    if args and not any(a in name for a in args):
        continue
    if w * h > max_pixels:
        continue
    keys = dict(keys)
    cut_at, clip_seed = keys.pop("cut_at", None), keys.pop("clip_seed", None)
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "in.yuv")
        gen_yuv.write_clip(yuv, w, h, frames, seed=clip_seed or 1234, cut_at=cut_at)
        turnstile = int(keys.get("wpp", 1)) > 1 or int(keys.get("engines", 1)) > 1
        cmd = [os.path.join(ROOT, "oracle", "_ref", "ref_ctudump" if turnstile else "ref_lockstep"), yuv, os.path.join(tmp, "out.265"), str(w), str(h), str(frames),
               "recon=" + os.path.join(tmp, "rec.yuv")] + [f"{k}={v}" for k, v in keys.items()]
        subprocess.run(cmd, check=True, timeout=1800, stdout=subprocess.DEVNULL, env=dict(os.environ, HOMER_TURNSTILE="1") if turnstile else None)
        t1 = time.time()
        r = subprocess.run([os.path.join(ROOT, "oracle", "hevcdec"), os.path.join(tmp, "out.265"), os.path.join(tmp, "dec.yuv")], capture_output=True, text=True)
        t2 = time.time()
        if r.returncode:
            print(f"{name}: DECODER rc={r.returncode} {r.stderr.strip()}")
            bad += 1
            continue
        rec, dec = open(os.path.join(tmp, "rec.yuv"), "rb").read(), open(os.path.join(tmp, "dec.yuv"), "rb").read()
        fsz = w * h * 3 // 2
        same = [hashlib.md5(rec[i * fsz:(i + 1) * fsz]).hexdigest() == hashlib.md5(dec[i * fsz:(i + 1) * fsz]).hexdigest() for i in range(frames)]
        ok = len(dec) == len(rec) and all(same)
        bad += not ok
        print(f"{name}: {'identical' if ok else 'DIFFERENT ' + str(same)}  ({frames} pictures; encode {t1 - t0:.1f} s, decode {t2 - t1:.1f} s)  {r.stdout.strip()}", flush=True)
print("different or failed:", bad)
sys.exit(1 if bad else 0)
