#!/usr/bin/env python3
"""One-off stream-swap fuzzing beyond the fixed test cases: other clip seeds / sizes / configurations.
usage: python tools/swap_fuzz.py WxHxFRAMES:seed[:key=value,...] ..."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_yuv  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_lockstep")
SWAP = os.path.join(ROOT, "oracle", "_ref", "ref_swap")


def encode(exe, clip, out, w, h, frames, extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe, clip, out, str(w), str(h), str(frames), *extra], capture_output=True, text=True, env=e, timeout=3000)
    assert r.returncode == 0, r.stderr[-1000:]
    return open(out, "rb").read()


def main():
    bad = 0
    for spec in sys.argv[1:]:
        parts = spec.split(":")
        w, h, frames = (int(v) for v in parts[0].split("x"))
        seed = int(parts[1]) if len(parts) > 1 else 1234
        extra = parts[2].split(",") if len(parts) > 2 and parts[2] else []
        with tempfile.TemporaryDirectory() as td:
            clip = os.path.join(td, "c.yuv")
            gen_yuv.write_clip(clip, w, h, frames, seed)
            ref = encode(REF, clip, os.path.join(td, "r.265"), w, h, frames, extra)
            gpu = encode(SWAP, clip, os.path.join(td, "g.265"), w, h, frames, extra, {"HOMER_SWAP": "all"})
        ok = ref == gpu
        bad += not ok
        import hashlib
        print(spec, len(ref), "bytes", "md5", hashlib.md5(gpu).hexdigest(), "IDENTICAL" if ok else "DIFFERENT", flush=True)
    sys.exit(1 if bad else 0)


main()
