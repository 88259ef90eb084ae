#!/usr/bin/env python3
"""Wall-clock of the reference encoder with the GPU kernels installed through the per-call drop-in entries, with and without the luma / chroma CU drivers routed as one
submission each: what taking the host round trips out of a decision is worth in the integration (not a throughput figure: every drop-in call is synchronous).
usage: python tools/swap_time.py W H FRAMES [key=value ...]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_yuv  # noqa: E402

SWAP = os.path.join(ROOT, "oracle", "_ref", "ref_swap")
w, h, frames = (int(v) for v in sys.argv[1:4])
extra = sys.argv[4:]
with tempfile.TemporaryDirectory() as td:
    clip = os.path.join(td, "c.yuv")
    gen_yuv.write_clip(clip, w, h, frames)
    out = {}
    for mode in ("all", "all,-intra_luma_cu,-intra_chroma_cu,-inter_cu", "none"):
        e = dict(os.environ, HOMER_SWAP=mode)
        t0 = time.perf_counter()
        r = subprocess.run([SWAP, clip, os.path.join(td, "o.265"), str(w), str(h), str(frames), *extra], capture_output=True, text=True, env=e, timeout=3000)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-500:]
        out[mode] = (dt, open(os.path.join(td, "o.265"), "rb").read())
        print(f"HOMER_SWAP={mode:36s} {dt:8.2f} s", flush=True)
    assert out["all"][1] == out["none"][1] == out["all,-intra_luma_cu,-intra_chroma_cu,-inter_cu"][1], "streams differ"
    print("streams identical")
