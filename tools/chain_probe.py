#!/usr/bin/env python3
"""GPU box tool: a stream fixture with num_enc_engines = E encoded in chains of 1 .. E overlapping frames (hmr_gpu_enc_encode_chain); reports frames/s and, where a
chain length gives a different stream, the first access unit that differs.  usage: tools/chain_probe.py [case ...]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402
import test_gpu_chain as t  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from stream_diff import split_nals  # noqa: E402

if os.environ.get("CHAIN_DUMP"):
    import torch
    torch.zeros(1, device="cuda:0")
lib = libs.load_gpu()
for case in sys.argv[1:] or ["416x240_eng3_wpp_rows"]:
    E = t.GOLD[case]["keys"]["engines"]
    base = None
    for chain in ([int(os.environ["CHAIN_ONLY"])] if os.environ.get("CHAIN_ONLY") else range(1, E + 1)):
        stream, g = t.encode_chained(lib, case, chain)
        ok = hashlib.md5(stream).hexdigest() == g["stream_md5"]
        nals = [n for n in split_nals(stream) if ((n[0] >> 1) & 63) < 32]
        if base is None:
            base = nals
        bad = next((k for k, (a, b) in enumerate(zip(base, nals)) if a != b), None)
        print(case, "chain", chain, "identical to the reference" if ok else f"DIFFERENT, first differing frame {bad}, sizes {[len(n) for n in nals]} vs {[len(n) for n in base]}")
    if os.environ.get("CHAIN_DUMP") and len(t.DUMPS) > 1:
        import numpy as np
        keys = sorted(t.DUMPS)
        g = t.GOLD[case]
        w, h = g["width"], g["height"]
        a, b = t.DUMPS[keys[0]], t.DUMPS[keys[-1]]
        for f in sorted(a):
            ya, yb = a[f][:w * h].reshape(h, w), b[f][:w * h].reshape(h, w)
            d = np.argwhere(ya != yb)
            if len(d):
                ctus = sorted({(int(y) // 64, int(x) // 64) for y, x in d})
                print(f"frame {f}: {len(d)} luma samples differ between chain lengths {keys[0]} and {keys[-1]}; CTUs (row, col): {ctus[:40]}; first sample {d[0]}")
                break
        else:
            print("no picture differs")
