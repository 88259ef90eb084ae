#!/usr/bin/env python3
"""Random differential runs for the composite functions - intra mode search, intra / inter / plain TU chains, the luma / chroma intra CU drivers, the SAO offset derivation - beyond the fixed case lists
of tests/kernel_cases.py: the CPU oracle against the compiled reference (build container: needs oracle/_ref), or, with --gpu, the drop-in
entries of libhomer_gpu.so against the oracle (GPU box).
usage: python tools/oracle_fuzz.py [--gpu] [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kernel_cases as kc  # noqa: E402
import libs  # noqa: E402


def neighbours(r, n):
    left, top = (int(r.integers(0, 2)), int(r.integers(0, 2))) if r.random() < 0.3 else (1, 1)
    bl, tr = int(r.integers(0, 2)) & left, int(r.integers(0, 2)) & top
    return dict(left=left, top=top, bl=bl, tr=tr, bl_size=(n if r.random() < 0.7 else max(n // 2, 4)) if bl else 0,
                tr_size=(n if r.random() < 0.7 else max(n // 2, 4)) if tr else 0, strong=int(r.integers(0, 2)))


def random_case(r):
    kind = r.choice(["intra_search", "intra_tu_chain", "inter_tu_chain", "tu_chain", "intra_luma_cu", "intra_chroma_cu", "sao_offsets"])
    if kind == "intra_search":
        n = int(r.choice([4, 8, 16, 32, 64]))
        p = dict(n=n, **neighbours(r, n), left_mode=int(r.integers(-1, 35)), top_mode=int(r.integers(-1, 35)), rd_mode=int(r.choice([2, 0])),
                 sqrt_lambda=float(r.uniform(0.5, 80)), theta=float(r.uniform(0, np.pi)), period=float(r.uniform(2, 30)), amp=float(r.uniform(0, 100)),
                 tilt=float(r.uniform(-8, 8)), noise=int(r.integers(0, 12)))
    elif kind == "intra_luma_cu":
        n = int(r.choice([8, 16, 32, 64]))
        flags = []
        for node in range(5):     # what a frame can produce: below-left needs left, above-right needs above; quadrant 1 has no below-left (quadrant 2 is coded
            fl, ft = (int(r.integers(0, 2)), int(r.integers(0, 2))) if r.random() < 0.3 else (1, 1)     # after it), quadrant 3 has neither
            flags.append((fl, ft, int(r.integers(0, 2)) & fl & (node not in (2, 4)), int(r.integers(0, 2)) & ft & (node != 4)))
        p = dict(n=n, flags=flags, pict_w=int(r.choice([n, n + n // 2, 2 * n, 4 * n])), pict_h=int(r.choice([n, n + n // 2, 2 * n, 4 * n])), qp=int(r.integers(10, 50)),
                 sqrt_lambda=float(r.uniform(0.5, 80)), rd_mode=int(r.choice([2, 2, 0])), slice_i=int(r.integers(0, 2)), sbh=int(r.integers(0, 2)),
                 strong=int(r.integers(0, 2)), theta=float(r.uniform(0, np.pi)), period=float(r.uniform(2, 30)), amp=float(r.uniform(0, 100)),
                 tilt=float(r.uniform(-8, 8)), noise=int(r.choice([0, 1, 3, 8, 20, 40])))
    elif kind == "intra_chroma_cu":
        n = int(r.choice([4, 8, 16, 32]))
        flags = []
        for node in range(5):
            fl, ft = (int(r.integers(0, 2)), int(r.integers(0, 2))) if r.random() < 0.3 else (1, 1)
            flags.append((fl, ft, int(r.integers(0, 2)) & fl & (node not in (2, 4)), int(r.integers(0, 2)) & ft & (node != 4)))
        p = dict(n=n, flags=flags, pict_w=int(r.choice([n, n + n // 2, 2 * n, 4 * n])), pict_h=int(r.choice([n, n + n // 2, 2 * n, 4 * n])),
                 luma_mode=int(r.integers(0, 35)), split=int(r.integers(0, 2)) if n < 32 else 1, qp=int(r.integers(10, 50)), slice_qp=int(r.integers(10, 50)),
                 sqrt_lambda=float(r.uniform(0.5, 80)), rd_mode=int(r.choice([2, 0])), slice_i=int(r.integers(0, 2)), sbh=int(r.integers(0, 2)),
                 theta=float(r.uniform(0, np.pi)), period=float(r.uniform(2, 30)), amp=float(r.uniform(0, 100)), tilt=float(r.uniform(-8, 8)),
                 noise=int(r.choice([0, 1, 3, 8, 20, 40])))
    elif kind == "sao_offsets":
        p = dict(conc=float(r.choice([0.1, 0.5, 2.0, 10.0])), empty=float(r.choice([0.0, 0.3, 0.8])), bias=float(r.choice([0.05, 0.3, 1.0, 3.0, 8.0, 20.0])),
                 chroma_ratio=float(r.uniform(0.3, 1.5)), **{"lambda": float(10 ** r.uniform(-1, 3.2))})
    elif kind == "intra_tu_chain":
        n = int(r.choice([4, 8, 16, 32]))
        p = dict(n=n, comp=int(r.choice([0, 0, 1, 2])) if n < 32 else 0, **neighbours(r, n), mode=int(r.integers(0, 35)), scan=int(r.integers(1, 4)),
                 slice_i=int(r.integers(0, 2)), sbh=int(r.integers(0, 2)), per=int(r.integers(0, 9)), rem=int(r.integers(0, 6)), theta=float(r.uniform(0, np.pi)),
                 period=float(r.uniform(2, 25)), amp=float(r.uniform(0, 100)), noise=int(r.choice([0, 1, 4, 12, 40])))
    elif kind == "inter_tu_chain":
        n = int(r.choice([4, 8, 16, 32]))
        p = dict(n=n, comp=int(r.choice([0, 0, 1, 2])) if n < 32 else 0, scan=3, slice_i=0, sbh=int(r.integers(0, 2)), per=int(r.integers(0, 9)),
                 rem=int(r.integers(0, 6)), weight=float(2.0 ** (r.integers(-3, 6) / 3.0)), thr=float(np.clip(r.uniform(0, 8000) / 2.5 - 5.0, 1.0, 20000.0)),
                 theta=float(r.uniform(0, np.pi)), period=float(r.uniform(2, 20)), amp=float(r.choice([0, 1, 3, 8, 25, 80, 250])), noise=int(r.choice([0, 1, 3, 10, 30])))
    else:
        n = int(r.choice([4, 8, 16, 32]))
        comp, intra = (int(r.choice([0, 1, 2])) if n < 32 else 0), int(r.integers(0, 2))
        p = dict(n=n, comp=comp, intra=intra, slice_i=intra | int(r.integers(0, 2)), sbh=int(r.integers(0, 2)), per=int(r.integers(0, 9)), rem=int(r.integers(0, 6)),
                 scan=int(r.integers(1, 4)), noise=int(r.choice([1, 3, 10, 40, 255])), dst=1 if (n == 4 and intra and comp == 0) else 0)
    return (str(kind), p, int(r.integers(1, 1 << 30)))


def main():
    args = [a for a in sys.argv[1:] if a != "--gpu"]
    gpu = "--gpu" in sys.argv[1:]
    seconds = float(args[0]) if args else 60.0
    r = np.random.default_rng(int(args[1]) if len(args) > 1 else 20261002)
    ora = libs.load_oracle()
    ref, prefix = (libs.load_gpu(), "hmr_gpu_") if gpu else (libs.load_ref(), "refh_")
    if ref is None:
        sys.exit("oracle/_ref/libhomer_ref.so is not built here")
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < seconds:
        case = random_case(r)
        a, b = kc.run(ora, "ora_", case), kc.run(ref, prefix, case)
        n += 1
        for k in (k for k in a if k in b):
            if not np.array_equal(a[k], b[k]):
                bad += 1
                print("MISMATCH", case, k)
                break
    print(f"{n} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


main()
