"""Motion cases: motion compensation (a17) and the search driver with fused sub-pel refinement (a15/a16).

Same three-prefix scheme as kernel_cases.py.  The reference entry (refh_motion_estimation) takes (qp, avg_dist) and
derives the cost factor itself; the oracle and the GPU take the host-side double `corr` = qp * clip(avg_dist/2000, .15, 1.4)
(calc_mv_correction, hmr_common.h:53) - the same IEEE-754 double operations in Python.
"""
import ctypes as C

import numpy as np

from kernel_cases import aligned, aligned_copy, ptr

FW, FH, PAD = 320, 192, 80
STRIDE = FW + 2 * PAD


def frames(rng, shift):
    yy, xx = np.mgrid[0:FH + 2 * PAD, 0:FW + 2 * PAD]
    base = 128 + 50 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 30 * np.sin((xx + 2 * yy) / 13.0)
    ref = np.clip(base + rng.integers(-6, 7, base.shape), 0, 255).astype(np.int16)
    cur = np.roll(np.roll(ref, shift[1], axis=0), shift[0], axis=1) + rng.integers(-3, 4, base.shape)
    return aligned_copy(np.clip(cur, 0, 255).astype(np.int16)), aligned_copy(ref)


def me_case(seed):
    rng = np.random.default_rng(seed)
    p = {"shift": (int(rng.integers(-9, 10)), int(rng.integers(-6, 7))), "size": int(rng.choice([8, 16, 32, 64]))}
    s = p["size"]
    p["gx"] = int(rng.integers(0, (FW - s) // 8 + 1)) * 8
    p["gy"] = int(rng.integers(0, (FH - s) // 8 + 1)) * 8
    p["n_amvp"] = int(rng.integers(1, 3))
    p["amvp"] = [int(v) for v in rng.integers(-40, 41, 4)]
    p["n_search"] = int(rng.integers(0, 5))
    p["search"] = [int(v) for v in rng.integers(-48, 49, 10)]
    p["qp"] = int(rng.integers(22, 40))
    p["avg_dist"] = float(rng.uniform(100, 4000))
    p["init"] = (int(rng.integers(-3, 4)), int(rng.integers(-3, 4)))
    p["action"] = int(rng.choice([7, 7, 7, 3, 1, 6]))
    return ("motion_estimation", p, seed)


def run_me(lib, prefix, case):
    _, p, seed = case
    rng = np.random.default_rng(seed + 100000)
    cur, ref = frames(rng, p["shift"])
    s, gx, gy = p["size"], p["gx"], p["gy"]
    ob = aligned((64, 64), np.int16)
    ob[:s, :s] = cur[PAD + gy:PAD + gy + s, PAD + gx:PAD + gx + s]
    off = (PAD + gy) * STRIDE + PAD + gx
    amvp, srch = np.array(p["amvp"], np.int32), np.array(p["search"], np.int32)
    out = np.zeros(4, np.int32)
    f = getattr(lib, prefix + "motion_estimation")
    f.restype = C.c_uint32
    common = [ptr(ob), 64, ptr(ref, off), STRIDE, gx, gy, p["init"][0], p["init"][1], s, 128, 64, FW, FH, ptr(amvp), p["n_amvp"], ptr(srch), p["n_search"]]
    if prefix == "refh_":
        r = f(*common, p["qp"], C.c_double(p["avg_dist"]), p["action"], ptr(out))
    else:
        corr = p["qp"] * min(max(p["avg_dist"] / 2000., .15), 1.4)
        r = f(*common, C.c_double(corr), p["action"], ptr(out))
    return {"mv": out.copy(), "sad": np.array([r], np.uint32)}


def mc_case(seed, luma):
    rng = np.random.default_rng(seed)
    if luma:
        w = int(rng.choice([8, 16, 32, 64]))
        p = {"luma": 1, "w": w, "h": w, "mvx": int(rng.integers(-60, 61)), "mvy": int(rng.integers(-40, 41))}
    else:
        w = int(rng.choice([4, 8, 16, 32]))
        p = {"luma": 0, "w": w, "h": w, "mvx": int(rng.integers(-120, 121)), "mvy": int(rng.integers(-80, 81))}
    p["bi"] = int(rng.random() < 0.25)
    if seed % 7 == 0:
        p["mvx"] &= ~(3 if luma else 7)     # integer / one-dimensional vectors
    if seed % 11 == 0:
        p["mvy"] &= ~(3 if luma else 7)
    return ("mc", p, seed)


def run_mc(lib, prefix, case):
    _, p, seed = case
    rng = np.random.default_rng(seed + 200000)
    ref = aligned((160, 192), np.int16)
    ref[...] = rng.integers(0, 256, ref.shape)
    pred = aligned((64, 64), np.int16)
    pred[...] = 0x1234
    off = 48 * 192 + 56
    if p["luma"]:
        getattr(lib, prefix + "mc_luma")(ptr(ref, off), 192, ptr(pred), 64, p["w"], p["h"], p["mvx"], p["mvy"], p["bi"])
    else:
        getattr(lib, prefix + "mc_chroma")(ptr(ref, off), 192, ptr(pred), 64, p["w"], p["mvx"], p["mvy"], p["bi"])
    return {"pred": pred[:p["h"], :p["w"]].copy()}


def run(lib, prefix, case):
    return run_me(lib, prefix, case) if case[0] == "motion_estimation" else run_mc(lib, prefix, case)


def all_cases(level="full"):
    n_me, n_mc = (400, 300) if level == "full" else (120, 120)
    cases = [me_case(5000 + i) for i in range(n_me)]
    cases += [mc_case(7000 + i, i % 2 == 0) for i in range(n_mc)]
    return cases
