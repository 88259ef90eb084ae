"""Replay of tests/golden/trace_200x136.npz: inputs and outputs of the reference's block drivers logged during a real encode
(tests/golden/make_trace.py).  `oracle_outputs` runs one record through the CPU oracle's flat functions; the GPU test batches whole groups."""
import ctypes as C
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    with open(os.path.join(GOLDEN, "trace_200x136.json")) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN, "trace_200x136.npz"))
    groups = []
    for g in meta["groups"]:
        t = g["tag"]
        blobs = []
        while f"{t}_blob{len(blobs)}" in data:
            blobs.append(data[f"{t}_blob{len(blobs)}"])
        groups.append({**g, "hdr": data[t + "_hdr"], "dbl": data[t + "_dbl"], "blobs": blobs})
    return groups


def lshape_tile(d, n):
    """(2n+1)^2 tile whose first row / column hold the logged neighbours; returns the tile (corner at [0, 0])."""
    s = 2 * n + 1
    tile = np.zeros((s, s), np.int16)
    tile[0, :] = d[:s]
    tile[1:, 0] = d[s:]
    return tile


def vp(a, off=0):
    return C.c_void_p(a.ctypes.data + off * a.itemsize)


def ref_planes(groups):
    """padded reference pictures logged by the trace, by id"""
    out = {}
    for g in groups:
        if g["kind"] == "ref_plane":
            for i in range(g["count"]):
                out[int(g["hdr"][i][0])] = np.ascontiguousarray(g["blobs"][0][i])
    return out


def oracle_outputs(ora, g, i, planes=None):
    """-> dict of outputs for record i of group g, same keys as `expected(g, i)`."""
    h, d, b = g["hdr"][i], g["dbl"][i], g["blobs"]
    kind = g["kind"]
    if kind == "inter_tu":
        n = int(h[0])
        res, pred = np.ascontiguousarray(b[0][i]), np.ascontiguousarray(b[1][i])
        lev, rec, ac = np.zeros(n * n, np.int16), np.zeros(n * n, np.int16), C.c_int(0)
        ora.ora_inter_tu_chain.restype = C.c_uint32
        r = ora.ora_inter_tu_chain(vp(res), n, vp(pred), n, vp(lev), vp(rec), n, n, int(h[2]), int(h[1]), int(h[3]), int(h[4]), int(h[5]), int(h[6]),
                                   C.c_double(d[0]), C.c_double(d[1]), C.byref(ac))
        return {"levels": lev, "recon": rec, "sum": ac.value, "ret": int(np.uint32(r).astype(np.int32)) if False else int(np.int32(np.uint32(r)))}
    if kind == "intra_tu":
        n = int(h[0])
        orig, tile = np.ascontiguousarray(b[0][i]), lshape_tile(b[1][i], n)
        pred, lev, rec, ac = np.zeros(n * n, np.int16), np.zeros(n * n, np.int16), np.zeros(n * n, np.int16), C.c_int(0)
        ora.ora_intra_tu_chain.restype = C.c_uint32
        r = ora.ora_intra_tu_chain(vp(orig), n, vp(tile), 2 * n + 1, *[int(v) for v in h[1:10]], 1, vp(pred), n, vp(lev), vp(rec), n, n, int(n == 4), int(h[10]), 0,
                                   int(h[11]), int(h[12]), int(h[13]), int(h[14]), C.byref(ac))
        return {"pred": pred, "levels": lev, "recon": rec, "sum": ac.value, "ret": int(np.int32(np.uint32(r)))}
    if kind == "intra_search":
        n = int(h[0])
        orig, tile = np.ascontiguousarray(b[0][i]), lshape_tile(b[1][i], n)
        adi, adif, pred = np.zeros(4 * n + 1, np.int16), np.zeros(4 * n + 1, np.int16), np.zeros(n * n, np.int16)
        out, cost = np.zeros(2, np.int32), C.c_double(0)
        preds, bits = np.ascontiguousarray(h[8:11], np.int32), np.ascontiguousarray(h[11:14], np.int32)
        ora.ora_intra_search(vp(orig), n, vp(tile), 2 * n + 1, n, *[int(v) for v in h[1:8]], vp(preds), vp(bits), int(h[14]), C.c_double(d[0]), vp(adi), vp(adif),
                             vp(pred), n, vp(out), C.byref(cost))
        return {"adi": adi, "adif": adif, "pred": pred, "best": int(out[0]), "bits": int(out[1]), "cost": cost.value}
    if kind == "me":
        n, pid, gx, gy, ix, iy, rx, ry, fw, fh, action, na, ns = (int(v) for v in h[:13])
        plane = planes[pid]
        st = fw + 160
        orig = np.ascontiguousarray(b[0][i])
        amvp, search, out = np.ascontiguousarray(h[13:17], np.int32), np.ascontiguousarray(h[17:27], np.int32), np.zeros(4, np.int32)
        ora.ora_motion_estimation.restype = C.c_uint32
        r = ora.ora_motion_estimation(vp(orig), n, vp(plane, (80 + gy) * st + 80 + gx), st, gx, gy, ix, iy, n, rx, ry, fw, fh, vp(amvp), na, vp(search), ns,
                                      C.c_double(d[0]), action, vp(out))
        return {"mv": out.copy(), "ret": int(np.int32(np.uint32(r)))}
    luma, w, hh, fx, fy, bi = (int(v) for v in h[:6])
    win = np.ascontiguousarray(b[0][i]).reshape(hh + 8, w + 8)
    pred = np.zeros(hh * w, np.int16)
    if luma:
        ora.ora_mc_luma(vp(win, 4 * (w + 8) + 4), w + 8, vp(pred), w, w, hh, fx, fy, bi)
    else:
        ora.ora_mc_chroma(vp(win, 4 * (w + 8) + 4), w + 8, vp(pred), w, w, fx, fy, bi)
    return {"pred": pred}


def expected(g, i):
    h, d, b = g["hdr"][i], g["dbl"][i], g["blobs"]
    kind = g["kind"]
    if kind == "inter_tu":
        return {"levels": b[2][i], "recon": b[3][i], "sum": int(h[7]), "ret": int(h[8])}
    if kind == "intra_tu":
        return {"pred": b[2][i], "levels": b[3][i], "recon": b[4][i], "sum": int(h[15]), "ret": int(h[16])}
    if kind == "intra_search":
        return {"adi": b[2][i], "adif": b[3][i], "pred": b[4][i], "best": int(h[15]), "bits": int(h[16]), "cost": float(d[1])}
    if kind == "me":
        return {"mv": h[27:31], "ret": int(h[31])}
    return {"pred": b[1][i]}
