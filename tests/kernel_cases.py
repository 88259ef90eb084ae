"""Shared case generator / caller for the table kernels (K1-K15, K19).

One flat C signature per kernel is exported three times:
  ora_*      oracle/liboracle.so        (this repo's CPU restatement, test infrastructure)
  refh_*     oracle/_ref/libhomer_ref.so (the compiled reference's SSE4.2 symbols; build container only)
  hmr_gpu_*  homerhevc_amd/libhomer_gpu.so (the product: HIP kernels behind the C ABI)
so one case description drives oracle-vs-reference pinning, golden minting and GPU parity.

A case is (kernel, params dict, seed).  `run(lib, prefix, case)` returns a dict of numpy outputs.
Inputs are rebuilt from the seed (numpy default_rng), so golden files hold params + outputs only.
"""
import ctypes as C

import numpy as np

VP = C.c_void_p


def aligned(shape, dtype, align=64):
    """Zero-filled array whose first element is `align`-byte aligned (SSE aligned loads)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    raw = np.zeros(n * dtype.itemsize + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n * dtype.itemsize].view(dtype).reshape(shape)


def aligned_copy(a, align=64):
    out = aligned(a.shape, a.dtype, align)
    out[...] = a
    return out


def ptr(a, off=0):
    return VP(a.ctypes.data + off * a.dtype.itemsize)


def fn(lib, prefix, name, restype=None):
    f = getattr(lib, prefix + name)
    f.restype = restype
    return f


PLANE_H, PLANE_W = 160, 192  # scratch plane: room for 64x64 blocks at any offset + filter margins


def pix_plane(rng, lo=0, hi=256):
    p = aligned((PLANE_H, PLANE_W), np.int16)
    p[...] = rng.integers(lo, hi, (PLANE_H, PLANE_W))
    return p


# ------------------------------------------------------------------ kernels

def k_sad(lib, prefix, p, rng, which="sad"):
    n = p["n"]
    src = pix_plane(rng, *p.get("src_range", (0, 256)))
    pred = pix_plane(rng, *p.get("pred_range", (0, 256)))
    so = 16 * PLANE_W + 16                       # aligned source block
    po = (8 + p.get("dy", 0)) * PLANE_W + 8 + p.get("dx", 0)  # unaligned candidate (ME walks +-1 px)
    ps = 0 if p.get("pred_stride0") else PLANE_W
    f = fn(lib, prefix, which, C.c_uint32)
    r = f(ptr(src, so), C.c_uint32(PLANE_W), ptr(pred, po), C.c_uint32(ps), C.c_int(n))
    return {"value": np.array([r], dtype=np.uint32)}


def k_ssd16b(lib, prefix, p, rng):
    return k_sad(lib, prefix, p, rng, which="ssd16b")


def k_predict(lib, prefix, p, rng):
    n = p["n"]
    orig, pred = pix_plane(rng), pix_plane(rng)
    res = aligned((64, 64), np.int16)
    fn(lib, prefix, "predict")(ptr(orig, 16 * PLANE_W + 16), C.c_int(PLANE_W), ptr(pred, 8 * PLANE_W + 32), C.c_int(PLANE_W),
                               ptr(res), C.c_int(64), C.c_int(n))
    return {"residual": res[:n, :n].copy()}


def k_reconst(lib, prefix, p, rng):
    n = p["n"]
    lo, hi = p.get("res_range", (-300, 301))
    pred = pix_plane(rng)
    res = aligned((64, 64), np.int16)
    res[...] = rng.integers(lo, hi, (64, 64))
    if p.get("extreme"):
        res[::3, ::5] = 32767
        res[1::4, 2::3] = -32768
    dec = aligned((64, 64), np.int16)
    rs = 0 if p.get("res_stride0") else 64
    if rs == 0:
        res[0, :] = 0
    fn(lib, prefix, "reconst")(ptr(pred, 16 * PLANE_W + 16), C.c_int(PLANE_W), ptr(res), C.c_int(rs), ptr(dec), C.c_int(64), C.c_int(n))
    return {"decoded": dec[:n, :n].copy()}


def k_modified_variance(lib, prefix, p, rng):
    src = pix_plane(rng)
    r = fn(lib, prefix, "modified_variance", C.c_uint32)(ptr(src, 16 * PLANE_W + 16), C.c_int(p["n"]), C.c_int(PLANE_W), C.c_int(p["modif"]))
    return {"value": np.array([r], dtype=np.uint32)}


def k_copy(lib, prefix, p, rng):
    kind, h, w = p["kind"], p["h"], p["w"]
    wpad = (w + 15) // 16 * 16
    if kind == "8_16":
        src = aligned((PLANE_H, PLANE_W), np.uint8)
        src[...] = rng.integers(0, 256, src.shape)
        dst = aligned((h, wpad + 16), np.int16)
    elif kind == "16_8":
        src = pix_plane(rng, -64, 400)
        dst = aligned((h, wpad + 16), np.uint8)
    else:
        src = pix_plane(rng, -32768, 32768)
        dst = aligned((h, wpad + 16), np.int16)
    fn(lib, prefix, "copy_" + kind)(ptr(src, 8 * PLANE_W + 16), C.c_uint32(PLANE_W), ptr(dst), C.c_uint32(dst.shape[1]), C.c_int(h), C.c_int(w))
    return {"dst": dst[:, :w].copy()}


def adi_buffer(rng, n, smooth=False):
    adi = aligned((4 * 64 + 1 + 15,), np.int16)
    if smooth:
        base = rng.integers(40, 200)
        adi[:4 * n + 1] = base + (np.arange(4 * n + 1) * rng.integers(-2, 3)) // 8
    else:
        adi[:4 * n + 1] = rng.integers(0, 256, 4 * n + 1)
    return adi


def k_intra_planar(lib, prefix, p, rng):
    n = p["n"]
    adi = adi_buffer(rng, n)
    pred = aligned((64, 64), np.int16)
    pred[...] = 0x1234  # poison: a partial write must show
    fn(lib, prefix, "intra_planar")(ptr(pred), C.c_int(64), ptr(adi), C.c_int(4 * n + 1), C.c_int(n))
    return {"pred": pred[:n, :n].copy()}


def k_intra_angular(lib, prefix, p, rng):
    n = p["n"]
    adi = adi_buffer(rng, n)
    if p.get("flat") is not None:
        adi[:4 * n + 1] = p["flat"]
    pred = aligned((64, 64), np.int16)
    pred[...] = 0x1234
    fn(lib, prefix, "intra_angular")(ptr(pred), C.c_int(64), ptr(adi), C.c_int(4 * n + 1), C.c_int(n), C.c_int(p["mode"]), C.c_int(p["luma"]))
    return {"pred": pred[:n, :n].copy()}


def k_fill_reference_samples(lib, prefix, p, rng):
    n = p["n"]
    dec = pix_plane(rng)
    adi = aligned((4 * 64 + 1 + 15,), np.int16)
    adi[...] = 0x1234
    fn(lib, prefix, "fill_reference_samples")(ptr(dec, 15 * PLANE_W + 15), C.c_int(PLANE_W), C.c_int(n), C.c_int(p["left"]), C.c_int(p["top"]),
                                              C.c_int(p["bl"]), C.c_int(p["tr"]), C.c_int(p["bl_size"]), C.c_int(p["tr_size"]), ptr(adi))
    return {"adi": adi[:4 * n + 1].copy()}


def k_adi_filter(lib, prefix, p, rng):
    n = p["n"]
    adi = adi_buffer(rng, n, smooth=p.get("smooth", False))
    out = aligned((4 * 64 + 1 + 15,), np.int16)
    fn(lib, prefix, "adi_filter")(ptr(adi), ptr(out), C.c_int(4 * n + 1), C.c_int(n), C.c_int(p["strong"]))
    return {"out": out[:4 * n + 1].copy()}


def k_interpolate(lib, prefix, p, rng):
    w, h, first = p["w"], p["h"], p["first"]
    src = pix_plane(rng, 0, 256) if first else pix_plane(rng, -8192, 8129)
    dst = aligned((80, 96), np.int16)
    dst[...] = 0x1234
    name = "interpolate_luma" if p["luma"] else "interpolate_chroma"
    fn(lib, prefix, name)(ptr(src, 16 * PLANE_W + 16 + p.get("dx", 0)), C.c_int(PLANE_W), ptr(dst), C.c_int(96), C.c_int(p["frac"]),
                          C.c_int(w), C.c_int(h), C.c_int(p["vert"]), C.c_int(first), C.c_int(p["last"]))
    return {"dst": dst[:h, :w].copy()}


def k_weighted_average(lib, prefix, p, rng):
    w, h = p["w"], p["h"]
    a, b = pix_plane(rng, -8192, 8129), pix_plane(rng, -8192, 8129)
    dst = aligned((64, 64), np.int16)
    fn(lib, prefix, "weighted_average")(ptr(a, 16 * PLANE_W + 16), C.c_int(PLANE_W), ptr(b, 8 * PLANE_W + 8), C.c_int(PLANE_W), ptr(dst), C.c_int(64),
                                        C.c_int(h), C.c_int(w))
    return {"dst": dst[:h, :w].copy()}


def k_transform(lib, prefix, p, rng):
    n, amp = p["n"], p.get("amp", 255)
    blk = aligned((64, 64), np.int16)
    blk[...] = rng.integers(-amp, amp + 1, (64, 64))
    if p.get("const") is not None:
        blk[...] = p["const"]
    coeff = aligned((32 * 32,), np.int16)
    fn(lib, prefix, "transform")(ptr(blk), ptr(coeff), C.c_int(64), C.c_int(n), C.c_int(p.get("dst", 0)))
    return {"coeff": coeff[:n * n].copy()}


def k_itransform(lib, prefix, p, rng):
    n, amp = p["n"], p.get("amp", 1500)
    coeff = aligned((32 * 32,), np.int16)
    c = rng.integers(-amp, amp + 1, n * n)
    if p.get("sparse", True):
        c = np.where(rng.random(n * n) < 0.25, c, 0)
    coeff[:n * n] = c
    blk = aligned((64, 64), np.int16)
    fn(lib, prefix, "itransform")(ptr(blk), ptr(coeff), C.c_int(64), C.c_int(n), C.c_int(p.get("dst", 0)))
    return {"block": blk[:n, :n].copy()}


def quant_depth(n, comp):
    """`depth` argument such that inv_depth = 6-(depth+(comp!=0)) equals log2(n) (hmr_sse42_functions_quant.c:39)."""
    return 6 - int(np.log2(n)) - (1 if comp else 0)


def k_quant(lib, prefix, p, rng):
    n, comp = p["n"], p["comp"]
    amp = p.get("amp", 3000)
    src = aligned((32 * 32,), np.int16)
    decay = 1.0 / (1.0 + 0.35 * (np.add.outer(np.arange(n), np.arange(n))).ravel())
    src[:n * n] = (rng.integers(-amp, amp + 1, n * n) * decay).astype(np.int64)
    if p.get("extreme"):
        src[0] = -32768
        src[1] = 32767
    dst = aligned((32 * 32,), np.int16)
    du = aligned((32 * 32,), np.int16)
    ac = C.c_int(0)
    fn(lib, prefix, "quant")(ptr(src), ptr(dst), ptr(du), C.c_int(p["scan"]), C.c_int(quant_depth(n, comp)), C.c_int(comp), C.c_int(p["intra"]),
                             C.c_int(p["slice_i"]), C.c_int(p["sbh"]), C.byref(ac), C.c_int(n), C.c_int(p["per"]), C.c_int(p["rem"]))
    return {"dst": dst[:n * n].copy(), "delta_u": du[:n * n].copy(), "ac_sum": np.array([ac.value], dtype=np.int32)}


def k_inv_quant(lib, prefix, p, rng):
    n, comp = p["n"], p["comp"]
    amp = p.get("amp", 200)
    src = aligned((32 * 32,), np.int16)
    src[:n * n] = np.where(rng.random(n * n) < 0.3, rng.integers(-amp, amp + 1, n * n), 0)
    dst = aligned((32 * 32,), np.int16)
    fn(lib, prefix, "inv_quant")(ptr(src), ptr(dst), C.c_int(quant_depth(n, comp)), C.c_int(comp), C.c_int(p["intra"]), C.c_int(n), C.c_int(p["per"]),
                                 C.c_int(p["rem"]))
    return {"dst": dst[:n * n].copy()}


def k_tu_chain(lib, prefix, p, rng):
    n, comp = p["n"], p["comp"]
    orig = aligned((64, 64), np.int16)
    orig[...] = rng.integers(0, 256, (64, 64))
    pred = aligned((64, 64), np.int16)
    pred[...] = np.clip(orig + rng.integers(-p["noise"], p["noise"] + 1, (64, 64)), 0, 255)
    levels = aligned((32 * 32,), np.int16)
    recon = aligned((64, 64), np.int16)
    ac = C.c_int(0)
    r = fn(lib, prefix, "tu_chain", C.c_uint32)(ptr(orig), C.c_int(64), ptr(pred), C.c_int(64), ptr(levels), ptr(recon), C.c_int(64), C.c_int(n),
                                                C.c_int(p.get("dst", 0)), C.c_int(p["scan"]), C.c_int(comp), C.c_int(p["intra"]), C.c_int(p["slice_i"]),
                                                C.c_int(p["sbh"]), C.c_int(p["per"]), C.c_int(p["rem"]), C.byref(ac))
    return {"levels": levels[:n * n].copy(), "recon": recon[:n, :n].copy(), "ssd": np.array([r], np.uint32), "ac_sum": np.array([ac.value], np.int32)}


def mpm_list(left_mode, top_mode):
    """Most-probable-mode list from the neighbours' modes (get_intra_dir_luma_predictor, hmr_arithmetic_encoding.c:545); -1 = not intra -> DC."""
    a = left_mode if left_mode >= 0 else 1
    b = top_mode if top_mode >= 0 else 1
    if a == b:
        return [a, ((a + 29) % 32) + 2, ((a - 1) % 32) + 2] if a > 1 else [0, 1, 26]
    return [a, b, 0 if (a and b) else (26 if a + b < 2 else 1)]


def k_intra_search(lib, prefix, p, rng):
    """homer_loop1_motion_intra: the reference derives the MPM list from neighbour modes, the oracle / GPU take it as input."""
    n = p["n"]
    H = W = 160
    yy, xx = np.mgrid[0:H, 0:W]
    th = p["theta"]
    base = 128 + p["amp"] * np.sin((xx * np.cos(th) + yy * np.sin(th)) / p["period"]) + p["tilt"] * (xx - yy) / 8.0
    dec = aligned((H, W), np.int16)
    dec[...] = np.clip(base + rng.integers(-p["noise"], p["noise"] + 1, (H, W)), 0, 255)
    orig = aligned((64, 64), np.int16)
    orig[...] = np.clip(base[16:80, 16:80] + rng.integers(-p["noise"], p["noise"] + 1, (64, 64)), 0, 255)
    adi = aligned((4 * 64 + 1 + 15,), np.int16)
    adif = aligned((4 * 64 + 1 + 15,), np.int16)
    pred = aligned((64, 64), np.int16)
    pred[...] = 0x1234
    cost = C.c_double(0)
    preds = mpm_list(p["left_mode"], p["top_mode"])
    corner = 15 * W + 15
    common = [ptr(orig), C.c_int(64), ptr(dec, corner), C.c_int(W), C.c_int(n), C.c_int(p["left"]), C.c_int(p["top"]), C.c_int(p["bl"]), C.c_int(p["tr"]),
              C.c_int(p["bl_size"]), C.c_int(p["tr_size"]), C.c_int(p["strong"])]
    if prefix == "refh_":
        out = np.zeros(6, np.int32)
        fn(lib, prefix, "intra_search")(*common, C.c_int(p["left_mode"]), C.c_int(p["top_mode"]), C.c_int(p["rd_mode"]), C.c_double(p["sqrt_lambda"]),
                                        ptr(adi), ptr(adif), ptr(pred), C.c_int(64), ptr(out), C.byref(cost))
        assert out[2] == 3 and list(out[3:6]) == preds, (list(out), preds)
    else:
        out = np.zeros(2, np.int32)
        bits = {0: ([0, 0, 0], 0), 2: ([1, 1, 1], 12)}[p["rd_mode"]]
        pa, ba = np.array(preds, np.int32), np.array(bits[0], np.int32)
        fn(lib, prefix, "intra_search")(*common, ptr(pa), ptr(ba), C.c_int(bits[1]), C.c_double(p["sqrt_lambda"]), ptr(adi), ptr(adif), ptr(pred), C.c_int(64),
                                        ptr(out), C.byref(cost))
    return {"best": out[:2].copy(), "cost": np.array([cost.value], np.float64), "adi": adi[:4 * n + 1].copy(), "adi_filtered": adif[:4 * n + 1].copy(),
            "last_pred": pred[:n, :n].copy()}


def intra_is_filtered(n, mode):
    """The host-side rule of encode_intra_cu (hmr_motion_intra.c:1011-1012, intra_filter :148)."""
    thr = {4: 10, 8: 7, 16: 1, 32: 0, 64: 10}[n]
    return int(mode != 1 and min(abs(mode - 10), abs(mode - 26)) > thr)


def k_intra_tu_chain(lib, prefix, p, rng):
    """encode_intra_cu's data path; the reconstruction goes back into the plane the neighbours come from, like in the reference."""
    n, comp = p["n"], p["comp"]
    H = W = 160
    yy, xx = np.mgrid[0:H, 0:W]
    base = 128 + p["amp"] * np.sin((xx * np.cos(p["theta"]) + yy * np.sin(p["theta"])) / p["period"])
    dec = aligned((H, W), np.int16)
    dec[...] = np.clip(base + rng.integers(-p["noise"], p["noise"] + 1, (H, W)), 0, 255)
    orig = aligned((64, 64), np.int16)
    orig[...] = np.clip(base[16:80, 16:80] + rng.integers(-p["noise"], p["noise"] + 1, (64, 64)), 0, 255)
    pred = aligned((64, 64), np.int16)
    pred[...] = 0x1234
    levels = aligned((32 * 32,), np.int16)
    ac = C.c_int(0)
    luma = 1 if comp == 0 else 0
    filt = intra_is_filtered(n, p["mode"]) if luma else 0
    r = fn(lib, prefix, "intra_tu_chain", C.c_uint32)(
        ptr(orig), C.c_int(64), ptr(dec, 15 * W + 15), C.c_int(W), C.c_int(p["left"]), C.c_int(p["top"]), C.c_int(p["bl"]), C.c_int(p["tr"]), C.c_int(p["bl_size"]),
        C.c_int(p["tr_size"]), C.c_int(p["strong"]), C.c_int(filt), C.c_int(p["mode"]), C.c_int(luma), ptr(pred), C.c_int(64), ptr(levels), ptr(dec, 16 * W + 16),
        C.c_int(W), C.c_int(n), C.c_int(1 if (n == 4 and luma) else 0), C.c_int(p["scan"]), C.c_int(comp), C.c_int(p["slice_i"]), C.c_int(p["sbh"]),
        C.c_int(p["per"]), C.c_int(p["rem"]), C.byref(ac))
    return {"pred": pred[:n, :n].copy(), "levels": levels[:n * n].copy(), "plane": dec.copy(), "ssd": np.array([r], np.uint32),
            "ac_sum": np.array([ac.value], np.int32)}


def k_inter_tu_chain(lib, prefix, p, rng):
    """encode_inter_cu / _chroma per-TU sequence: the residual of a motion-compensated CU, keep-or-drop decision included."""
    n, comp = p["n"], p["comp"]
    res = aligned((64, 64), np.int16)
    yy, xx = np.mgrid[0:64, 0:64]
    res[...] = (p["amp"] * np.sin((xx * np.cos(p["theta"]) + yy * np.sin(p["theta"])) / p["period"]) + rng.integers(-p["noise"], p["noise"] + 1, (64, 64))).astype(np.int64)
    pred = pix_plane(rng)
    levels = aligned((32 * 32,), np.int16)
    levels[...] = 0x1234
    recon = aligned((64, 64), np.int16)
    ac = C.c_int(0)
    r = fn(lib, prefix, "inter_tu_chain", C.c_uint32)(ptr(res), C.c_int(64), ptr(pred, 16 * PLANE_W + 16), C.c_int(PLANE_W), ptr(levels), ptr(recon), C.c_int(64),
                                                     C.c_int(n), C.c_int(p["scan"]), C.c_int(comp), C.c_int(p["slice_i"]), C.c_int(p["sbh"]), C.c_int(p["per"]),
                                                     C.c_int(p["rem"]), C.c_double(p["weight"] if comp else 1.0), C.c_double(p["thr"]), C.byref(ac))
    return {"levels": levels[:n * n].copy(), "recon": recon[:n, :n].copy(), "ssd": np.array([r], np.uint32), "ac_sum": np.array([ac.value], np.int32)}


def cu_tree_neighbours(n, flags, pict_w, pict_h):
    """5 x {left, top, bottom_left, top_right, bl_size, tr_size} for a CU of size n at the origin and its four quadrants: the flags are the partition
    nodes' neighbour fields, the run lengths follow from the picture extent (hmr_motion_intra.c:289,335)."""
    h = n // 2
    geo = [(0, 0, n)] + [((k & 1) * h, (k >> 1) * h, h) for k in range(4)]
    nb = []
    for (x, y, sz), f in zip(geo, flags):
        nb += list(f) + [min(sz, pict_h - (y + sz)), min(sz, pict_w - (x + sz))]
    return np.array(nb, np.int32)


def k_intra_luma_cu(lib, prefix, p, rng):
    """encode_intra_luma for one 2Nx2N CU: mode search + one-level transform tree + consolidation.  The reference runs its real function on its own thread
    context (CU at the origin of CTU 0, no neighbouring CTUs: MPM list planar / DC / vertical); oracle and GPU take the flat form."""
    n = p["n"]
    W = 192
    yy, xx = np.mgrid[0:W, 0:W]
    th = p["theta"]
    base = 128 + p["amp"] * np.sin((xx * np.cos(th) + yy * np.sin(th)) / p["period"]) + p["tilt"] * (xx - yy) / 8.0
    img = np.clip(base + rng.integers(-p["noise"], p["noise"] + 1, (W, W)), 0, 255).astype(np.int16)
    rec = np.clip(img + rng.integers(-3, 4, (W, W)), 0, 255).astype(np.int16)     # "already coded" neighbourhood
    orig = aligned((n, n), np.int16)
    orig[...] = img[16:16 + n, 16:16 + n]
    top = np.ascontiguousarray(rec[15, 15:16 + 2 * n])
    left = np.ascontiguousarray(rec[16:16 + 2 * n, 15])
    flags = p["flags"]
    nb = cu_tree_neighbours(n, flags, p["pict_w"], p["pict_h"])
    out = np.zeros(24, np.int32)
    dec_par, dec_chl = aligned((n, n), np.int16), aligned((n, n), np.int16)
    lev_par, lev_chl = aligned((n * n,), np.int16), aligned((n * n,), np.int16)
    if prefix == "refh_":
        fl = np.array([f for node in flags for f in node], np.int32)
        rc = fn(lib, prefix, "intra_luma_cu", C.c_int)(ptr(orig), ptr(top), ptr(left), ptr(fl), C.c_int(p["pict_w"]), C.c_int(p["pict_h"]), C.c_int(n), C.c_int(p["qp"]),
                                                       C.c_double(p["sqrt_lambda"]), C.c_int(p["rd_mode"]), C.c_int(p["slice_i"]), C.c_int(p["sbh"]),
                                                       C.c_int(p["strong"]), ptr(out), ptr(dec_par), ptr(dec_chl), ptr(lev_par), ptr(lev_chl))
        assert rc == 0
        res = {}
    else:
        S = 2 * n + 16
        planes = []
        for _ in range(2):
            pl = aligned((S, S), np.int16)
            pl[...] = 0x0101
            pl[7, 7:8 + 2 * n] = top
            pl[8:8 + 2 * n, 7] = left
            planes.append(pl)
        pred = aligned((n, n), np.int16)
        adi, adif = aligned((4 * 64 + 16,), np.int16), aligned((4 * 64 + 16,), np.int16)
        bits = {0: ([0, 0, 0], 0), 2: ([1, 1, 1], 12)}[p["rd_mode"]]
        pa, ba = np.array([0, 1, 26], np.int32), np.array(bits[0], np.int32)
        cost = C.c_double(0)
        fn(lib, prefix, "intra_luma_cu")(ptr(orig), C.c_int(n), ptr(planes[0], 8 * S + 8), C.c_int(S), ptr(planes[1], 8 * S + 8), C.c_int(S), ptr(nb), C.c_int(p["strong"]),
                                         ptr(pa), ptr(ba), C.c_int(bits[1]), C.c_double(p["sqrt_lambda"]), ptr(adi), ptr(adif), ptr(pred), C.c_int(n), ptr(lev_par),
                                         ptr(lev_chl), C.c_int(n), C.c_int(p["slice_i"]), C.c_int(p["sbh"]), C.c_int(p["qp"] // 6), C.c_int(p["qp"] % 6),
                                         C.c_int(1 if p["rd_mode"] == 2 else 0), ptr(out), C.byref(cost))
        dec_par[...] = planes[0][8:8 + n, 8:8 + n]
        dec_chl[...] = planes[1][8:8 + n, 8:8 + n]
        # nothing outside the CU may change in either plane
        for pl in planes:
            chk = pl.copy()
            chk[8:8 + n, 8:8 + n] = 0x0101
            chk[7, 7:8 + 2 * n] = 0x0101
            chk[8:8 + 2 * n, 7] = 0x0101
            assert (chk == 0x0101).all(), "write outside the CU"
        res = {"pred": pred.copy(), "bits_cost": np.array([out[20], cost.value], np.float64)}
    info = out[1:20].copy()
    if n == 64:
        info[8] = info[13] = 0            # no parent TU: its ssd / sum slots are not defined
    res.update({"info": info, "dec_par": dec_par.copy(), "dec_chl": dec_chl.copy(), "lev_par": lev_par.copy(), "lev_chl": lev_chl.copy()})
    return res


# chroma QP mapping of the standard (chroma_scale_conversion_table, hmr_tables.c) - checked against the compiled reference in test_oracle_vs_ref.py
CHROMA_QP = list(range(30)) + [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37] + [q - 6 for q in range(44, 58)]


def k_intra_chroma_cu(lib, prefix, p, rng):
    """encode_intra_chroma for one 2Nx2N CU (n = chroma size): five-candidate search on U and V, then the TUs of the winner along the luma tree.  The reference
    runs its real function on its own thread context (chroma_qp_offset 2 as configured in refh_open); oracle and GPU take the flat form."""
    n = p["n"]
    W = 96
    yy, xx = np.mgrid[0:W, 0:W]
    orig, top, left, planes = [], [], [], []
    for c in range(2):
        th = p["theta"] + 0.4 * c
        base = 128 + p["amp"] * np.sin((xx * np.cos(th) + yy * np.sin(th)) / p["period"]) + p["tilt"] * (xx - yy) / 8.0
        img = np.clip(base + rng.integers(-p["noise"], p["noise"] + 1, (W, W)), 0, 255).astype(np.int16)
        rec = np.clip(img + rng.integers(-3, 4, (W, W)), 0, 255).astype(np.int16)
        o = aligned((n, n), np.int16)
        o[...] = img[16:16 + n, 16:16 + n]
        orig.append(o)
        top.append(np.ascontiguousarray(rec[15, 15:16 + 2 * n])); left.append(np.ascontiguousarray(rec[16:16 + 2 * n, 15]))
    flags = p["flags"]
    nb = cu_tree_neighbours(n, flags, p["pict_w"], p["pict_h"])
    out = np.zeros(16, np.int32)
    dec = [aligned((n, n), np.int16), aligned((n, n), np.int16)]
    lev = [aligned((n * n,), np.int16), aligned((n * n,), np.int16)]
    qpc = CHROMA_QP[min(max(p["qp"] + 2, 0), 57)]
    weight = 2.0 ** ((p["slice_qp"] - CHROMA_QP[min(max(p["slice_qp"] + 2, 0), 57)]) / 3.0)
    if prefix == "refh_":
        fl = np.array([f for node in flags for f in node], np.int32)
        rc = fn(lib, prefix, "intra_chroma_cu", C.c_int)(ptr(orig[0]), ptr(orig[1]), ptr(top[0]), ptr(left[0]), ptr(top[1]), ptr(left[1]), ptr(fl), C.c_int(p["pict_w"]),
                                                         C.c_int(p["pict_h"]), C.c_int(n), C.c_int(p["luma_mode"]), C.c_int(p["split"]), C.c_int(p["qp"]),
                                                         C.c_int(p["slice_qp"]), C.c_double(p["sqrt_lambda"]), C.c_int(p["rd_mode"]), C.c_int(p["slice_i"]),
                                                         C.c_int(p["sbh"]), ptr(out), ptr(dec[0]), ptr(dec[1]), ptr(lev[0]), ptr(lev[1]))
        assert rc == 0, "work window and consolidated window differ"
        res = {}
    else:
        S = 2 * n + 16
        for c in range(2):
            pl = aligned((S, S), np.int16)
            pl[...] = 0x0101
            pl[7, 7:8 + 2 * n] = top[c]
            pl[8:8 + 2 * n, 7] = left[c]
            planes.append(pl)
        pred = [aligned((n, n), np.int16), aligned((n, n), np.int16)]
        fn(lib, prefix, "intra_chroma_cu")(ptr(orig[0]), ptr(orig[1]), C.c_int(n), ptr(planes[0], 8 * S + 8), ptr(planes[1], 8 * S + 8), C.c_int(S), ptr(nb),
                                           C.c_int(p["luma_mode"]), C.c_int(p["split"]), C.c_double(p["sqrt_lambda"]), C.c_double(weight), ptr(pred[0]), ptr(pred[1]),
                                           C.c_int(n), ptr(lev[0]), ptr(lev[1]), C.c_int(n), C.c_int(p["slice_i"]), C.c_int(p["sbh"]), C.c_int(qpc // 6),
                                           C.c_int(qpc % 6), ptr(out))
        for c in range(2):
            dec[c][...] = planes[c][8:8 + n, 8:8 + n]
            chk = planes[c].copy()
            chk[8:8 + n, 8:8 + n] = 0x0101
            chk[7, 7:8 + 2 * n] = 0x0101
            chk[8:8 + 2 * n, 7] = 0x0101
            assert (chk == 0x0101).all(), "write outside the CU"
        res = {"pred_u": pred[0].copy(), "pred_v": pred[1].copy(), "search": out[[1, 3]].copy(), "ac": out[6:14].copy()}
    res.update({"info": out[[0, 2, 4, 5]].copy(), "dec_u": dec[0].copy(), "dec_v": dec[1].copy(), "lev_u": lev[0].copy(), "lev_v": lev[1].copy()})
    return res


def k_sao_offsets(lib, prefix, p, rng):
    """sao_derive_offsets + sao_invert_quant_offsets + sao_get_distortion for the 15 (component, type) pairs of a CTU."""
    stats = np.zeros((3, 5, 2, 32), np.int32)
    for comp in range(3):
        area = 4096 if comp == 0 else 1024
        for t in range(5):
            ncls = 32 if t == 4 else 5
            cnt = rng.multinomial(int(area * rng.uniform(0.2, 1.0)), rng.dirichlet(np.full(ncls, p["conc"])))
            cnt[rng.random(ncls) < p["empty"]] = 0
            sign = rng.choice([-1, 1], ncls) if t == 4 else np.array([1, 1, 0, -1, -1]) * rng.choice([1, 1, 1, -1], ncls)   # valleys are mostly lifted, peaks lowered
            stats[comp, t, 1, :ncls] = cnt
            stats[comp, t, 0, :ncls] = np.round(cnt * sign * rng.gamma(1.0, p["bias"], ncls)).astype(np.int64)
    lambdas = np.array([p["lambda"], p["lambda"] * p["chroma_ratio"], p["lambda"] * p["chroma_ratio"]], np.float64)
    offsets, aux, dist = np.zeros((3, 5, 32), np.int32), np.zeros((3, 5), np.int32), np.zeros((3, 5), np.int64)
    fn(lib, prefix, "sao_offsets_ctu")(ptr(stats), ptr(lambdas), ptr(offsets), ptr(aux), ptr(dist))
    return {"offsets": offsets, "aux": aux, "dist": dist}


KERNELS = {
    "sad": k_sad, "ssd16b": k_ssd16b, "predict": k_predict, "reconst": k_reconst, "modified_variance": k_modified_variance,
    "copy": k_copy, "intra_planar": k_intra_planar, "intra_angular": k_intra_angular,
    "fill_reference_samples": k_fill_reference_samples, "adi_filter": k_adi_filter, "interpolate": k_interpolate,
    "weighted_average": k_weighted_average, "transform": k_transform, "itransform": k_itransform, "quant": k_quant,
    "inv_quant": k_inv_quant, "tu_chain": k_tu_chain, "intra_search": k_intra_search, "intra_tu_chain": k_intra_tu_chain, "inter_tu_chain": k_inter_tu_chain,
    "intra_luma_cu": k_intra_luma_cu, "intra_chroma_cu": k_intra_chroma_cu, "sao_offsets": k_sao_offsets,
}


def run(lib, prefix, case):
    kernel, params, seed = case
    return KERNELS[kernel](lib, prefix, params, np.random.default_rng(seed))


# ------------------------------------------------------------------ case lists

def all_cases(level="full"):
    """level 'golden' = the committed fixture set (small), 'full' = the sweep used for pinning/parity."""
    full = level == "full"
    cases = []
    seed = [1000]

    def add(kernel, **p):
        seed[0] += 1
        cases.append((kernel, p, seed[0]))

    sizes = [4, 8, 16, 32, 64]
    for n in sizes:
        for dx, dy in ([(0, 0), (1, 0), (3, -1), (-1, 1), (7, 5)] if full else [(0, 0), (3, -1)]):
            add("sad", n=n, dx=dx, dy=dy)
            add("ssd16b", n=n, dx=dx, dy=dy)
        add("ssd16b", n=n, dx=0, dy=0, pred_stride0=1, src_range=(-255, 256), pred_range=(0, 1))
        add("ssd16b", n=n, dx=2, dy=0, src_range=(-255, 256), pred_range=(-255, 256))
        add("predict", n=n)
        add("reconst", n=n)
        add("reconst", n=n, res_stride0=1)
        add("reconst", n=n, extreme=1)
    for n in [2, 4, 8, 16, 32, 64]:
        for modif in (1, 2):
            add("modified_variance", n=n, modif=modif)
    for kind in ("16_16", "8_16", "16_8"):
        for h, w in [(4, 4), (8, 8), (16, 16), (7, 32), (33, 48), (64, 64)]:
            add("copy", kind=kind, h=h, w=w)
    for n in sizes:
        for _ in range(3 if full else 1):
            add("intra_planar", n=n)
        for mode in range(1, 35):
            for luma in (1, 0):
                if full or mode in (1, 2, 6, 10, 14, 18, 22, 26, 30, 34) or n == 8:
                    add("intra_angular", n=n, mode=mode, luma=luma)
        add("intra_angular", n=n, mode=1, luma=1, flat=255)
        add("intra_angular", n=n, mode=10, luma=1, flat=0)
        add("intra_angular", n=n, mode=26, luma=1, flat=255)
        for left, top, bl, tr in [(0, 0, 0, 0), (1, 0, 0, 0), (0, 1, 0, 0), (1, 1, 0, 0), (1, 1, 1, 0), (1, 1, 0, 1), (1, 1, 1, 1), (1, 0, 1, 0), (0, 1, 0, 1)]:
            add("fill_reference_samples", n=n, left=left, top=top, bl=bl, tr=tr, bl_size=n if bl else 0, tr_size=n if tr else 0)
        if n >= 8:
            add("fill_reference_samples", n=n, left=1, top=1, bl=1, tr=1, bl_size=n // 2, tr_size=n // 4)
            add("fill_reference_samples", n=n, left=1, top=1, bl=1, tr=1, bl_size=n, tr_size=n // 2)
        for strong in (0, 1):
            add("adi_filter", n=n, strong=strong)
            add("adi_filter", n=n, strong=strong, smooth=True)
    stage_modes = [(1, 0), (0, 1), (1, 1), (0, 0)]
    for frac in range(4):
        for first, last in stage_modes:
            for vert in (0, 1):
                for w, h in ([(8, 8), (16, 16), (9, 16), (17, 24), (32, 40), (64, 64), (65, 72)] if full else [(8, 8), (17, 24), (64, 64)]):
                    add("interpolate", luma=1, frac=frac, first=first, last=last, vert=vert, w=w, h=h, dx=(frac + w) % 3)
    for frac in range(8):
        for first, last in stage_modes:
            for vert in (0, 1):
                for w, h in ([(4, 4), (4, 9), (8, 8), (8, 13), (16, 16), (32, 32), (32, 37)] if full else [(4, 4), (8, 13), (32, 32)]):
                    add("interpolate", luma=0, frac=frac, first=first, last=last, vert=vert, w=w, h=h, dx=frac % 2)
    add("interpolate", luma=0, frac=0, first=1, last=1, vert=0, w=2, h=2)
    for w, h in [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4)]:
        add("weighted_average", w=w, h=h)
    for n in (4, 8, 16, 32):
        for dst in ((0, 1) if n == 4 else (0,)):
            for _ in range(4 if full else 2):
                add("transform", n=n, dst=dst)
                add("itransform", n=n, dst=dst)
            add("transform", n=n, dst=dst, const=255)
            add("transform", n=n, dst=dst, const=-255)
            add("transform", n=n, dst=dst, amp=1)
            add("itransform", n=n, dst=dst, amp=20000, sparse=False)
            add("itransform", n=n, dst=dst, amp=32767, sparse=False)
            add("itransform", n=n, dst=dst, amp=60)
    for n in (4, 8, 16, 32):
        for comp in ((0, 1, 2) if n < 32 else (0,)):
            for intra in (1, 0):
                for slice_i in ((1,) if intra and not full else (1, 0)):
                    for sbh in (1, 0):
                        for per, rem in ([(5, 2), (3, 0), (0, 5), (8, 3), (4, 4), (6, 1)] if full else [(5, 2), (3, 0)]):
                            for scan in ((1, 2, 3) if (full and intra) else (3,)):
                                add("quant", n=n, comp=comp, intra=intra, slice_i=slice_i, sbh=sbh, per=per, rem=rem, scan=scan)
            add("quant", n=n, comp=comp, intra=1, slice_i=1, sbh=1, per=5, rem=2, scan=3, extreme=1)
            add("quant", n=n, comp=comp, intra=0, slice_i=0, sbh=1, per=4, rem=1, scan=3, amp=40)
            for intra in (1, 0):
                for per, rem in [(5, 2), (3, 0), (0, 5), (8, 3), (1, 1)]:
                    add("inv_quant", n=n, comp=comp, intra=intra, per=per, rem=rem)
            add("inv_quant", n=n, comp=comp, intra=1, per=8, rem=5, amp=32000)
    for n in (4, 8, 16, 32):
        for comp in ((0, 1, 2) if n < 32 else (0,)):
            for intra in (1, 0):
                for noise in ((1, 6, 40, 255) if full else (2, 40)):
                    for per, rem in ([(5, 2), (3, 4)] if full else [(5, 2)]):
                        add("tu_chain", n=n, comp=comp, intra=intra, slice_i=intra, sbh=1, per=per, rem=rem, scan=3, noise=noise,
                            dst=1 if (n == 4 and intra and comp == 0) else 0)
    r = np.random.default_rng(4242)
    for n in sizes:
        for i in range(40 if full else 8):
            left, top = (int(r.integers(0, 2)), int(r.integers(0, 2))) if i % 5 == 0 else (1, 1)
            bl, tr = int(r.integers(0, 2)) & left, int(r.integers(0, 2)) & top
            add("intra_search", n=n, left=left, top=top, bl=bl, tr=tr, bl_size=(n if r.random() < 0.7 else max(n // 2, 4)) if bl else 0,
                tr_size=(n if r.random() < 0.7 else max(n // 2, 4)) if tr else 0, strong=int(r.integers(0, 2)),
                left_mode=int(r.integers(-1, 35)), top_mode=int(r.integers(-1, 35)), rd_mode=int(r.choice([2, 2, 0])),
                sqrt_lambda=float(r.uniform(2.0, 60.0)), theta=float(r.uniform(0, np.pi)), period=float(r.uniform(3.0, 25.0)),
                amp=float(r.uniform(5, 90)), tilt=float(r.uniform(-6, 6)), noise=int(r.integers(0, 9)))
    r = np.random.default_rng(777)
    for n in (4, 8, 16, 32):
        for i in range(24 if full else 6):
            left, top = (int(r.integers(0, 2)), int(r.integers(0, 2))) if i % 6 == 0 else (1, 1)
            bl, tr = int(r.integers(0, 2)) & left, int(r.integers(0, 2)) & top
            add("intra_tu_chain", n=n, comp=int(r.choice([0, 0, 1, 2])) if n < 32 else 0, left=left, top=top, bl=bl, tr=tr,
                bl_size=(n if r.random() < 0.7 else max(n // 2, 4)) if bl else 0, tr_size=(n if r.random() < 0.7 else max(n // 2, 4)) if tr else 0,
                strong=int(r.integers(0, 2)), mode=int(r.integers(0, 35)), scan=int(r.integers(1, 4)), slice_i=int(r.integers(0, 2)), sbh=int(r.integers(0, 2)),
                per=int(r.integers(2, 7)), rem=int(r.integers(0, 6)), theta=float(r.uniform(0, np.pi)), period=float(r.uniform(3.0, 25.0)),
                amp=float(r.uniform(5, 90)), noise=int(r.choice([0, 2, 8, 30])))
    r = np.random.default_rng(991)
    for n in (4, 8, 16, 32):
        for i in range(30 if full else 8):
            comp = int(r.choice([0, 0, 1, 2])) if n < 32 else 0
            add("inter_tu_chain", n=n, comp=comp, scan=3, slice_i=0, sbh=int(r.integers(0, 2)), per=int(r.integers(2, 7)), rem=int(r.integers(0, 6)),
                weight=float(2.0 ** (r.integers(-2, 5) / 3.0)), thr=float(np.clip(r.uniform(0, 3000) / 2.5 - 5.0, 1.0, 20000.0)),
                theta=float(r.uniform(0, np.pi)), period=float(r.uniform(2.0, 20.0)), amp=float(r.choice([0, 2, 6, 20, 60])), noise=int(r.choice([0, 1, 3, 10])))
    r = np.random.default_rng(1213)
    for n in (8, 16, 32, 64):
        for i in range(30 if full else 6):
            left, top = (int(r.integers(0, 2)), int(r.integers(0, 2))) if i % 5 == 0 else (1, 1)
            bl, tr = int(r.integers(0, 2)) & left, int(r.integers(0, 2)) & top
            # the quadrants' flags as cu_partition_get_neighbours derives them, with an occasional arbitrary combination
            flags = [(left, top, bl, tr), (left, top, left, top), (1, top, 0, tr), (left, 1, bl, 1), (1, 1, 0, 0)]
            if i % 7 == 3:        # any combination a frame can produce: below-left needs left, above-right needs above, quadrant 1 has no below-left, the last quadrant neither
                flags = []
                for node in range(5):
                    fl, ft = int(r.integers(0, 2)), int(r.integers(0, 2))
                    flags.append((fl, ft, int(r.integers(0, 2)) & fl & (node not in (2, 4)), int(r.integers(0, 2)) & ft & (node != 4)))
            add("intra_luma_cu", n=n, flags=flags, pict_w=int(r.choice([n, n + n // 2, 2 * n, 4 * n])), pict_h=int(r.choice([n, n + n // 2, 2 * n, 4 * n])),
                qp=int(r.integers(18, 45)), sqrt_lambda=float(r.uniform(2.0, 60.0)), rd_mode=int(r.choice([2, 2, 0])), slice_i=int(r.integers(0, 2)),
                sbh=int(r.integers(0, 2)), strong=int(r.integers(0, 2)), theta=float(r.uniform(0, np.pi)), period=float(r.uniform(3.0, 25.0)),
                amp=float(r.uniform(5, 90)), tilt=float(r.uniform(-6, 6)), noise=int(r.choice([0, 1, 3, 8, 20])))
    r = np.random.default_rng(1415)
    for n in (4, 8, 16, 32):        # chroma sizes: CUs of 8 ... 64 luma samples
        for i in range(30 if full else 6):
            left, top = (int(r.integers(0, 2)), int(r.integers(0, 2))) if i % 5 == 0 else (1, 1)
            bl, tr = int(r.integers(0, 2)) & left, int(r.integers(0, 2)) & top
            flags = [(left, top, bl, tr), (left, top, left, top), (1, top, 0, tr), (left, 1, bl, 1), (1, 1, 0, 0)]
            add("intra_chroma_cu", n=n, flags=flags, pict_w=int(r.choice([n, n + n // 2, 2 * n, 4 * n])), pict_h=int(r.choice([n, n + n // 2, 2 * n, 4 * n])),
                luma_mode=int(r.choice([0, 1, 10, 26, int(r.integers(2, 35))])), split=int(r.integers(0, 2)) if n < 32 else 1, qp=int(r.integers(18, 45)),
                slice_qp=int(r.integers(22, 40)), sqrt_lambda=float(r.uniform(2.0, 60.0)), rd_mode=int(r.choice([2, 2, 0])), slice_i=int(r.integers(0, 2)),
                sbh=int(r.integers(0, 2)), theta=float(r.uniform(0, np.pi)), period=float(r.uniform(3.0, 25.0)), amp=float(r.uniform(5, 90)),
                tilt=float(r.uniform(-6, 6)), noise=int(r.choice([0, 1, 3, 8, 20])))
    r = np.random.default_rng(1617)
    for i in range(120 if full else 16):
        add("sao_offsets", conc=float(r.choice([0.2, 1.0, 5.0])), empty=float(r.choice([0.0, 0.2, 0.6])), bias=float(r.choice([0.1, 0.5, 1.5, 4.0, 9.0])),
            chroma_ratio=float(r.uniform(0.5, 1.2)), **{"lambda": float(r.choice([0.5, 4.0, 20.0, 56.0, 150.0, 600.0]) * r.uniform(0.7, 1.4))})
    return cases
