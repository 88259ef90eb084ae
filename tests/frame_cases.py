"""Frame-level in-loop filter cases (deblock K17, SAO stats K16, SAO offset K18, padding K20).

Side-info comes from REAL encodes with the reference encoder (build container, tests/golden/make_golden_frames.py)
and is committed as a fixture; sample planes are seeded synthetic pictures with block-edge steps so that the
strong / weak / chroma filter branches and every SAO class are exercised.
"""
import ctypes as C

import numpy as np

from kernel_cases import ptr

UNIT_INTRA, UNIT_CBF_Y, UNIT_EDGE_VER, UNIT_EDGE_HOR = 1, 2, 4, 8
PAD_X, PAD_Y = 80, 80      # reference frames: wnd_realloc(..., ctu+16, ctu+16) hmr_encoder_lib.c:1514


def blocky_planes(orig_y, orig_u, orig_v, seed):
    """Pre-deblock stand-in: the source picture averaged over 8x8 luma / 4x4 chroma blocks (flat blocks, so the
    d < beta decision fires), plus a random DC step per block (mix of strong / weak / no filtering) and +-1 noise."""
    rng = np.random.default_rng(seed)
    out = []
    for pl, bs in ((orig_y, 8), (orig_u, 4), (orig_v, 4)):
        h, w = pl.shape
        hb, wb = (h + bs - 1) // bs, (w + bs - 1) // bs
        padded = np.zeros((hb * bs, wb * bs), np.int64)
        padded[:h, :w] = pl
        padded[h:, :] = padded[h - 1:h, :]
        padded[:, w:] = padded[:, w - 1:w]
        mean = padded.reshape(hb, bs, wb, bs).mean(axis=(1, 3)).astype(np.int64)
        step = rng.integers(-6, 7, (hb, wb)) * (rng.random((hb, wb)) < 0.7)
        flat = np.kron(mean + step, np.ones((bs, bs), dtype=np.int64))[:h, :w]
        noise = rng.integers(-1, 2, (h, w)) * (rng.random((h, w)) < 0.3)
        out.append(np.clip(flat + noise, 0, 255).astype(np.int16))
    return out


def flags_from_info(info, width, height, lib, prefix):
    """flags = intra | cbf | edge bits; edge bits derived by the library under test from (pred_depth, tr_idx)."""
    flags = ((info["pred_mode"] != 0) * UNIT_INTRA + (info["cbf_y"] != 0) * UNIT_CBF_Y).astype(np.uint8)
    flags = np.ascontiguousarray(flags)
    f = getattr(lib, prefix + "make_edge_flags")
    f.restype = None
    f(ptr(np.ascontiguousarray(info["pred_depth"])), ptr(np.ascontiguousarray(info["tr_idx"])), C.c_int(width), C.c_int(height),
      C.c_int(flags.shape[1]), ptr(flags))
    return flags


def random_sao_params(n_ctus, seed):
    """Valid-looking SAO parameters: every type, offsets within the 8-bit range of +-7."""
    rng = np.random.default_rng(seed)
    p = np.zeros((n_ctus, 3, 34), np.int32)
    for c in range(n_ctus):
        for comp in range(3):
            mode = int(rng.integers(0, 3) > 0)
            typ = int(rng.integers(0, 5))
            p[c, comp, 0], p[c, comp, 1] = mode, typ
            if typ == 4:
                band = int(rng.integers(0, 29))
                p[c, comp, 2 + band:2 + band + 4] = rng.integers(-7, 8, 4)
            else:
                p[c, comp, 2:7] = [rng.integers(0, 8), rng.integers(0, 8), 0, -rng.integers(0, 8), -rng.integers(0, 8)]
    return p


# ------------------------------------------------------------------ runners

INFO_NAMES = ["mvx", "mvy", "ref_idx", "qp", "pred_mode", "cbf_y", "pred_depth", "tr_idx"]


class Frame(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("stride_y", C.c_int), ("stride_c", C.c_int)]


class Units(C.Structure):
    _fields_ = [("units_stride", C.c_int), ("mvx", C.c_void_p), ("mvy", C.c_void_p), ("ref_idx", C.c_void_p), ("qp", C.c_void_p), ("flags", C.c_void_p)]


def run_oracle(ora, case):
    """case: dict(width, height, info arrays, planes pre-deblock (int16), orig planes, sao params, deblock params)."""
    W, H = case["width"], case["height"]
    info = case["info"]
    W4 = info["qp"].shape[1]
    flags = flags_from_info(info, W, H, ora, "ora_")
    y, u, v = [np.ascontiguousarray(p.copy()) for p in case["pre"]]
    bsv = np.zeros(info["qp"].shape, np.uint8)
    bsh = bsv.copy()
    ora.ora_deblock_frame(ptr(y), C.c_int(W), ptr(u), ptr(v), C.c_int(W // 2), C.c_int(W), C.c_int(H), C.c_int(W4), ptr(info["mvx"]), ptr(info["mvy"]),
                          ptr(info["ref_idx"]), ptr(info["qp"]), ptr(flags), *[C.c_int(int(x)) for x in case["dbk"]], ptr(bsv), ptr(bsh))
    n_ctu = ((W + 63) // 64) * ((H + 63) // 64)
    stats = np.zeros((n_ctu, 3, 5, 2, 32), np.int32)
    oy, ou, ov = [np.ascontiguousarray(p) for p in case["orig"]]
    ora.ora_sao_stats_frame(ptr(oy), ptr(ou), ptr(ov), C.c_int(W), C.c_int(W // 2), ptr(y), ptr(u), ptr(v), C.c_int(W), C.c_int(W // 2), C.c_int(W), C.c_int(H),
                            ptr(stats))
    ay, au, av = y.copy(), u.copy(), v.copy()
    params = np.ascontiguousarray(case["sao_params"])
    ora.ora_sao_apply_frame(ptr(y), ptr(u), ptr(v), ptr(ay), ptr(au), ptr(av), C.c_int(W), C.c_int(W // 2), C.c_int(W), C.c_int(H), ptr(params))
    padded = []
    for pl, pad in ((ay, PAD_X), (au, PAD_X // 2), (av, PAD_X // 2)):
        h, w = pl.shape
        st = w + 2 * pad
        P = np.full((h + 2 * pad, st), 0x1234, np.int16)
        P[pad:pad + h, pad:pad + w] = pl
        ora.ora_pad_plane(ptr(P, pad * st + pad), C.c_int(st), C.c_int(w), C.c_int(h), C.c_int(pad), C.c_int(pad))
        padded.append(P)
    return {"flags": flags, "bs_ver": bsv, "bs_hor": bsh, "deblocked": [y, u, v], "stats": stats, "sao": [ay, au, av], "padded": padded}


class GpuSession:
    """Device buffers through the C ABI (hmr_gpu_malloc / upload / download); no torch needed."""

    def __init__(self, lib):
        self.lib = lib
        self.ctx = C.c_void_p()
        rc = lib.hmr_gpu_create(C.byref(self.ctx), 0, None)
        if rc != 0:
            lib.hmr_gpu_last_error.restype = C.c_char_p
            raise RuntimeError(lib.hmr_gpu_last_error().decode())
        self.bufs = []

    def up(self, arr):
        arr = np.ascontiguousarray(arr)
        d = C.c_void_p()
        assert self.lib.hmr_gpu_malloc(self.ctx, C.byref(d), C.c_size_t(max(arr.nbytes, 16))) == 0
        assert self.lib.hmr_gpu_upload(self.ctx, d, ptr(arr), C.c_size_t(arr.nbytes)) == 0
        self.bufs.append(d)
        return d

    def down(self, d, like, offset_bytes=0):
        out = np.empty_like(like)
        assert self.lib.hmr_gpu_download(self.ctx, ptr(out), C.c_void_p(d.value + offset_bytes), C.c_size_t(out.nbytes)) == 0
        return out

    def close(self):
        for d in self.bufs:
            self.lib.hmr_gpu_free(self.ctx, d)
        self.lib.hmr_gpu_destroy(self.ctx)


def run_gpu(lib, case):
    """Same pipeline on the GPU, planes resident in padded device windows (stride = width + 2*pad)."""
    W, H = case["width"], case["height"]
    info = case["info"]
    W4 = info["qp"].shape[1]
    s = GpuSession(lib)
    try:
        pads = (PAD_X, PAD_X // 2, PAD_X // 2)

        def padded_host(pl, pad):
            h, w = pl.shape
            P = np.full((h + 2 * pad, w + 2 * pad), 0x1234, np.int16)
            P[pad:pad + h, pad:pad + w] = pl
            return P

        def frame_of(dev, shapes):
            f = Frame()
            f.width, f.height = W, H
            ptrs = []
            for d, (h, w), pad in zip(dev, shapes, pads):
                ptrs.append(d.value + 2 * (pad * (w + 2 * pad) + pad))
            f.y, f.u, f.v = ptrs
            f.stride_y, f.stride_c = W + 2 * PAD_X, W // 2 + PAD_X
            return f

        shapes = [p.shape for p in case["pre"]]
        host_pre = [padded_host(p, pad) for p, pad in zip(case["pre"], pads)]
        d_rec = [s.up(p) for p in host_pre]
        d_org = [s.up(padded_host(p, pad)) for p, pad in zip(case["orig"], pads)]
        f_rec, f_org = frame_of(d_rec, shapes), frame_of(d_org, shapes)
        flags0 = ((info["pred_mode"] != 0) * UNIT_INTRA + (info["cbf_y"] != 0) * UNIT_CBF_Y).astype(np.uint8)
        d_flags = s.up(flags0)
        assert lib.hmr_gpu_edge_flags_frame(s.ctx, s.up(info["pred_depth"]), s.up(info["tr_idx"]), C.c_int(W), C.c_int(H), C.c_int(W4), d_flags) == 0
        units = Units(W4, s.up(info["mvx"]), s.up(info["mvy"]), s.up(info["ref_idx"]), s.up(info["qp"]), d_flags)
        d_bsv, d_bsh = s.up(np.zeros(info["qp"].shape, np.uint8)), s.up(np.zeros(info["qp"].shape, np.uint8))
        rc = lib.hmr_gpu_deblock_frame(s.ctx, C.byref(f_rec), C.byref(units), *[C.c_int(int(x)) for x in case["dbk"]], d_bsv, d_bsh)
        assert rc == 0, rc
        out = {"flags": s.down(d_flags, flags0), "bs_ver": s.down(d_bsv, flags0), "bs_hor": s.down(d_bsh, flags0)}

        def crop(dev):
            res = []
            for d, hp, (h, w), pad in zip(dev, host_pre, shapes, pads):
                res.append(np.ascontiguousarray(s.down(d, hp)[pad:pad + h, pad:pad + w]))
            return res

        out["deblocked"] = crop(d_rec)
        n_ctu = ((W + 63) // 64) * ((H + 63) // 64)
        stats0 = np.zeros((n_ctu, 3, 5, 2, 32), np.int32)
        d_stats = s.up(stats0)
        assert lib.hmr_gpu_sao_stats_frame(s.ctx, C.byref(f_org), C.byref(f_rec), d_stats) == 0
        out["stats"] = s.down(d_stats, stats0)
        # SAO: dst starts as a copy of the deblocked picture
        d_dst = [s.up(s.down(d, hp)) for d, hp in zip(d_rec, host_pre)]
        f_dst = frame_of(d_dst, shapes)
        assert lib.hmr_gpu_sao_apply_frame(s.ctx, C.byref(f_rec), C.byref(f_dst), s.up(np.ascontiguousarray(case["sao_params"]))) == 0
        out["sao"] = crop(d_dst)
        assert lib.hmr_gpu_pad_frame(s.ctx, C.byref(f_dst), C.c_int(PAD_X), C.c_int(PAD_Y)) == 0
        out["padded"] = [s.down(d, hp) for d, hp in zip(d_dst, host_pre)]
        return out
    finally:
        s.close()


def compare(got, exp, width, height):
    """Assert bit-exact equality on everything inside the picture."""
    u4, w4 = height // 4, width // 4
    for k in ("flags", "bs_ver", "bs_hor"):
        a, b = got[k][:u4, :w4].copy(), exp[k][:u4, :w4].copy()
        if k == "bs_ver":
            a[:, 1::2] = 0; b[:, 1::2] = 0
        if k == "bs_hor":
            a[1::2, :] = 0; b[1::2, :] = 0
        assert np.array_equal(a, b), k
    for k in ("deblocked", "sao", "padded"):
        for i, (a, b) in enumerate(zip(got[k], exp[k])):
            assert np.array_equal(a, b), (k, i, int((a != b).sum()))
    assert np.array_equal(got["stats"], exp["stats"]), "stats"
