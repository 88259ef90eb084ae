"""GPU parity of the frame-level in-loop kernels through the C ABI: bit-exact against the CPU oracle and against the
golden vectors minted from the reference's own deblock / SAO / padding code."""
import ctypes as C

import numpy as np
import pytest

import frame_cases as fc
import golden_io
import libs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    return libs.load_gpu()


def test_frames_match_oracle_and_goldens(gpu, oracle):
    for case, exp, meta in golden_io.load_frame_goldens():
        got = fc.run_gpu(gpu, case)
        ora = fc.run_oracle(oracle, case)
        fc.compare(got, ora, case["width"], case["height"])
        W, H = case["width"], case["height"]
        for k in ("deblocked", "sao"):
            for i in range(3):
                assert np.array_equal(got[k][i], exp[k][i]), (meta["tag"], k, i)
        assert np.array_equal(got["stats"], exp["stats"]), meta["tag"]
        for i in range(3):   # golden padded planes have the reference's window width
            assert np.array_equal(got["padded"][i], exp["padded"][i]), (meta["tag"], "padded", i)


def test_deblock_rejects_unaligned_geometry(gpu):
    s = fc.GpuSession(gpu)
    try:
        f = fc.Frame(width=100, height=64, y=0, u=0, v=0, stride_y=100, stride_c=50)
        u = fc.Units()
        assert gpu.hmr_gpu_deblock_frame(s.ctx, C.byref(f), C.byref(u), 0, 0, 0, 0, None, None) != 0
    finally:
        s.close()


def test_get_sao_stats_dropin_per_ctu(gpu, oracle):
    """The table's own granularity: one CTU per call, host pointers (low_level_funcs_t.get_sao_stats)."""
    case, exp, meta = golden_io.load_frame_goldens()[2]
    ora = fc.run_oracle(oracle, case)
    W, H = case["width"], case["height"]
    org = [np.ascontiguousarray(p) for p in case["orig"]]
    rec = [np.ascontiguousarray(p) for p in ora["deblocked"]]
    P3, I3 = C.c_void_p * 3, C.c_int * 3
    strides = I3(W, W // 2, W // 2)
    ctus_x = (W + 63) // 64
    for ctu in range(ctus_x * ((H + 63) // 64)):
        out = np.zeros((3, 5, 2, 32), np.int64)
        gpu.hmr_gpu_get_sao_stats(P3(*[p.ctypes.data for p in org]), strides, P3(*[p.ctypes.data for p in rec]), strides, W, H,
                                  (ctu % ctus_x) * 64, (ctu // ctus_x) * 64, out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out, ora["stats"][ctu].astype(np.int64)), ctu


def test_loop_filters_ctu_by_ctu_dropins(gpu, oracle):
    """The reference's own call granularity (one CTU per call, host pointers): deblocking direction by direction, SAO offset and border
    padding issued CTU by CTU must give the frame-level (oracle / reference golden) result."""
    for case, exp, meta in golden_io.load_frame_goldens():
        ora = fc.run_oracle(oracle, case)
        W, H = case["width"], case["height"]
        info = {k: np.ascontiguousarray(v) for k, v in case["info"].items()}
        W4 = info["qp"].shape[1]
        flags = np.ascontiguousarray(((info["pred_mode"] != 0) * 1 + (info["cbf_y"] != 0) * 2).astype(np.uint8))
        ctus = [(x, y) for y in range(0, H, 64) for x in range(0, W, 64)]
        P3, I3 = C.c_void_p * 3, C.c_int * 3
        # deblocking in padded windows, like the encoder's reference frame
        pl = []
        for p, pad in zip(case["pre"], (fc.PAD_X, fc.PAD_X // 2, fc.PAD_X // 2)):
            h, w = p.shape
            win = np.full((h + 2 * pad, w + 2 * pad), 0x1234, np.int16)
            win[pad:pad + h, pad:pad + w] = p
            pl.append(win)
        def origin(win, pad):
            return win.ctypes.data + 2 * (pad * win.shape[1] + pad)
        planes = P3(origin(pl[0], fc.PAD_X), origin(pl[1], fc.PAD_X // 2), origin(pl[2], fc.PAD_X // 2))
        strides = I3(pl[0].shape[1], pl[1].shape[1], pl[2].shape[1])
        vp = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        for d in (0, 1):
            for x, y in ctus:
                gpu.hmr_gpu_deblock_filter_ctu(planes, strides, W, H, W4, vp(info["mvx"]), vp(info["mvy"]), vp(info["ref_idx"]), vp(info["qp"]), vp(flags),
                                               vp(info["pred_depth"]), vp(info["tr_idx"]), x, y, 64, d, *[int(v) for v in case["dbk"]])
        inner = lambda win, pad: win[pad:win.shape[0] - pad, pad:win.shape[1] - pad]   # noqa: E731
        for i, pad in enumerate((fc.PAD_X, fc.PAD_X // 2, fc.PAD_X // 2)):
            assert np.array_equal(inner(pl[i], pad), ora["deblocked"][i]), (meta["tag"], "deblock", i)
            assert np.array_equal(inner(pl[i], pad), exp["deblocked"][i]), (meta["tag"], "deblock golden", i)
            outside = pl[i].copy()
            inner(outside, pad)[...] = 0x1234
            assert np.all(outside == 0x1234), (meta["tag"], "deblock wrote outside the picture", i)
        # SAO offset: source = deblocked copy, destination = the frame
        src = [w.copy() for w in pl]
        sp = P3(origin(src[0], fc.PAD_X), origin(src[1], fc.PAD_X // 2), origin(src[2], fc.PAD_X // 2))
        params = np.ascontiguousarray(case["sao_params"])
        for k, (x, y) in enumerate(ctus):
            gpu.hmr_gpu_sao_offset_ctu(sp, strides, planes, strides, W, H, x, y, vp(params[k]))
        for i, pad in enumerate((fc.PAD_X, fc.PAD_X // 2, fc.PAD_X // 2)):
            assert np.array_equal(inner(pl[i], pad), ora["sao"][i]), (meta["tag"], "sao", i)
            assert np.array_equal(inner(pl[i], pad), exp["sao"][i]), (meta["tag"], "sao golden", i)
        # border padding
        for x, y in ctus:
            gpu.hmr_gpu_pad_ctu(planes, strides, W, H, fc.PAD_X, fc.PAD_Y, x, y, 64)
        for i in range(3):
            assert np.array_equal(pl[i], ora["padded"][i]), (meta["tag"], "pad", i)


def synthetic_case(width, height, seed):
    """Full-size picture with a synthetic coding tree / motion field (the shape bench.py uses) - no reference run is needed, the C
    oracle is fast enough to be the checker at BASELINE.json's picture sizes."""
    rng = np.random.default_rng(seed)
    ctus_x, ctus_y = (width + 63) // 64, (height + 63) // 64
    W4, H4 = ctus_x * 16, ctus_y * 16
    blk = lambda a, n: np.kron(a, np.ones((n, n), a.dtype))   # noqa: E731
    depth = blk(rng.integers(0, 4, (ctus_y * 2, ctus_x * 2)).astype(np.uint8), 8)            # per 32x32
    tr = blk((rng.random((H4 // 2, W4 // 2)) < 0.3).astype(np.uint8), 2)
    intra = blk((rng.random((H4 // 4, W4 // 4)) < 0.1).astype(np.uint8), 4)
    info = {
        "mvx": blk(rng.integers(-40, 41, (H4 // 2, W4 // 2)).astype(np.int16), 2), "mvy": blk(rng.integers(-24, 25, (H4 // 2, W4 // 2)).astype(np.int16), 2),
        "ref_idx": np.where(intra, -1, 0).astype(np.int8), "qp": blk(rng.integers(26, 38, (H4 // 4, W4 // 4)).astype(np.uint8), 4),
        "pred_mode": intra, "cbf_y": blk((rng.random((H4 // 2, W4 // 2)) < 0.5).astype(np.uint8), 2), "pred_depth": depth, "tr_idx": tr,
    }
    info = {k: np.ascontiguousarray(v) for k, v in info.items()}
    yy, xx = np.mgrid[0:height, 0:width]
    y = np.clip(128 + 60 * np.sin(xx / 53.0) * np.cos(yy / 41.0) + rng.integers(-12, 13, (height, width)), 0, 255).astype(np.int16)
    u = np.clip(128 + 30 * np.sin(xx[::2, ::2] / 97.0) + rng.integers(-3, 4, (height // 2, width // 2)), 0, 255).astype(np.int16)
    v = np.clip(128 + 30 * np.cos(yy[::2, ::2] / 89.0) + rng.integers(-3, 4, (height // 2, width // 2)), 0, 255).astype(np.int16)
    return {"width": width, "height": height, "info": info, "pre": fc.blocky_planes(y, u, v, seed + 1), "orig": [y, u, v],
            "sao_params": fc.random_sao_params(ctus_x * ctus_y, seed + 2), "dbk": [2, 2, int(rng.integers(-2, 3)), int(rng.integers(-2, 3))]}


@pytest.mark.parametrize("width,height", [(1920, 1080), (3840, 2160)], ids=["1080p", "2160p"])
def test_full_size_pictures_match_oracle(gpu, oracle, width, height):
    """BASELINE.json's picture sizes: edge flags, boundary strengths, deblocking, SAO statistics, SAO offset and border padding of a whole
    picture (incl. the partial last CTU row) bit-exact against the CPU oracle."""
    case = synthetic_case(width, height, 5 + width)
    got = fc.run_gpu(gpu, case)
    ora = fc.run_oracle(oracle, case)
    fc.compare(got, ora, width, height)
    # size-independent property: padding is edge replication of the SAO output
    for pl, pad in zip(got["padded"], (fc.PAD_X, fc.PAD_X // 2, fc.PAD_X // 2)):
        inner = pl[pad:-pad, pad:-pad]
        assert np.array_equal(pl, np.pad(inner, pad, mode="edge"))


def test_units_from_ctus(gpu, oracle):
    """a3 on the device: the encoder's per-CTU z-order side-info re-ordered into the raster arrays the in-loop kernels take."""
    rng = np.random.default_rng(3)
    ctus_x, ctus_y = 7, 3
    n = ctus_x * ctus_y * 256
    src = {"mvx": rng.integers(-200, 201, n).astype(np.int16), "mvy": rng.integers(-200, 201, n).astype(np.int16), "ref_idx": rng.integers(-1, 2, n).astype(np.int8),
           "qp": rng.integers(10, 50, n).astype(np.uint8), "pred_mode": rng.integers(0, 3, n).astype(np.uint8), "cbf_y": rng.integers(0, 8, n).astype(np.uint8),
           "pred_depth": rng.integers(0, 4, n).astype(np.uint8), "tr_idx": rng.integers(0, 3, n).astype(np.uint8)}
    us = ctus_x * 16 + 5
    shape = (ctus_y * 16, us)
    exp = {k: np.full(shape, 77, d) for k, d in (("mvx", np.int16), ("mvy", np.int16), ("ref", np.int8), ("qp", np.uint8), ("flags", np.uint8), ("pd", np.uint8), ("tr", np.uint8))}
    vp = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    oracle.ora_units_from_ctus(*[vp(src[k]) for k in ("mvx", "mvy", "ref_idx", "qp", "pred_mode", "cbf_y", "pred_depth", "tr_idx")], ctus_x, ctus_y, us,
                               *[vp(exp[k]) for k in ("mvx", "mvy", "ref", "qp", "flags", "pd", "tr")])
    s = fc.GpuSession(gpu)
    try:
        class CtuUnits(C.Structure):
            _fields_ = [(k, C.c_void_p) for k in ("mvx", "mvy", "ref_idx", "qp", "pred_mode", "cbf_y", "pred_depth", "tr_idx")]
        cu = CtuUnits(*[s.up(src[k]) for k in ("mvx", "mvy", "ref_idx", "qp", "pred_mode", "cbf_y", "pred_depth", "tr_idx")])
        dev = {k: s.up(np.full(shape, 77, v.dtype)) for k, v in exp.items()}
        units = fc.Units(us, dev["mvx"], dev["mvy"], dev["ref"], dev["qp"], dev["flags"])
        assert gpu.hmr_gpu_units_from_ctus(s.ctx, C.byref(cu), ctus_x, ctus_y, C.byref(units), dev["pd"], dev["tr"]) == 0
        assert gpu.hmr_gpu_sync(s.ctx) == 0
        for k in exp:
            assert np.array_equal(s.down(dev[k], exp[k]), exp[k]), k
    finally:
        s.close()
