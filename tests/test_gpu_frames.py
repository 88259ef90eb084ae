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
