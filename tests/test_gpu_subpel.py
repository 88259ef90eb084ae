"""-m gpu: the phase planes of a reference picture (hmr_gpu_subpel_planes, include/homer_gpu.h section 13) against the pinned oracle's
motion compensation (ora_mc_luma / ora_mc_chroma, which tests/test_oracle_vs_ref.py and tests/test_motion.py hold equal to the compiled
reference's hmr_motion_compensation_* and interpolation kernels): every sample of every plane must be what the reference would predict
for a block at that position with a vector of that phase."""
import ctypes as C

import numpy as np
import pytest

import libs

pytestmark = pytest.mark.gpu
VP = C.c_void_p


@pytest.fixture(scope="module")
def rig():
    gpu, ora = libs.load_gpu(), libs.load_oracle()
    gpu.hmr_gpu_create.argtypes = [C.POINTER(VP), C.c_int, VP]
    gpu.hmr_gpu_malloc.argtypes = [VP, C.POINTER(VP), C.c_size_t]
    gpu.hmr_gpu_upload.argtypes = [VP, VP, VP, C.c_size_t]
    gpu.hmr_gpu_download.argtypes = [VP, VP, VP, C.c_size_t]
    gpu.hmr_gpu_free.argtypes = [VP, VP]
    gpu.hmr_gpu_subpel_planes.argtypes = [VP] * 4 + [C.c_int] * 4 + [VP] * 3
    gpu.hmr_gpu_last_error.restype = C.c_char_p
    ora.ora_mc_luma.argtypes = [VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    ora.ora_mc_chroma.argtypes = [VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    ctx = VP()
    assert gpu.hmr_gpu_create(C.byref(ctx), 0, None) == 0, gpu.hmr_gpu_last_error()
    return gpu, ora, ctx


def dev(gpu, ctx, arr):
    p = VP()
    assert gpu.hmr_gpu_malloc(ctx, C.byref(p), C.c_size_t(arr.nbytes)) == 0
    assert gpu.hmr_gpu_upload(ctx, p, VP(arr.ctypes.data), C.c_size_t(arr.nbytes)) == 0
    return p


@pytest.mark.parametrize("stride_y,rows_y,seed", [(208, 72, 1), (368, 100, 2), (2080, 40, 3)])
def test_planes_equal_oracle_motion_compensation(rig, stride_y, rows_y, seed):
    gpu, ora, ctx = rig
    rng = np.random.default_rng(seed)
    stride_c, rows_c = stride_y // 2 + 4, rows_y // 2 + 3            # any layout: the planes take the picture's own
    pic = [rng.integers(0, 256, (rows_y, stride_y)).astype(np.int16), rng.integers(0, 256, (rows_c, stride_c)).astype(np.int16),
           rng.integers(0, 256, (rows_c, stride_c)).astype(np.int16)]
    pic[0][: rows_y // 2, : stride_y // 2] = 255                      # a flat saturated area and hard edges: the clipping paths
    pic[1][rows_c // 3:, stride_c // 3:] = 0
    d_pic = [dev(gpu, ctx, p) for p in pic]
    out = [np.zeros((rows_y, 16, stride_y), np.uint8), np.zeros((rows_c, 64, stride_c), np.uint8), np.zeros((rows_c, 64, stride_c), np.uint8)]      # row-interleaved planes
    d_out = [dev(gpu, ctx, o) for o in out]
    assert gpu.hmr_gpu_subpel_planes(ctx, *d_pic, stride_y, rows_y, stride_c, rows_c, *d_out) == 0, gpu.hmr_gpu_last_error()
    for o, d in zip(out, d_out):
        assert gpu.hmr_gpu_download(ctx, VP(o.ctypes.data), d, C.c_size_t(o.nbytes)) == 0
    # luma: the interior the filter taps reach inside the allocation (the oracle reads through row ends linearly too, so whole rows compare)
    w, h = stride_y - 8, rows_y - 8
    want = np.zeros((h, w), np.int16)
    n = 8
    hv, wv = (h // n) * n, (w // n) * n
    for f in range(16):
        fx, fy = f & 3, f >> 2
        src = pic[0].ctypes.data + 2 * (4 * stride_y + 4)
        for y0 in range(0, hv, n):
            for x0 in range(0, wv, n):
                ora.ora_mc_luma(VP(src + 2 * (y0 * stride_y + x0)), stride_y, VP(want.ctypes.data + 2 * (y0 * w + x0)), w, n, n, fx, fy, 0)
        bad = np.argwhere(out[0][4:4 + hv, f, 4:4 + wv] != want[:hv, :wv].astype(np.uint8))
        assert len(bad) == 0, f"luma plane {f}: {len(bad)} samples differ, first at {bad[:4].tolist()}"
    wc, hc = stride_c - 8, rows_c - 8
    wantc = np.zeros((hc, wc), np.int16)
    n = 8
    for comp in (1, 2):
        for f in range(64):
            fx, fy = f & 7, f >> 3
            src = pic[comp].ctypes.data + 2 * (4 * stride_c + 4)
            for y0 in range(0, hc - n + 1, n):
                for x0 in range(0, wc - n + 1, n):
                    ora.ora_mc_chroma(VP(src + 2 * (y0 * stride_c + x0)), stride_c, VP(wantc.ctypes.data + 2 * (y0 * wc + x0)), wc, n, fx, fy, 0)
            hv, wv = (hc // n) * n, (wc // n) * n
            bad = np.argwhere(out[comp][4:4 + hv, f, 4:4 + wv] != wantc[:hv, :wv].astype(np.uint8))
            assert len(bad) == 0, f"chroma {comp} plane {f}: {len(bad)} samples differ, first at {bad[:4].tolist()}"
    for d in d_pic + d_out:
        gpu.hmr_gpu_free(ctx, d)
