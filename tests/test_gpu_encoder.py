"""-m gpu: the device-resident CTU encoder (include/homer_gpu.h section 12) through the C ABI against the per-CTU fixtures
minted from the compiled reference: side-info arrays, motion vectors, levels, the pre-filter reconstruction and the single worker
thread's mode buffers after every CTU of an I frame and the P frames that follow it (outside-picture window content excluded,
encoder_cases.crop_recon)."""
import ctypes as C

import numpy as np
import pytest

import encoder_cases as ec
import libs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    lib = libs.load_gpu()
    lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_frame_ctus.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int] + [C.c_char_p] * 3 + [C.c_double, C.c_char_p]
    lib.hmr_gpu_enc_destroy.argtypes = [C.c_void_p]
    lib.hmr_gpu_enc_last_ctu_ms.restype = C.c_float
    lib.hmr_gpu_enc_last_ctu_ms.argtypes = [C.c_void_p]
    lib.hmr_gpu_last_error.restype = C.c_char_p
    ctx = C.c_void_p()
    rc = lib.hmr_gpu_create(C.byref(ctx), 0, None)
    assert rc == 0, lib.hmr_gpu_last_error()
    assert lib.hmr_gpu_enc_record_bytes() == ec.REC
    lib._ctx = ctx
    return lib


def run_fixture(lib, name):
    fx = ec.load_fixture(name)
    w, h, frames = int(fx["width"]), int(fx["height"]), int(fx["frames"])
    nctu = ((w + 63) // 64) * ((h + 63) // 64)
    cfg = ec.default_cfg(w, h)
    enc = C.c_void_p()
    rc = lib.hmr_gpu_enc_create(lib._ctx, C.byref(cfg), C.byref(enc))
    assert rc == 0, lib.hmr_gpu_last_error()
    bad, ms = [], []
    ysz = w * h
    recs = C.create_string_buffer(ec.REC * nctu)
    for f, planes in enumerate(ec.clip_frames(w, h, frames)):
        refs = [None, None, None]
        if f:
            r = fx[f"f{f - 1}_recon"].tobytes()
            refs = [r[:ysz], r[ysz:ysz + ysz // 4], r[ysz + ysz // 4:]]
        st = lib.hmr_gpu_enc_frame_ctus(enc, *planes, 0, *refs, -1.0, recs)
        assert st == (2 if f == 0 else 1), lib.hmr_gpu_last_error()
        ms.append(lib.hmr_gpu_enc_last_ctu_ms(enc))
        bad += ec.check_frame_against_fixture(fx, f, recs.raw, w, h)
    lib.hmr_gpu_enc_destroy(enc)
    print(name, "CTU kernel ms per frame:", ["%.2f" % m for m in ms])
    return bad


@pytest.mark.parametrize("name", ["ctus_200x136", "ctus_416x240"])
def test_ctu_encoder_matches_reference_fixture(gpu, name):
    bad = run_fixture(gpu, name)
    assert not bad, f"{len(bad)} mismatches\n" + "\n".join(bad[:12])


def test_unsupported_configuration_is_refused(gpu):
    cfg = ec.default_cfg(416, 240, rd=1, bitrate_mode=1, bitrate=400)      # (RD_FULL under rate control)
    enc = C.c_void_p()
    assert gpu.hmr_gpu_enc_create(gpu._ctx, C.byref(cfg), C.byref(enc)) == -3
