"""CPU tests of the frame encoder's decision core and host logic (no GPU): the one-lane checker build of
homerhevc_amd/csrc/enc (oracle/libenc_cpu.so) against the per-CTU fixtures minted from the compiled reference, and - where
the reference is present - against a fresh reference run at other sizes and settings."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import encoder_cases as ec
import libs

CPU_SO = os.path.join(libs.ORACLE_DIR, "libenc_cpu.so")


@pytest.fixture(scope="module")
def cpu():
    subprocess.check_call(["make", "-s", "-C", libs.ORACLE_DIR, CPU_SO])
    lib = C.CDLL(CPU_SO)
    lib.henc_cpu_create.restype = C.c_void_p
    lib.henc_cpu_create.argtypes = [C.POINTER(ec.EncCfg)]
    lib.henc_cpu_frame_ctus.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int] + [C.c_char_p] * 3 + [C.c_double, C.c_int, C.c_int]
    lib.henc_cpu_records.restype = C.POINTER(C.c_uint8)
    lib.henc_cpu_records.argtypes = [C.c_void_p]
    lib.henc_cpu_destroy.argtypes = [C.c_void_p]
    assert lib.henc_cpu_record_bytes() == ec.REC
    return lib


def run_fixture(lib, name):
    fx = ec.load_fixture(name)
    w, h, frames = int(fx["width"]), int(fx["height"]), int(fx["frames"])
    nctu = ((w + 63) // 64) * ((h + 63) // 64)
    cfg = ec.default_cfg(w, h)
    enc = lib.henc_cpu_create(C.byref(cfg))
    assert enc
    bad = []
    ysz = w * h
    for f, planes in enumerate(ec.clip_frames(w, h, frames)):
        refs = [None, None, None]
        if f:
            r = fx[f"f{f - 1}_recon"].tobytes()
            refs = [r[:ysz], r[ysz:ysz + ysz // 4], r[ysz + ysz // 4:]]
        st = lib.henc_cpu_frame_ctus(enc, *planes, 0, *refs, -1.0, 0, -1)
        assert st == (2 if f == 0 else 1)
        bad += ec.check_frame_against_fixture(fx, f, C.string_at(lib.henc_cpu_records(enc), ec.REC * nctu), w, h)
    lib.henc_cpu_destroy(enc)
    return bad


@pytest.mark.parametrize("name", ["ctus_200x136", "ctus_416x240"])
def test_core_matches_reference_fixture(cpu, name):
    bad = run_fixture(cpu, name)
    assert not bad, "\n".join(bad[:10])


def test_unsupported_configurations_are_refused(cpu):
    for kw in ({"rd": 1, "wpp": 4, "bitrate_mode": 1, "bitrate": 400}, {"rd": 1, "bitrate_mode": 1, "bitrate": 400}, {"bitrate_mode": 3},      # (RD_FULL: fixed QP; with one thread or a thread per row since round 6)
               {"num_b": 1, "gop_size": 2}, {"cu_size": 32},      # (rate control with several engines is accepted since round 6 ...
               {"rd": 1, "engines": 2, "wpp": 4},             # ... RD_FULL with several engines is not: its replay matches the reference on most configurations only)      # (performance_mode 3 is accepted since round 5)
               {"wpp": 2},      # 7 CTU columns: two threads would have to be in rows 0 and 2 at once under the synchronous wavefront
               {"wpp": 5}):     # more threads than the 4 CTU rows
        cfg = ec.default_cfg(416, 240, **kw)
        assert not cpu.henc_cpu_create(C.byref(cfg)), kw
    cfg = ec.default_cfg(136, 456, sao=0)      # three CTU columns x four or more rows: refused with SAO off as well
    assert not cpu.henc_cpu_create(C.byref(cfg))
    for size in ((64, 256), (192, 256), (320, 320)):      # one CTU wide; SAO on grids of at most five columns with at least as many rows
        cfg = ec.default_cfg(*size)
        assert not cpu.henc_cpu_create(C.byref(cfg)), size
    # grids the compiled reference itself cannot run: two CTU columns x several rows (crash); several engines on fewer than nine columns with more than four rows (deadlock)
    for size, kw in (((128, 128), {}), ((120, 88), {"sao": 0}), ((320, 320), {"sao": 0, "engines": 2}), ((512, 384), {"engines": 2}), ((448, 576), {"engines": 3, "wpp": 9})):
        cfg = ec.default_cfg(*size, **kw)
        assert not cpu.henc_cpu_create(C.byref(cfg)), (size, kw)
    for size, kw in (((128, 64), {}), ((576, 384), {"engines": 2}), ((512, 256), {"engines": 4}),
                     ((416, 240), {"wpp": 4}), ((328, 264), {"wpp": 3}), ((320, 320), {"sao": 0}), ((416, 240), {"bitrate_mode": 1, "bitrate": 400}), ((416, 240), {"bitrate_mode": 2, "bitrate": 400, "wpp": 4}), ((416, 240), {"rd": 1, "wpp": 4, "intra_tr": 4}), ((416, 240), {"rd": 1}), ((416, 240), {"rd": 1, "intra_tr": 4, "perf": 0})):
        cfg = ec.default_cfg(*size, **kw)
        assert cpu.henc_cpu_create(C.byref(cfg)), (size, kw)


@pytest.mark.parametrize("size,keys", [((264, 200), {}), ((416, 240), {"perf": 0}), ((416, 240), {"perf": 1, "qp": 26}), ((328, 264), {"sign_hiding": 0, "qp": 38})])
def test_core_matches_fresh_reference_run(cpu, size, keys):
    """other sizes / settings straight against the compiled reference (build container only)"""
    if not os.path.exists(os.path.join(libs.ORACLE_DIR, "_ref", "ref_ctudump")):
        pytest.skip("oracle/_ref not built (reference sources are only present in the build container)")
    import sys
    sys.path.insert(0, os.path.join(ec.ROOT, "tools"))
    import ctu_diff
    import tempfile
    w, h = size
    with tempfile.TemporaryDirectory() as tmp:
        yuv = ctu_diff.run_reference(tmp, w, h, 3, {k: str(v) for k, v in keys.items()})
        ref = open(os.path.join(tmp, "ctus.bin"), "rb").read()
        rec = open(os.path.join(tmp, "rec.yuv"), "rb").read()
        src = open(yuv, "rb").read()
    nx = (w + 63) // 64
    nctu = nx * ((h + 63) // 64)
    fsz, ysz = w * h * 3 // 2, w * h
    cfg = ec.default_cfg(w, h, **keys)
    enc = cpu.henc_cpu_create(C.byref(cfg))
    for f in range(3):
        fr = src[f * fsz:(f + 1) * fsz]
        refs = [None, None, None]
        if f:
            pr = rec[(f - 1) * fsz:f * fsz]
            refs = [pr[:ysz], pr[ysz:ysz + ysz // 4], pr[ysz + ysz // 4:]]
        cpu.henc_cpu_frame_ctus(enc, fr[:ysz], fr[ysz:ysz + ysz // 4], fr[ysz + ysz // 4:], 0, *refs, -1.0, 0, -1)
        mine = C.string_at(cpu.henc_cpu_records(enc), ec.REC * nctu)
        assert mine == ref[f * nctu * ec.REC:(f + 1) * nctu * ec.REC], f"frame {f}: records differ (serial run: every byte, mode buffers included)"
    cpu.henc_cpu_destroy(enc)
