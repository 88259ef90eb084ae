"""CPU tests of the whole frame pipeline as the product runs it, minus the device: the one-lane checker build of the CTU core
(oracle/libenc_cpu.so), the round-1 oracle's in-loop filters, and the product's host entropy stage (homerhevc_amd/csrc/enc/enc_entropy.h:
SAO decision, CABAC, parameter sets, Annex-B) free running over several frames.  The .265 bytes and every reconstructed picture must
equal what the compiled reference produced (tests/golden/streams.json, minted by tests/golden/make_stream_golden.py)."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import pytest

import decoder_check
import encoder_cases as ec
import libs

CPU_SO = os.path.join(libs.ORACLE_DIR, "libenc_cpu.so")
GOLD = json.load(open(os.path.join(ec.GOLDEN, "streams.json")))


@pytest.fixture(scope="module")
def cpu():
    subprocess.check_call(["make", "-s", "-C", libs.ORACLE_DIR, CPU_SO])
    lib = C.CDLL(CPU_SO)
    lib.henc_cpu_create.restype = C.c_void_p
    lib.henc_cpu_create.argtypes = [C.POINTER(ec.EncCfg)]
    lib.henc_cpu_encode_frame.restype = C.c_long
    lib.henc_cpu_encode_frame.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int, C.c_char_p, C.c_long, C.c_char_p]
    lib.henc_cpu_set_sched.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.henc_cpu_sched_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    lib.henc_cpu_destroy.argtypes = [C.c_void_p]
    return lib


def encode(lib, case, sched=0, raw_recon=None):
    g = GOLD[case]
    w, h, frames = g["width"], g["height"], g["frames"]
    keys = dict(g["keys"])
    cut_at = keys.pop("cut_at", None)
    clip_seed = keys.pop("clip_seed", 1234)
    image_type = 3 if keys.pop("force_intra", 0) else 0          # encoder_in_out_t.image_type: IMAGE_I on every frame
    cfg = ec.default_cfg(w, h, **keys)
    enc = lib.henc_cpu_create(C.byref(cfg))
    assert enc
    lib.henc_cpu_set_sched(enc, sched, 1)
    buf = C.create_string_buffer(4 << 20)
    rec = C.create_string_buffer(w * h * 3 // 2)
    stream, recon = b"", []
    for planes in ec.clip_frames(w, h, frames, cut_at, clip_seed):
        n = lib.henc_cpu_encode_frame(enc, *planes, image_type, buf, len(buf), rec)
        assert n > 0
        stream += buf.raw[:n]
        recon.append(hashlib.md5(rec.raw).hexdigest())
        if raw_recon is not None:
            raw_recon.append(rec.raw)
    st = (C.c_int * 3)()
    lib.henc_cpu_sched_stats(enc, st, 0)
    lib.henc_cpu_destroy(enc)
    return stream, recon, list(st)


@pytest.mark.parametrize("case", ["200x136", "416x240", "416x240_nosao", "416x240_qp22_perf0", "328x264_qp38_nosbh", "416x240_intra_period1", "832x480", "392x136_qp22_clip931814", "400x104_qp22_perf0_nosao_wpp_rows_clip657909", "1280x720_intra_period1", "1280x720_force_intra", "416x240_force_intra", "416x240_force_intra_wpp_rows", "416x240_force_intra_rdfull_wpp_rows", "416x240_force_intra_rdfull_tr4_wpp_rows", "328x264_force_intra_rdfull_tr3_perf0_wpp_rows", "416x240_rdfull_wpp_rows", "832x480_rdfull_tr3_wpp_rows", "200x136_scene_cut", "416x240_scene_cut", "416x240_wpp_rows", "416x240_scene_cut_wpp_rows", "832x480_wpp_rows", "416x240_qp22_perf0_wpp_rows", "416x240_nosao_wpp_rows", "328x264_wpp3", "200x136_wpp2", "416x240_eng2", "416x240_eng3_wpp_rows", "832x480_eng2_wpp_rows", "416x240_scene_cut_eng2_wpp_rows",
                                  "416x240_force_intra_rdfull_tr4", "416x240_rdfull", "328x264_force_intra_rdfull_tr3_perf0", "416x240_vbr400",
                                  "416x240_cbr400_perf1_eng2_wpp_rows", "416x240_vbr400_eng3_wpp_rows", "416x240_cbr300_eng2", "832x480_cbr1500_perf1_eng4_wpp_rows",
                                  "416x240_cbr400_perf1", "416x240_cbr400_perf1_wpp_rows", "416x240_vbr400_wpp_rows", "832x480_cbr1500_perf1_wpp_rows", "416x240_cbr300_nosao_wpp_rows", "416x240_qp4", "416x240_perf3", "416x240_perf3_wpp_rows", "416x240_force_intra_perf3_wpp_rows", "416x240_scene_cut_perf3_wpp_rows", "832x480_qp26_perf3_rdfull_wpp_rows"])
def test_stream_is_byte_identical_to_the_reference(cpu, case):
    raw = []
    stream, recon, _ = encode(cpu, case, raw_recon=raw)
    g = GOLD[case]
    assert len(stream) == g["stream_bytes"]
    assert hashlib.md5(stream).hexdigest() == g["stream_md5"]
    assert recon == g["recon_md5"]
    decoder_check.check(stream, g, case, raw)      # ... and a decoder reconstructs from it what the reference encoder reconstructed (tests/decoder_check.py)


def test_stream_matches_the_ctu_fixture_stream(cpu):
    """the same bytes the per-CTU fixture carries (ref_ctudump's own .265)"""
    fx = ec.load_fixture("ctus_200x136")
    stream, _, _ = encode(cpu, "200x136")
    assert stream == fx["stream"].tobytes()


@pytest.mark.parametrize("case", ["200x136", "416x240", "416x240_qp22_perf0", "200x136_scene_cut", "416x240_scene_cut"])
def test_row_parallel_schedule_reproduces_the_single_thread_stream(cpu, case):
    """the device's schedule (row workers with guessed inputs, raster-order verification, selective re-encode; enc_sched.h) emulated with one lane"""
    stream, recon, st = encode(cpu, case, sched=1)
    g = GOLD[case]
    assert hashlib.md5(stream).hexdigest() == g["stream_md5"]
    assert recon == g["recon_md5"]
    assert st[0] >= g["frames"] and st[1] >= g["frames"]   # at least one pass and nctu encodes per frame


def test_1080p_cfg2_one_thread_per_row(cpu):
    """the same configuration with wfpp_num_threads = 17 (one WPP thread per CTU row) against the reference forced into the synchronous-wavefront schedule"""
    stream, recon, _ = encode(cpu, "1920x1080_cfg2_wpp_rows")
    assert hashlib.md5(stream).hexdigest() == GOLD["1920x1080_cfg2_wpp_rows"]["stream_md5"]
    assert recon == GOLD["1920x1080_cfg2_wpp_rows"]["recon_md5"]


def test_1080p_cfg2_stream_md5(cpu):
    """BASELINE.json configs[1]: 1920x1080 IPPP, QP 32, 8 frames -> the reference's 2f0c3447..."""
    stream, recon, _ = encode(cpu, "1920x1080_cfg2")
    assert hashlib.md5(stream).hexdigest() == GOLD["1920x1080_cfg2"]["stream_md5"] == "2f0c3447dabb6fbd87cac9821bb479fd"
    assert recon == GOLD["1920x1080_cfg2"]["recon_md5"]
