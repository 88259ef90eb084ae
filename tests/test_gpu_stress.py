"""-m gpu: repetition tests of the device encoder (the round-2 stress tools as tests).  Every run starts from fresh encoder objects, so anything that depends
on timing, on what a buffer held before, or on which workgroup picked which CTU (k_encode_pool hands the CTUs of a launch to whatever worker is free) shows up as
a stream that differs from the reference's fixture."""
import ctypes as C
import hashlib
import json
import os

import pytest

import encoder_cases as ec
import libs

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(ec.GOLDEN, "streams.json")))


@pytest.fixture(scope="module")
def gpu():
    lib = libs.load_gpu()
    lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.hmr_gpu_destroy.argtypes = [C.c_void_p]
    lib.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_source.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_long, C.POINTER(C.c_long), C.c_char_p]
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.hmr_gpu_enc_encode_batch_pipelined.argtypes = lib.hmr_gpu_enc_encode_batch.argtypes
    lib.hmr_gpu_enc_destroy.argtypes = [C.c_void_p]
    lib.hmr_gpu_last_error.restype = C.c_char_p
    return lib


def make(lib, case, clips):
    g = GOLD[case]
    keys = dict(g["keys"])
    cut_at = keys.pop("cut_at", None)
    ctx, enc = C.c_void_p(), C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0, lib.hmr_gpu_last_error()
    cfg = ec.default_cfg(g["width"], g["height"], **keys)
    assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
    if case not in clips:
        clips[case] = ec.clip_frames(g["width"], g["height"], g["frames"], cut_at)
    for f, planes in enumerate(clips[case]):
        assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0, lib.hmr_gpu_last_error()
    return ctx, enc, g["frames"]


def drop(lib, ctx, enc):
    lib.hmr_gpu_enc_destroy(enc)
    lib.hmr_gpu_destroy(ctx)


def run_batch(lib, cases, clips):
    made = [make(lib, c, clips) for c in cases]
    bufs = [C.create_string_buffer(4 << 20) for _ in cases]
    md5 = [hashlib.md5() for _ in cases]
    for f in range(max(m[2] for m in made)):
        live = [i for i, m in enumerate(made) if f < m[2]]
        k = len(live)
        e_arr = (C.c_void_p * k)(*[made[i][1] for i in live])
        ptrs = (C.c_char_p * k)(*[C.cast(bufs[i], C.c_char_p) for i in live])
        caps = (C.c_long * k)(*[len(bufs[i]) for i in live])
        got = (C.c_long * k)()
        assert lib.hmr_gpu_enc_encode_batch(e_arr, k, (C.c_int * k)(*([f] * k)), None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        for j, i in enumerate(live):
            md5[i].update(C.string_at(bufs[i], got[j]))
    for m in made:
        drop(lib, m[0], m[1])
    return [m.hexdigest() for m in md5]


def run_batch_pipelined(lib, cases, clips):
    """the same through the pipelined call: call k delivers the access units of call k - 1; the list is flushed before it changes (a sequence has ended) and at the end"""
    made = [make(lib, c, clips) for c in cases]
    bufs = [C.create_string_buffer(4 << 20) for _ in cases]
    md5 = [hashlib.md5() for _ in cases]
    prev = None

    def call(live, f):
        k = len(live)
        e_arr = (C.c_void_p * k)(*[made[i][1] for i in live])
        ptrs = (C.c_char_p * k)(*[C.cast(bufs[i], C.c_char_p) for i in live])
        caps = (C.c_long * k)(*[len(bufs[i]) for i in live])
        got = (C.c_long * k)()
        slots = None if f is None else (C.c_int * k)(*([f] * k))
        assert lib.hmr_gpu_enc_encode_batch_pipelined(e_arr, k, slots, None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        for j, i in enumerate(live):
            md5[i].update(C.string_at(bufs[i], got[j]))
        return [got[j] for j in range(k)]

    for f in range(max(m[2] for m in made)):
        live = [i for i, m in enumerate(made) if f < m[2]]
        if prev is not None and live != prev:
            assert all(call(prev, None))
        got = call(live, f)
        assert all(got) if prev == live else not any(got)       # nothing is delivered by the first call after a flush
        prev = live
    assert all(call(prev, None))
    for m in made:
        drop(lib, m[0], m[1])
    return [m.hexdigest() for m in md5]


def test_mixed_batch_pipelined_three_times(gpu):
    """the pipelined call (download and entropy coding of a step under the next step's CTU launch) delivers the same access units, one call later"""
    cases = ["416x240_wpp_rows", "832x480_wpp_rows", "416x240_scene_cut_wpp_rows", "328x264_wpp3", "416x240_eng3_wpp_rows", "832x480_eng2_wpp_rows"]
    clips = {}
    for it in range(3):
        assert run_batch_pipelined(gpu, cases, clips) == [GOLD[c]["stream_md5"] for c in cases], f"round {it}"


def test_outstanding_access_units_are_not_dropped(gpu):
    """an encoder whose access unit is still to be delivered is refused by the other encode calls until the flush"""
    clips = {}
    made = [make(gpu, "416x240_wpp_rows", clips) for _ in range(2)]
    bufs = [C.create_string_buffer(1 << 20) for _ in made]
    e_arr = (C.c_void_p * 2)(*[m[1] for m in made])
    ptrs = (C.c_char_p * 2)(*[C.cast(b, C.c_char_p) for b in bufs])
    caps = (C.c_long * 2)(*[len(b) for b in bufs])
    got, n = (C.c_long * 2)(), C.c_long()
    slots = (C.c_int * 2)(0, 0)
    assert gpu.hmr_gpu_enc_encode_batch_pipelined(e_arr, 2, slots, None, ptrs, caps, got) == 0 and not any(got)
    assert gpu.hmr_gpu_enc_encode_batch(e_arr, 2, slots, None, ptrs, caps, got) < 0 and b"outstanding" in gpu.hmr_gpu_last_error()
    assert gpu.hmr_gpu_enc_encode_source(made[1][1], 0, 0, bufs[1], len(bufs[1]), C.byref(n), None) < 0 and b"outstanding" in gpu.hmr_gpu_last_error()
    swapped = (C.c_void_p * 2)(made[1][1], made[0][1])
    assert gpu.hmr_gpu_enc_encode_batch_pipelined(swapped, 2, slots, None, ptrs, caps, got) < 0
    assert gpu.hmr_gpu_enc_encode_batch_pipelined(e_arr, 2, None, None, ptrs, caps, got) == 0 and all(got)
    first = [C.string_at(bufs[i], got[i]) for i in range(2)]
    assert first[0] == first[1] and len(first[0]) > 100
    for m in made:
        drop(gpu, m[0], m[1])


def test_twenty_fresh_encoders_one_after_the_other(gpu):
    """832x480, one WPP thread per CTU row: the size at which round 2 found an allocation race one run in twenty"""
    clips = {}
    buf, n = C.create_string_buffer(4 << 20), C.c_long()
    for it in range(20):
        ctx, enc, frames = make(gpu, "832x480_wpp_rows", clips)
        md5 = hashlib.md5()
        for f in range(frames):
            assert gpu.hmr_gpu_enc_encode_source(enc, f, 0, buf, len(buf), C.byref(n), None) in (1, 2), gpu.hmr_gpu_last_error()
            md5.update(C.string_at(buf, n.value))
        drop(gpu, ctx, enc)
        assert md5.hexdigest() == GOLD["832x480_wpp_rows"]["stream_md5"], f"encoder {it}"


def test_mixed_batch_five_times(gpu):
    """pictures of different sizes, lengths and thread counts (one with a scene cut, one with two engines) share the launches of a batch; five rounds from scratch"""
    cases = ["416x240_wpp_rows", "832x480_wpp_rows", "416x240_scene_cut_wpp_rows", "328x264_wpp3", "416x240_eng3_wpp_rows", "832x480_eng2_wpp_rows"]
    clips = {}
    for it in range(5):
        assert run_batch(gpu, cases, clips) == [GOLD[c]["stream_md5"] for c in cases], f"round {it}"


def test_2160p_batch_four_times(gpu):
    cases = ["3840x2160_cfg2_wpp32"] * 6
    clips = {}
    for it in range(4):
        assert run_batch(gpu, cases, clips) == [GOLD[c]["stream_md5"] for c in cases], f"round {it}"
