"""-m gpu: HOMER_enc_encode on the device through the C ABI (hmr_gpu_enc_encode, include/homer_gpu.h section 12), free running over
several frames: CTU decisions on the row-parallel schedule, deblocking, SAO statistics / decision / offsets and padding as kernels, CABAC
on the host.  The .265 bytes and every reconstructed picture must equal what the compiled reference produced
(tests/golden/streams.json), including BASELINE.json configs[1] at full size (1920x1080, 8 frames, md5 2f0c3447...)."""
import ctypes as C
import hashlib
import json
import os

import pytest

import decoder_check
import encoder_cases as ec
import libs

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(ec.GOLDEN, "streams.json")))


@pytest.fixture(scope="module")
def gpu():
    lib = libs.load_gpu()
    lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_encode.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int, C.c_char_p, C.c_long, C.POINTER(C.c_long), C.c_char_p]
    lib.hmr_gpu_enc_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.hmr_gpu_enc_destroy.argtypes = [C.c_void_p]
    lib.hmr_gpu_last_error.restype = C.c_char_p
    ctx = C.c_void_p()
    rc = lib.hmr_gpu_create(C.byref(ctx), 0, None)
    assert rc == 0, lib.hmr_gpu_last_error()
    lib._ctx = ctx
    return lib


def encode(lib, case, raw_recon=None):
    g = GOLD[case]
    w, h, frames = g["width"], g["height"], g["frames"]
    keys = dict(g["keys"])
    cut_at = keys.pop("cut_at", None)
    clip_seed = keys.pop("clip_seed", 1234)
    image_type = 3 if keys.pop("force_intra", 0) else 0          # encoder_in_out_t.image_type: IMAGE_I on every frame
    cfg = ec.default_cfg(w, h, **keys)
    enc = C.c_void_p()
    rc = lib.hmr_gpu_enc_create(lib._ctx, C.byref(cfg), C.byref(enc))
    assert rc == 0, lib.hmr_gpu_last_error()
    buf = C.create_string_buffer(4 << 20)
    rec = C.create_string_buffer(w * h * 3 // 2)
    nbytes = C.c_long()
    stream, recon, log = b"", [], []
    for f, planes in enumerate(ec.clip_frames(w, h, frames, cut_at, clip_seed)):
        st = lib.hmr_gpu_enc_encode(enc, *planes, image_type, buf, len(buf), C.byref(nbytes), rec)
        assert st in (1, 2), lib.hmr_gpu_last_error()
        stream += buf.raw[:nbytes.value]
        recon.append(hashlib.md5(rec.raw).hexdigest())
        if raw_recon is not None:
            raw_recon.append(rec.raw)
        p, n, ms, tot = C.c_int(), C.c_int(), C.c_float(), C.c_float()
        lib.hmr_gpu_enc_last_stats(enc, C.byref(p), C.byref(n), C.byref(ms), C.byref(tot))
        log.append(f"f{f}: {p.value} passes {n.value} encodes {ms.value:.1f}/{tot.value:.1f} ms")
    lib.hmr_gpu_enc_destroy(enc)
    print(case, "; ".join(log))
    return stream, recon


@pytest.mark.parametrize("case", ["200x136", "416x240", "416x240_nosao", "416x240_qp22_perf0", "328x264_qp38_nosbh", "416x240_intra_period1", "832x480", "392x136_qp22_clip931814", "400x104_qp22_perf0_nosao_wpp_rows_clip657909", "1920x1080_cfg2", "1280x720_intra_period1", "1280x720_force_intra", "416x240_force_intra", "416x240_force_intra_wpp_rows", "416x240_force_intra_rdfull_wpp_rows", "416x240_force_intra_rdfull_tr4_wpp_rows", "328x264_force_intra_rdfull_tr3_perf0_wpp_rows", "416x240_rdfull_wpp_rows", "832x480_rdfull_tr3_wpp_rows", "3840x2160_cfg2", "200x136_scene_cut", "416x240_scene_cut", "416x240_wpp_rows", "416x240_scene_cut_wpp_rows", "1920x1080_cfg2_wpp_rows", "832x480_wpp_rows", "416x240_qp22_perf0_wpp_rows", "416x240_nosao_wpp_rows", "328x264_wpp3", "200x136_wpp2", "416x240_eng3_wpp_rows", "832x480_eng2_wpp_rows", "416x240_scene_cut_eng2_wpp_rows", "1920x1080_cfg2_eng2", "1920x1080_cfg2_eng3", "3840x2160_cfg2_eng8", "3840x2160_cfg2_wpp32", "3840x2160_force_intra_rdfull_tr4",
                                  "416x240_cbr400_perf1_wpp_rows", "416x240_vbr400_wpp_rows", "832x480_cbr1500_perf1_wpp_rows", "416x240_cbr300_nosao_wpp_rows", "416x240_qp4", "416x240_perf3", "416x240_perf3_wpp_rows", "416x240_force_intra_perf3_wpp_rows", "416x240_scene_cut_perf3_wpp_rows", "832x480_qp26_perf3_rdfull_wpp_rows", "1920x1080_cbr5000_perf1_wpp_rows",
                                  "3840x2160_cbr20000_perf1_wpp32",
                                  # one WPP thread under rate control / RD_FULL (the pool's raster schedule): the reference's deterministic single-thread streams
                                  "416x240_cbr400_perf1", "416x240_vbr400", "416x240_force_intra_rdfull_tr4", "416x240_rdfull", "328x264_force_intra_rdfull_tr3_perf0",
                                  "3840x2160_cbr20000_perf1", "3840x2160_force_intra_rdfull_tr4_perf0",
                                  # rate control with several engines
                                  "416x240_cbr400_perf1_eng2_wpp_rows", "416x240_vbr400_eng3_wpp_rows", "832x480_cbr1500_perf1_eng4_wpp_rows"])
def test_device_stream_is_byte_identical_to_the_reference(gpu, case):
    raw = []
    stream, recon = encode(gpu, case, raw_recon=raw)
    g = GOLD[case]
    first_bad = next((f for f in range(g["frames"]) if recon[f] != g["recon_md5"][f]), None)
    assert first_bad is None, f"reconstructed picture {first_bad} differs"
    assert len(stream) == g["stream_bytes"]
    assert hashlib.md5(stream).hexdigest() == g["stream_md5"]
    # the decoder-side check: what the device wrote decodes (sub-stream ends, entry points, ranges) to the pictures the device reconstructed (tests/decoder_check.py)
    decoder_check.check(stream, g, case, raw)


def test_batch_of_sequences_in_one_launch(gpu):
    """hmr_gpu_enc_encode_batch: four sequences of different sizes and lengths (one with a scene cut, one with fewer threads than rows; row-per-thread schedule)
    advance frame by frame with ONE launch for all their CTU stages; every sequence's stream must be the one it gets when encoded alone, i.e. the turnstile
    reference's."""
    lib = gpu
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    cases = ["416x240_wpp_rows", "832x480_wpp_rows", "416x240_scene_cut_wpp_rows", "328x264_wpp3", "832x480_cbr1500_perf1_wpp_rows", "416x240_cbr300_nosao_wpp_rows"]
    encs, ctxs, frames = [], [], []
    for case in cases:
        g = GOLD[case]
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0, lib.hmr_gpu_last_error()      # a context (stream) of its own per sequence
        keys = dict(g["keys"])
        cut_at = keys.pop("cut_at", None)
        cfg = ec.default_cfg(g["width"], g["height"], **keys)
        assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        for f, planes in enumerate(ec.clip_frames(g["width"], g["height"], g["frames"], cut_at)):
            assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0, lib.hmr_gpu_last_error()
        encs.append(enc); ctxs.append(ctx); frames.append(g["frames"])
    bufs = [C.create_string_buffer(1 << 20) for _ in cases]
    out = [b"" for _ in cases]
    for f in range(max(frames)):
        live = [i for i in range(len(cases)) if f < frames[i]]
        n = len(live)
        e_arr = (C.c_void_p * n)(*[encs[i] for i in live])
        slots = (C.c_int * n)(*([f] * n))
        ptrs = (C.c_char_p * n)(*[C.cast(bufs[i], C.c_char_p) for i in live])
        caps = (C.c_long * n)(*[len(bufs[i]) for i in live])
        got = (C.c_long * n)()
        assert lib.hmr_gpu_enc_encode_batch(e_arr, n, slots, None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        for k, i in enumerate(live):
            out[i] += bufs[i].raw[:got[k]]
    for i, case in enumerate(cases):
        assert hashlib.md5(out[i]).hexdigest() == GOLD[case]["stream_md5"], (i, case)
        lib.hmr_gpu_enc_destroy(encs[i])
    # the single-thread order is not a batch schedule
    cfg = ec.default_cfg(416, 240)
    enc = C.c_void_p()
    assert lib.hmr_gpu_enc_create(ctxs[0], C.byref(cfg), C.byref(enc)) == 0
    planes = next(iter(ec.clip_frames(416, 240, 1)))
    assert lib.hmr_gpu_enc_load_source(enc, 0, *planes) == 0
    e_arr, slots, ptrs, caps, got = (C.c_void_p * 1)(enc), (C.c_int * 1)(0), (C.c_char_p * 1)(C.cast(bufs[0], C.c_char_p)), (C.c_long * 1)(len(bufs[0])), (C.c_long * 1)()
    assert lib.hmr_gpu_enc_encode_batch(e_arr, 1, slots, None, ptrs, caps, got) == -3
    lib.hmr_gpu_enc_destroy(enc)


def test_batch_of_300_sequences_in_one_launch(gpu):
    """more pictures in one launch than round 5's cap of 256 (BATCH_MAX = 512: the pool's state rows, the staging arrays and the gather buffer are sized by it): 300
    sequences of the 200x136_wpp2 fixture, every stream the fixture's; 513 are refused"""
    lib = gpu
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    case, n = "200x136_wpp2", 300
    g = GOLD[case]
    clip = list(ec.clip_frames(g["width"], g["height"], g["frames"], None))
    cfg = ec.default_cfg(g["width"], g["height"], **g["keys"])
    encs = []
    for _ in range(n):
        enc = C.c_void_p()
        assert lib.hmr_gpu_enc_create(lib._ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        for f, planes in enumerate(clip):
            assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0, lib.hmr_gpu_last_error()
        encs.append(enc)
    bufs = [C.create_string_buffer(1 << 16) for _ in range(n)]
    out = [b""] * n
    e_arr = (C.c_void_p * n)(*encs)
    ptrs = (C.c_char_p * n)(*[C.cast(b, C.c_char_p) for b in bufs])
    caps = (C.c_long * n)(*[len(b) for b in bufs])
    got = (C.c_long * n)()
    for f in range(g["frames"]):
        assert lib.hmr_gpu_enc_encode_batch(e_arr, n, (C.c_int * n)(*([f] * n)), None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        for i in range(n):
            out[i] += bufs[i].raw[:got[i]]
    assert {hashlib.md5(o).hexdigest() for o in out} == {g["stream_md5"]}
    big = (C.c_void_p * 513)(*([encs[0]] * 513))
    assert lib.hmr_gpu_enc_encode_batch(big, 513, (C.c_int * 513)(), None, (C.c_char_p * 513)(), (C.c_long * 513)(), (C.c_long * 513)()) != 0
    for enc in encs:
        lib.hmr_gpu_enc_destroy(enc)


def test_serial_order_batch(gpu):
    """hmr_gpu_enc_create_serial_pool: sequences in the reference's deterministic single-thread order (wfpp_num_threads = 1) as ONE batch - the pool's raster schedule, a CTU
    at a time per picture.  The fixtures are the plain reference's (ref_lockstep, no pinned interleaving): fixed QP, RD_FAST, incl. a scene cut, forced intra pictures,
    performance_mode 0 and 3, and the one the merge shortcut of round 3 got wrong."""
    lib = gpu
    lib.hmr_gpu_enc_create_serial_pool.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    cases = ["416x240", "200x136_scene_cut", "416x240_qp22_perf0", "328x264_qp38_nosbh", "416x240_nosao", "416x240_force_intra", "392x136_qp22_clip931814", "416x240_perf3", "832x480"]
    encs, frames, itype = [], [], []
    for case in cases:
        g = GOLD[case]
        keys = dict(g["keys"])
        assert int(keys.get("wpp", 1)) == 1 and int(keys.get("engines", 1)) == 1
        cut_at, seed = keys.pop("cut_at", None), keys.pop("clip_seed", 1234)
        itype.append(3 if keys.pop("force_intra", 0) else 0)
        cfg = ec.default_cfg(g["width"], g["height"], **keys)
        enc = C.c_void_p()
        assert lib.hmr_gpu_enc_create_serial_pool(lib._ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        for f, planes in enumerate(ec.clip_frames(g["width"], g["height"], g["frames"], cut_at, seed)):
            assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0, lib.hmr_gpu_last_error()
        encs.append(enc); frames.append(g["frames"])
    bufs = [C.create_string_buffer(1 << 20) for _ in cases]
    out = [b"" for _ in cases]
    for f in range(max(frames)):
        live = [i for i in range(len(cases)) if f < frames[i]]
        n = len(live)
        got = (C.c_long * n)()
        assert lib.hmr_gpu_enc_encode_batch((C.c_void_p * n)(*[encs[i] for i in live]), n, (C.c_int * n)(*([f] * n)), (C.c_int * n)(*[itype[i] for i in live]),
                                            (C.c_char_p * n)(*[C.cast(bufs[i], C.c_char_p) for i in live]), (C.c_long * n)(*[len(bufs[i]) for i in live]), got) == 0, lib.hmr_gpu_last_error()
        for k, i in enumerate(live):
            out[i] += bufs[i].raw[:got[k]]
    for i, case in enumerate(cases):
        assert hashlib.md5(out[i]).hexdigest() == GOLD[case]["stream_md5"], case
        decoder_check.check(out[i], GOLD[case], case)
        lib.hmr_gpu_enc_destroy(encs[i])
    cfg = ec.default_cfg(416, 240, wpp=4)
    enc = C.c_void_p()
    assert lib.hmr_gpu_enc_create_serial_pool(lib._ctx, C.byref(cfg), C.byref(enc)) != 0


def test_stale_window_count_is_the_same_frame_by_frame_and_in_a_batch(gpu):
    """hmr_gpu_enc_stale_predictions (quirk Q12: merge candidates evaluated on what the thread's prediction window held) is the API's only indicator that byte identity may
    not hold: the batch call must count a picture once (it used to count it twice) and agree with the frame-by-frame count of the same clip - which has such evaluations."""
    lib = gpu
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.hmr_gpu_enc_stale_predictions.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    case = "400x104_qp22_perf0_nosao_wpp_rows_clip657909"
    g = GOLD[case]
    keys = dict(g["keys"])
    seed = keys.pop("clip_seed")
    clip = list(ec.clip_frames(g["width"], g["height"], g["frames"], None, seed))
    totals = []
    for batch in (False, True):
        enc = C.c_void_p()
        cfg = ec.default_cfg(g["width"], g["height"], **keys)
        assert lib.hmr_gpu_enc_create(lib._ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        buf, n = C.create_string_buffer(1 << 20), C.c_long()
        stream = b""
        for f, planes in enumerate(clip):
            if batch:
                assert lib.hmr_gpu_enc_load_source(enc, f, *planes) == 0
                got = (C.c_long * 1)()
                assert lib.hmr_gpu_enc_encode_batch((C.c_void_p * 1)(enc), 1, (C.c_int * 1)(f), None, (C.c_char_p * 1)(C.cast(buf, C.c_char_p)), (C.c_long * 1)(len(buf)), got) == 0, lib.hmr_gpu_last_error()
                stream += buf.raw[:got[0]]
            else:
                assert lib.hmr_gpu_enc_encode(enc, *planes, 0, buf, len(buf), C.byref(n), None) in (1, 2), lib.hmr_gpu_last_error()
                stream += buf.raw[:n.value]
        last, tot = C.c_long(), C.c_long()
        assert lib.hmr_gpu_enc_stale_predictions(enc, C.byref(last), C.byref(tot)) == 0
        totals.append(tot.value)
        assert hashlib.md5(stream).hexdigest() == g["stream_md5"]
        lib.hmr_gpu_enc_destroy(enc)
    assert totals[0] == totals[1] and totals[0] > 0, totals


def test_stale_window_count_is_kept_in_the_single_thread_order_too(gpu):
    """wfpp_num_threads = 1 goes through the other CTU kernel (k_encode_ctus); the counter is read there as well: -1 only before the first picture, then the picture's count
    (0 for this clip, whose stream is the reference's), and the total is the sum of the pictures' counts."""
    lib = gpu
    lib.hmr_gpu_enc_stale_predictions.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    g = GOLD["392x136_qp22_clip931814"]
    keys = dict(g["keys"])
    seed = keys.pop("clip_seed")
    enc = C.c_void_p()
    cfg = ec.default_cfg(g["width"], g["height"], **keys)
    assert cfg.wfpp_num_threads == 1
    assert lib.hmr_gpu_enc_create(lib._ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
    last, tot = C.c_long(7), C.c_long(7)
    assert lib.hmr_gpu_enc_stale_predictions(enc, C.byref(last), C.byref(tot)) == 0 and (last.value, tot.value) == (-1, 0)
    buf, n, stream, counts = C.create_string_buffer(1 << 20), C.c_long(), b"", []
    for planes in ec.clip_frames(g["width"], g["height"], g["frames"], None, seed):
        assert lib.hmr_gpu_enc_encode(enc, *planes, 0, buf, len(buf), C.byref(n), None) in (1, 2), lib.hmr_gpu_last_error()
        stream += buf.raw[:n.value]
        assert lib.hmr_gpu_enc_stale_predictions(enc, C.byref(last), C.byref(tot)) == 0
        counts.append(last.value)
        assert last.value >= 0 and tot.value == sum(counts)
    assert hashlib.md5(stream).hexdigest() == g["stream_md5"]
    lib.hmr_gpu_enc_destroy(enc)
