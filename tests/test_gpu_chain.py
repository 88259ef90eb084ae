"""-m gpu: overlapping frames of one sequence (hmr_gpu_enc_encode_chain, include/homer_gpu.h section 12c): the E engine objects of a sequence encode E consecutive
frames in ONE launch of the CTU kernel - a frame's CTUs start as soon as the part of the previous frame's final picture their vectors can reach has been filtered,
padded and interpolated (the phase planes are tasks of the same launch) - which is the overlap the reference's engines have (encoder_engine_thread,
hmr_encoder_lib.c:3154-3211, :2393-2445).  The streams must be the ones the compiled reference produces with num_enc_engines = E under oracle/ref_ctudump.c's engine
turnstile (tests/golden/streams.json: its comment says "a device can overlap the frames as far as the reference rows it reads allow without changing any of this")."""
import ctypes as C
import hashlib
import json
import os
import time

import pytest

import encoder_cases as ec
import libs

pytestmark = pytest.mark.gpu
DUMPS = {}
GOLD = json.load(open(os.path.join(ec.GOLDEN, "streams.json")))


def encode_chained(lib, case, chain=None, sets=1):
    """the fixture's clip in chains of `chain` frames (default: sets x E); sets > 1: every engine has `sets` objects (twins sharing its persistent state), so a chain
    holds up to sets x E frames - an engine's next frame starts inside the same launch when its previous one is finished"""
    g = GOLD[case]
    w, h, frames, keys = g["width"], g["height"], g["frames"], dict(g["keys"])
    cut_at = keys.pop("cut_at", None)
    E = keys["engines"]
    chain = chain or E * sets
    assert chain <= E * sets
    lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.hmr_gpu_enc_create_engine.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.c_int, C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_create_engine_twin.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_load_source.argtypes = [C.c_void_p, C.c_int] + [C.c_char_p] * 3
    lib.hmr_gpu_enc_encode_chain.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.hmr_gpu_enc_destroy.argtypes = [C.c_void_p]
    lib.hmr_gpu_destroy.argtypes = [C.c_void_p]
    lib.hmr_gpu_last_error.restype = C.c_char_p
    cfg = ec.default_cfg(w, h, **keys)
    ctxs, encs = [], []
    for k in range(E * sets):          # object k: engine k % E, set k // E
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0, lib.hmr_gpu_last_error()
        if k < E:
            assert lib.hmr_gpu_enc_create_engine(ctx, C.byref(cfg), k, C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        else:
            assert lib.hmr_gpu_enc_create_engine_twin(ctx, encs[k % E], C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        ctxs.append(ctx)
        encs.append(enc)
    starts = list(range(0, frames, chain))
    if os.environ.get("CHAIN_FIRST"):
        k0 = int(os.environ["CHAIN_FIRST"])
        starts = [0] + list(range(k0, frames, chain))
    # which object encodes frame f, and in which of its picture slots the frame lies: position j of its chain -> set j // E
    obj_of, slot_of, used = {}, {}, [0] * len(encs)
    for si, first in enumerate(starts):
        for f in range(first, min(starts[si + 1] if si + 1 < len(starts) else frames, frames)):
            k = ((f - first) // E) * E + f % E
            obj_of[f], slot_of[f] = k, used[k]
            used[k] += 1
    for f, planes in enumerate(ec.clip_frames(w, h, frames, cut_at)):
        assert lib.hmr_gpu_enc_load_source(encs[obj_of[f]], slot_of[f], *planes) == 0, lib.hmr_gpu_last_error()
    bufs = [C.create_string_buffer(8 << 20) for _ in range(chain)]
    stream = b""
    t0 = time.time()
    for si, first in enumerate(starts):
        fs = list(range(first, min(starts[si + 1] if si + 1 < len(starts) else frames, frames)))
        n = len(fs)
        e_arr = (C.c_void_p * n)(*[encs[obj_of[f]] for f in fs])
        slots = (C.c_int * n)(*[slot_of[f] for f in fs])
        ptrs = (C.c_char_p * n)(*[C.cast(bufs[i], C.c_char_p) for i in range(n)])
        caps = (C.c_long * n)(*[len(bufs[i]) for i in range(n)])
        got = (C.c_long * n)()
        prev = encs[obj_of[first - 1]] if first else None
        assert lib.hmr_gpu_enc_encode_chain(e_arr, n, prev, slots, None, ptrs, caps, got) == 0, lib.hmr_gpu_last_error()
        for i in range(n):
            stream += bufs[i].raw[:got[i]]
        if os.environ.get("CHAIN_DUMP"):          # (debugging aid of tools/chain_probe.py: every frame's final picture as the object holds it after the call)
            import torch
            lib.hmr_gpu_enc_reference_bytes.restype = C.c_long
            lib.hmr_gpu_enc_reference_bytes.argtypes = [C.c_void_p]
            lib.hmr_gpu_enc_state_bytes.restype = C.c_int
            lib.hmr_gpu_enc_export_references8.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_long, C.c_void_p]
            nb = lib.hmr_gpu_enc_reference_bytes(encs[0])
            dev = torch.zeros((n, nb), dtype=torch.uint8, device="cuda:0")
            states = C.create_string_buffer(lib.hmr_gpu_enc_state_bytes() * n)
            assert lib.hmr_gpu_enc_export_references8(e_arr, n, dev.data_ptr(), nb, states) == 0, lib.hmr_gpu_last_error()
            torch.cuda.synchronize()
            for i, f in enumerate(fs):
                DUMPS.setdefault(chain if not os.environ.get("CHAIN_FIRST") else -chain, {})[f] = dev[i].cpu().numpy().copy()
    dt = time.time() - t0
    for enc in reversed(encs):          # (twins before the objects they were made from)
        lib.hmr_gpu_enc_destroy(enc)
    for ctx in ctxs:
        lib.hmr_gpu_destroy(ctx)
    print(f"{case}: {frames} frames in chains of {chain}: {frames / dt:.2f} frames/s")
    return stream, g


@pytest.mark.parametrize("case,chain,sets", [("416x240_eng3_wpp_rows", None, 1), ("832x480_eng2_wpp_rows", None, 1), ("416x240_scene_cut_eng2_wpp_rows", None, 1), ("1920x1080_cfg2_eng3", None, 1),
                                             ("1920x1080_cfg2_eng2", None, 1), ("3840x2160_cfg2_eng8", None, 1), ("3840x2160_cfg2_eng8", 3, 1),
                                             # an engine's next frame in the same launch, on a twin of its object
                                             ("416x240_eng3_wpp_rows", None, 2), ("416x240_eng3_wpp_rows", 7, 3), ("832x480_eng2_wpp_rows", None, 3), ("1920x1080_cfg2_eng3", None, 2),
                                             ("1920x1080_cfg2_eng2", None, 4)])
def test_overlapping_frames_reproduce_the_reference_engine_stream(case, chain, sets):
    lib = libs.load_gpu()
    stream, g = encode_chained(lib, case, chain, sets)
    assert len(stream) == g["stream_bytes"]
    assert hashlib.md5(stream).hexdigest() == g["stream_md5"]
