"""-m gpu: the multi-GPU product path on ONE GPU.  bench.py --gpus N drives homerhevc_amd.engines.EngineRing + GpuEngines: one engine object per sequence and rank
(hmr_gpu_enc_create_engine), frame t of sequence s on rank (s + t) mod E, every rank's frames of a step as one batch launch, then the reconstructed pictures (8-bit,
without margins: hmr_gpu_enc_export_references8 / hmr_gpu_enc_import_references8, widened and padded on arrival) and the frame-to-frame scalars handed to the next
rank.  Here the E ranks are E EngineRing objects in one process on one GPU and the transfer is a loop-back copy between their device buffers instead of an RCCL
send / recv - everything else is the code the driver's scaling run executes.  The streams must be the ones the compiled reference produces with
num_enc_engines = E under oracle/ref_ctudump.c's engine turnstile (tests/golden/streams.json)."""
import hashlib
import json
import os

import pytest

import encoder_cases as ec

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(ec.GOLDEN, "streams.json")))
# bench.py's fixtures hold the digest after every access unit: the first frames of its 1080p four-engine stream as one more case
_B = json.load(open(os.path.join(ec.GOLDEN, "bench_md5.json")))["cfg2-1080p-encode-engines4"]
GOLD["1920x1080_cfg2_eng4"] = {"width": 1920, "height": 1080, "frames": 9, "keys": _B["keys"], "stream_md5": _B["cumulative_md5"][8]}


def run_ring(case, sequences, pipelined, flush_after=None):
    from homerhevc_amd.engines import EngineRing, GpuEngines, engine_of
    g = GOLD[case]
    w, h, frames, keys = g["width"], g["height"], g["frames"], dict(g["keys"])
    cut_at = keys.pop("cut_at", None)
    world = keys["engines"]
    rings = []

    def loop_back(ring, send_rows, recv_rows):
        # rank r's send buffer is rank r + 1's receive buffer of the next step: a device-to-device copy stands in for the RCCL transfer
        rings[(ring.rank + 1) % world].recv_buf[:send_rows.shape[0]].copy_(send_rows)

    adapters = [GpuEngines(lambda seq: ec.default_cfg(w, h, **keys), 0, pipelined=pipelined) for _ in range(world)]
    for r in range(world):
        rings.append(EngineRing(adapters[r], sequences, r, world, exchange=loop_back))
    clip = ec.clip_frames(w, h, frames, cut_at)
    for ring in rings:
        ring.load_sources(clip)
    aus = {}
    for t in range(frames):
        last = t + 1 == frames
        for ring in rings:
            for s, au in ring.step_encode(t, last).items():
                aus[(s, ring.delivered)] = au
                assert engine_of(s, ring.delivered, world) == ring.rank
        if not last:
            import torch
            for ring in rings:
                ring.step_exchange(t)
            torch.cuda.synchronize()
        if flush_after is not None and t == flush_after:      # (bench.py empties the pipelines between its warm-up and its timed steps)
            for ring in rings:
                for f, units in ring.flush():
                    for s, au in units.items():
                        aus[(s, f)] = au
    for ring in rings:
        for f, units in ring.flush():
            for s, au in units.items():
                aus[(s, f)] = au
    for ring, ad in zip(rings, adapters):
        for hnd in ring.enc.values():
            ad.destroy(hnd)
    assert len(aus) == sequences * frames
    return [hashlib.md5(b"".join(aus[(s, t)] for t in range(frames))).hexdigest() for s in range(sequences)], g["stream_md5"]


@pytest.mark.parametrize("case,sequences,pipelined", [("832x480_eng2_wpp_rows", 3, False), ("416x240_eng3_wpp_rows", 4, True), ("416x240_eng4_wpp_rows", 5, True), ("416x240_eng4_wpp_rows", 8, False), ("1920x1080_cfg2_eng2", 2, True), ("1920x1080_cfg2_eng3", 3, False), ("1920x1080_cfg2_eng4", 8, True)])
def test_engine_ring_on_one_gpu_reproduces_the_reference_engine_stream(case, sequences, pipelined):
    md5s, gold = run_ring(case, sequences, pipelined)
    assert md5s == [gold] * sequences


@pytest.mark.parametrize("case,sequences,flush_after", [("416x240_eng4_wpp_rows", 8, 2), ("416x240_eng3_wpp_rows", 4, 3), ("832x480_eng2_wpp_rows", 3, 2)])
def test_pipelines_emptied_in_the_middle_of_a_run(case, sequences, flush_after):
    """bench.py flushes every rank's pipelines after the warm-up steps: the frames after that must not notice"""
    md5s, gold = run_ring(case, sequences, True, flush_after)
    assert md5s == [gold] * sequences
