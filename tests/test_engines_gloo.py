"""Engine per rank on CPU: world_size 2, 3 and 4 over gloo.  homerhevc_amd.engines.EngineRing - the code bench.py --gpus N runs over RCCL - deals the frames
of several sequences to the ranks (frame t of sequence s on rank (s + t) mod E), every rank encodes its frames with an engine object of its own and hands the
reconstructed picture and the frame-to-frame scalars to the next rank.  Here the engine behind the adapter is the one-lane checker build
(oracle/libenc_cpu.so, test infrastructure); the streams that come out must be the ones the compiled reference produces with num_enc_engines = E under
oracle/ref_ctudump.c's engine turnstile (tests/golden/streams.json: 416x240_eng2, 416x240_eng3_wpp_rows)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import ctypes as C, hashlib, json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import encoder_cases as ec
from homerhevc_amd.engines import EngineRing, engine_of
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
case = json.load(open(os.path.join({root!r}, "tests", "golden", "streams.json")))[{case!r}]
w, h, frames, keys = case["width"], case["height"], case["frames"], dict(case["keys"])
assert keys["engines"] == world
lib = C.CDLL(os.path.join({root!r}, "oracle", "libenc_cpu.so"))
lib.henc_cpu_create_engine.restype = C.c_void_p
lib.henc_cpu_create_engine.argtypes = [C.POINTER(ec.EncCfg), C.c_int]
lib.henc_cpu_encode_frame.restype = C.c_long
lib.henc_cpu_encode_frame.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int, C.c_char_p, C.c_long, C.c_char_p]
lib.henc_cpu_reference_elems.restype = C.c_long
lib.henc_cpu_reference_elems.argtypes = [C.c_void_p, C.c_int]
lib.henc_cpu_export_reference.argtypes = [C.c_void_p] * 5
lib.henc_cpu_import_reference.argtypes = [C.c_void_p] * 5

class CpuEngines:          # the adapter EngineRing asks for, over the checker build
    def __init__(self):
        self.src, self.elems, self.state_bytes, self.buf = {{}}, None, lib.henc_cpu_state_bytes(), C.create_string_buffer(4 << 20)
    def create(self, seq, index):
        cfg = ec.default_cfg(w, h, **keys)
        hnd = lib.henc_cpu_create_engine(C.byref(cfg), index)
        assert hnd
        if self.elems is None:
            self.elems = [lib.henc_cpu_reference_elems(hnd, c) for c in range(3)]
        self.src[hnd] = {{}}
        return hnd
    @property
    def row_elems(self):
        return sum(self.elems) + (self.state_bytes + 1) // 2
    def new_buffer(self, rows):
        return torch.zeros((max(rows, 1), self.row_elems), dtype=torch.int16)
    def load_source(self, hnd, frame, planes):
        self.src[hnd][frame] = planes
    def encode(self, handles, frame):
        out = []
        for hnd in handles:
            n = lib.henc_cpu_encode_frame(hnd, *self.src[hnd][frame], 0, self.buf, len(self.buf), None)
            assert n > 0
            out.append(self.buf.raw[:n])
        return out
    def _ptrs(self, row):
        p = row.data_ptr()
        return p, p + 2 * self.elems[0], p + 2 * (self.elems[0] + self.elems[1]), p + 2 * sum(self.elems)
    def export(self, hnd, row):
        lib.henc_cpu_export_reference(hnd, *self._ptrs(row))
    def import_(self, hnd, row):
        lib.henc_cpu_import_reference(hnd, *self._ptrs(row))

S = {sequences}
ring = EngineRing(CpuEngines(), S, rank, world)
clip = ec.clip_frames(w, h, frames, keys.pop("cut_at", None))
ring.load_sources(clip)
mine = {{}}
for t in range(frames):
    for s, au in ring.step(t, last=t + 1 == frames).items():
        mine[(s, t)] = au
        assert engine_of(s, t, world) == rank
everything = [None] * world
dist.all_gather_object(everything, mine)
if rank == 0:
    aus = {{}}
    for part in everything:
        aus.update(part)
    print(json.dumps({{"md5": [hashlib.md5(b"".join(aus[(s, t)] for t in range(frames))).hexdigest() for s in range(S)], "per_rank": [len(p) for p in everything]}}), flush=True)
dist.destroy_process_group()
"""


@pytest.mark.parametrize("case,world,sequences,port", [("416x240_eng2", 2, 3, 29641), ("416x240_eng3_wpp_rows", 3, 2, 29643), ("416x240_eng4_wpp_rows", 4, 5, 29647),
                                                       ("416x240_cbr400_perf1_eng2_wpp_rows", 2, 3, 29651), ("416x240_vbr400_eng3_wpp_rows", 3, 2, 29653)])      # (rate control: its state travels with the frame scalars)
def test_engine_ring_reproduces_the_reference_engine_stream(tmp_path, case, world, sequences, port):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "libenc_cpu.so")])
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, case=case, sequences=sequences))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0
        outs.append(out)
    res = json.loads(outs[0].strip().splitlines()[-1])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "streams.json")))[case]
    assert res["md5"] == [gold["stream_md5"]] * sequences           # every sequence: the reference's engine-mode stream
    assert sum(res["per_rank"]) == sequences * gold["frames"] and min(res["per_rank"]) >= sequences * gold["frames"] // world - sequences   # the frames really were dealt round


def test_frame_dealing():
    from homerhevc_amd.engines import engine_index, engine_of
    for world in (1, 2, 3, 8):
        for s in range(5):
            for t in range(20):
                r = engine_of(s, t, world)
                assert t % world == engine_index(s, r, world)        # the engine object on that rank is the reference's engine t mod E
                assert engine_of(s, t + 1, world) == (r + 1) % world   # and the next frame is the next rank's
