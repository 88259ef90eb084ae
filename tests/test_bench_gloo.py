"""The N > 1 path of bench.py on CPU: world_size 2 over gloo.  bench.measure is the timing contract (warm-up, barrier + device sync on both
sides of the timed steps, wall time = MAX over the ranks); here two ranks run it with a stand-in step of different length per rank and a
per-rank "encoder" that is the one-lane checker build, so the ranks really encode independent sequences (replicas, as on the GPUs)."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import ctypes as C, hashlib, json, os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
import bench, encoder_cases as ec
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = C.CDLL(os.path.join({root!r}, "oracle", "libenc_cpu.so"))
lib.henc_cpu_create.restype = C.c_void_p
lib.henc_cpu_create.argtypes = [C.POINTER(ec.EncCfg)]
lib.henc_cpu_encode_frame.restype = C.c_long
lib.henc_cpu_encode_frame.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int, C.c_char_p, C.c_long, C.c_char_p]
cfg = ec.default_cfg(200, 136)
enc = lib.henc_cpu_create(C.byref(cfg))
frames = ec.clip_frames(200, 136, 3)
buf = C.create_string_buffer(1 << 20)
md5 = hashlib.md5()
own = []
def step(f):
    t0 = time.perf_counter()
    n = lib.henc_cpu_encode_frame(enc, *frames[f], 0, buf, len(buf), None)
    md5.update(buf.raw[:n])
    time.sleep(0.2 * rank)            # rank 1 is the slow one
    own.append(time.perf_counter() - t0)
dt = bench.measure(step, 1, 3, world, lambda: None, "cpu")
print(json.dumps({{"rank": rank, "dt": dt, "own": sum(own[1:]), "md5": md5.hexdigest()}}), flush=True)
dist.destroy_process_group()
"""


def test_two_rank_timing_is_the_max_over_ranks(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "libenc_cpu.so")])
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0
        outs.append(json.loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    # both ranks report the same wall time, and it covers the slow rank's two timed steps
    assert abs(outs[0]["dt"] - outs[1]["dt"]) < 1e-9
    assert outs[0]["dt"] >= outs[1]["own"] - 0.02 and outs[0]["dt"] >= 0.4
    # independent replicas of the same sequence: identical streams
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "streams.json")))["200x136"]
    assert outs[0]["md5"] == outs[1]["md5"] == gold["stream_md5"]


RING_WORKER = r"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch, torch.distributed as dist
import bench, encoder_cases as ec
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
w, h, keys = bench.WORKLOADS["cfg2-416x240-encode"]
keys = dict(keys, engines=world)
lib = C.CDLL(os.path.join({root!r}, "oracle", "libenc_cpu.so"))
lib.henc_cpu_create_engine.restype = C.c_void_p
lib.henc_cpu_create_engine.argtypes = [C.POINTER(ec.EncCfg), C.c_int]
lib.henc_cpu_encode_frame.restype = C.c_long
lib.henc_cpu_encode_frame.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int, C.c_char_p, C.c_long, C.c_char_p]
lib.henc_cpu_reference_elems.restype = C.c_long
lib.henc_cpu_reference_elems.argtypes = [C.c_void_p, C.c_int]
lib.henc_cpu_export_reference.argtypes = [C.c_void_p] * 5
lib.henc_cpu_import_reference.argtypes = [C.c_void_p] * 5

class CpuEngines:          # the adapter EngineRing asks for, over the checker build (test infrastructure: bench.py itself only knows GpuEngines)
    on_cpu = True
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
    def row_bytes(self):
        return 2 * (sum(self.elems) + (self.state_bytes + 1) // 2)
    def new_buffer(self, rows):
        return torch.zeros((max(rows, 1), self.row_bytes // 2), dtype=torch.int16)
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

a = argparse.Namespace(workload="cfg2-416x240-encode", warmup=1, steps=4, sequences=2, no_pipeline=True, no_cpu_baseline=True, gpus=world)
out = bench.run_engine_ring(a, world, rank, 0, torch, adapter=CpuEngines())
if rank == 0:
    print(json.dumps(out), flush=True)
dist.destroy_process_group()
"""


def test_bench_engine_ring_entry_at_world_2(tmp_path):
    """bench.py's own N > 1 entry (run_engine_ring: frame dealing, exchange, per-access-unit check against the reference's num_enc_engines = 2 digests, the JSON line)
    with two ranks over gloo and the checker build behind the adapter"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "libenc_cpu.so")])
    script = tmp_path / "ring_worker.py"
    script.write_text(RING_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0
        outs.append(out)
    line = json.loads(outs[0].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 4 and line["warmup"] == 1
    assert line["config"]["workload"] == "cfg2-416x240-encode-engines2" and line["config"]["sequences_per_gpu"] == 2 and line["config"]["sequences"] == 4
    assert line["stream_matches_reference"] is True and line["config"]["stream_matches_reference"] is True
    assert line["access_units_produced"] == 4 * 5 and line["access_units_checked_against_reference"] == 20 and line["access_units_differing"] == 0
    assert line["value"] > 0 and abs(line["value"] - 4 * 4 / (line["ms_per_step"] * 4 / 1e3)) < 1e-2 * line["value"]


def test_gpus_flag_must_agree_with_the_launcher():
    """--gpus N under a launcher that set another WORLD_SIZE is refused (it used to be ignored: a `--gpus 8` run on one rank labelled itself n_gpus 1)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(os.environ, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
