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
