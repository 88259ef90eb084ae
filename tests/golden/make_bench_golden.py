#!/usr/bin/env python3
"""Mint bench.py's self-check fixture from the COMPILED REFERENCE (build container only).

bench.py hashes every stream it produces; whatever --steps / --warmup it is run with, the digest after frame k has to be the
reference's digest after frame k.  For each bench workload this script encodes the synthetic clip once with the reference
(oracle/_ref/ref_lockstep for one thread; oracle/_ref/ref_ctudump under HOMER_TURNSTILE for one thread per CTU row, with
engines = E its engine turnstile as well) and records the md5 of the stream's first k access units for every k:
tests/golden/bench_md5.json  {workload: {"frames": N, "cumulative_md5": [md5 after AU 0, after AU 1, ...]}}.

usage: make_bench_golden.py [workload ...]        (no arguments: all)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_yuv  # noqa: E402

# name -> (width, height, frames, keys)  - the names are bench.py's WORKLOADS
CASES = {
    "cfg2-1080p-encode": (1920, 1080, 40, {"wpp": 17}),
    "cfg2-1080p-encode-single-thread-order": (1920, 1080, 40, {}),
    "cfg2-2160p-encode": (3840, 2160, 10, {"wpp": 32}),
    "cfg2-416x240-encode": (416, 240, 40, {"wpp": 4}),
    # bench.py --gpus N: the same 1080p encode with one engine per GPU (num_enc_engines = N), the reference pinned by the engine turnstile
    "cfg2-1080p-encode-engines2": (1920, 1080, 40, {"wpp": 17, "engines": 2}),
    "cfg2-1080p-encode-engines3": (1920, 1080, 40, {"wpp": 17, "engines": 3}),
    "cfg2-1080p-encode-engines4": (1920, 1080, 40, {"wpp": 17, "engines": 4}),
    "cfg2-1080p-encode-engines8": (1920, 1080, 40, {"wpp": 17, "engines": 8}),
    "cfg2-416x240-encode-engines2": (416, 240, 40, {"wpp": 4, "engines": 2}),
    # BASELINE.json configs[3]: 2160p with one engine per GPU
    "cfg2-2160p-encode-engines2": (3840, 2160, 10, {"wpp": 32, "engines": 2}),
    "cfg2-2160p-encode-engines4": (3840, 2160, 10, {"wpp": 32, "engines": 4}),
    "cfg2-2160p-encode-engines8": (3840, 2160, 34, {"wpp": 32, "engines": 8}),
    # BASELINE.json configs[2]: 2160p IPPP, CBR 20000 kbps, performance_mode 1
    "cfg3-2160p-cbr": (3840, 2160, 10, {"wpp": 32, "bitrate_mode": 1, "bitrate": 20000, "perf": 1}),
    "cfg3-1080p-cbr": (1920, 1080, 24, {"wpp": 17, "bitrate_mode": 1, "bitrate": 5000, "perf": 1}),
    # BASELINE.json configs[4] as BASELINE.md realises it: 2160p all-intra, rd = 1 (full RDO), performance_mode 0, max_intra_tr_depth = 4, SAO on
    "cfg5-2160p-intra-rdfull": (3840, 2160, 8, {"wpp": 32, "force_intra": 1, "rd": 1, "intra_tr": 4, "perf": 0}),
    "cfg5-1080p-intra-rdfull": (1920, 1080, 12, {"wpp": 17, "force_intra": 1, "rd": 1, "intra_tr": 4, "perf": 0}),
}
# bench.py's batch encodes eight DIFFERENT clips side by side (tools/gen_yuv.py: seed 1234 is the published clip, the others differ in texture, pan, pattern and box path)
CLIP_SEEDS = [1234, 1, 2, 3, 4, 5, 6, 7]
for _seed in CLIP_SEEDS[1:]:
    CASES[f"cfg2-1080p-encode-seed{_seed}"] = (1920, 1080, 24, {"wpp": 17}, _seed)
    CASES[f"cfg2-2160p-encode-seed{_seed}"] = (3840, 2160, 6, {"wpp": 32}, _seed)
    # ... and in the single-thread order (bench.py's serial_order_batch: hmr_gpu_enc_create_serial_pool), minted by the plain ref_lockstep
    CASES[f"cfg2-1080p-encode-single-thread-order-seed{_seed}"] = (1920, 1080, 24, {}, _seed)


def access_unit_ends(stream):
    """byte offsets at which the access units of an Annex-B stream end (one slice per picture: an access unit ends with its VCL NAL unit)"""
    starts, i = [], 0
    while True:
        i = stream.find(b"\x00\x00\x01", i)
        if i < 0:
            break
        starts.append((i - 1 if i > 0 and stream[i - 1] == 0 else i, stream[i + 3]))   # (first byte of the start code, first header byte)
        i += 3
    ends = []
    for k, (pos, hdr) in enumerate(starts):
        if ((hdr >> 1) & 0x3f) < 32:                                                       # VCL
            ends.append(starts[k + 1][0] if k + 1 < len(starts) else len(stream))
    return ends


def run(width, height, frames, keys, seed=1234):
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "in.yuv")
        gen_yuv.write_clip(yuv, width, height, frames, seed)
        turnstile = int(keys.get("wpp", 1)) > 1 or int(keys.get("engines", 1)) > 1
        cmd = [os.path.join(ROOT, "oracle", "_ref", "ref_ctudump" if turnstile else "ref_lockstep"), yuv, os.path.join(tmp, "out.265"), str(width), str(height), str(frames)]
        cmd += [f"{k}={v}" for k, v in keys.items()]
        subprocess.run(cmd, check=True, timeout=7200, stdout=subprocess.DEVNULL, env=dict(os.environ, HOMER_TURNSTILE="1") if turnstile else None)
        stream = open(os.path.join(tmp, "out.265"), "rb").read()
    ends = access_unit_ends(stream)
    assert len(ends) == frames, (len(ends), frames)
    h, out, each, pos = hashlib.md5(), [], [], 0
    for e in ends:
        h.update(stream[pos:e])
        each.append(hashlib.md5(stream[pos:e]).hexdigest())
        pos = e
        out.append(h.copy().hexdigest())
    rec = {"width": width, "height": height, "frames": frames, "keys": keys, "cumulative_md5": out, "au_md5": each}
    if seed != 1234:
        rec["clip_seed"] = seed
    return rec


if __name__ == "__main__":
    # usage: make_bench_golden.py [-jN] [workload or prefix* ...]: N reference runs side by side (a 2160p run holds a few hundred MB)
    import threading
    from concurrent.futures import ThreadPoolExecutor
    path = os.path.join(HERE, "bench_md5.json")
    args = sys.argv[1:]
    jobs = 1
    for a in list(args):
        if a.startswith("-j"):
            jobs = int(a[2:])
            args.remove(a)
    out = json.load(open(path)) if os.path.exists(path) else {}
    lock = threading.Lock()

    def wanted(name):
        return not args or any(name == a or (a.endswith("*") and name.startswith(a[:-1])) for a in args)

    def one(name):
        rec = run(*CASES[name])
        with lock:
            out[name] = rec
            print(name, rec["frames"], rec["cumulative_md5"][-1], flush=True)
            json.dump(out, open(path, "w"), indent=1)

    with ThreadPoolExecutor(jobs) as ex:
        list(ex.map(one, [n for n in CASES if wanted(n)]))
