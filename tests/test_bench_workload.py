"""Host-side checks of the bench workload definition (no GPU): the clip generator and bench.py's own self-checks."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench      # noqa: E402
import gen_yuv    # noqa: E402


def test_clip_generator_matches_published_md5():
    """md5 of the generated clips (720p x 8 here; the 1080p / 2160p clips use the same code)."""
    md5 = hashlib.md5()
    for planes in gen_yuv.gen_frames(1280, 720, 8):
        for p in planes:
            md5.update(p.tobytes())
    assert md5.hexdigest().startswith("eba88043")
    # long clips (CPU baseline sample): the texture window wraps instead of running off its margin
    frames = list(gen_yuv.gen_frames(320, 192, 40))
    assert len(frames) == 40 and frames[39][0].shape == (192, 320)


def test_every_bench_workload_has_the_reference_digests_it_checks_against():
    """bench.py verifies every access unit it produces against tests/golden/bench_md5.json (minted from the compiled reference by tests/golden/make_bench_golden.py):
    the workloads it runs by default must be there, for all eight clips of the headline, with enough frames for the default --warmup / --steps"""
    for name in ("cfg2-1080p-encode", "cfg2-1080p-encode-single-thread-order", "cfg2-2160p-encode", "cfg3-2160p-cbr", "cfg5-2160p-intra-rdfull",
                 "cfg2-1080p-encode-engines2", "cfg2-1080p-encode-engines4", "cfg2-1080p-encode-engines8", "cfg2-416x240-encode-engines2"):
        assert name in bench.REFERENCE_MD5 and len(bench.REFERENCE_MD5[name]["cumulative_md5"]) == bench.REFERENCE_MD5[name]["frames"]
    for seed in bench.CLIP_SEEDS:
        assert bench.REFERENCE_MD5[bench.seed_workload("cfg2-1080p-encode", seed)]["frames"] >= 3 + 20
    ok, n = bench.check_against_reference("cfg2-1080p-encode", bench.REFERENCE_MD5["cfg2-1080p-encode"]["cumulative_md5"][:5])
    assert ok and n == 5
    ok, _ = bench.check_against_reference("cfg2-1080p-encode", ["0" * 32])
    assert not ok
