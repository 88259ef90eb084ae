#!/usr/bin/env python3
"""GPU box tool: aggregate frames/s of S independent sequences encoded concurrently on one GPU (bench.multi_stream) for several S.
usage: [GPU_MAX_HW_QUEUES=16] tools/multi_stream.py 1 2 4 8"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

lib = bench.load_lib()
keys = {"wpp": int(os.environ["WPP"])} if os.environ.get("WPP") else {}
for s in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    print(s, bench.multi_stream(lib, 0, 1920, 1080, keys, s, 2, 3), flush=True)
