#!/usr/bin/env python3
"""GPU box tool: the live roofline of the phase-plane kernels alone (what bench.py reports as roofline.subpel_planes)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "homerhevc_amd", "libhomer_gpu.so"))
lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
lib.hmr_gpu_last_error.restype = C.c_char_p
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
print(json.dumps(bench.subpel_planes_roofline(lib, torch, w, h, reps=50)))
