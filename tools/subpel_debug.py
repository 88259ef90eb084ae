#!/usr/bin/env python3
"""GPU box tool: per-plane mismatch report of hmr_gpu_subpel_planes against the oracle's motion compensation."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402

VP = C.c_void_p
gpu, ora = libs.load_gpu(), libs.load_oracle()
gpu.hmr_gpu_create.argtypes = [C.POINTER(VP), C.c_int, VP]
gpu.hmr_gpu_malloc.argtypes = [VP, C.POINTER(VP), C.c_size_t]
gpu.hmr_gpu_upload.argtypes = [VP, VP, VP, C.c_size_t]
gpu.hmr_gpu_download.argtypes = [VP, VP, VP, C.c_size_t]
gpu.hmr_gpu_subpel_planes.argtypes = [VP] * 4 + [C.c_int] * 4 + [VP] * 3
ora.ora_mc_luma.argtypes = [VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
ora.ora_mc_chroma.argtypes = [VP, C.c_int, VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
ctx = VP()
assert gpu.hmr_gpu_create(C.byref(ctx), 0, None) == 0


def dev(arr):
    p = VP()
    assert gpu.hmr_gpu_malloc(ctx, C.byref(p), C.c_size_t(arr.nbytes)) == 0
    assert gpu.hmr_gpu_upload(ctx, p, VP(arr.ctypes.data), C.c_size_t(arr.nbytes)) == 0
    return p


S, R = 208, 72
Sc, Rc = 108, 39
rng = np.random.default_rng(1)
pic = [rng.integers(0, 256, (R, S)).astype(np.int16), rng.integers(0, 256, (Rc, Sc)).astype(np.int16), rng.integers(0, 256, (Rc, Sc)).astype(np.int16)]
out = [np.zeros((R, 16, S), np.uint8), np.zeros((Rc, 64, Sc), np.uint8), np.zeros((Rc, 64, Sc), np.uint8)]      # row-interleaved planes
d_pic, d_out = [dev(p) for p in pic], [dev(o) for o in out]
assert gpu.hmr_gpu_subpel_planes(ctx, *d_pic, S, R, Sc, Rc, *d_out) == 0
for o, d in zip(out, d_out):
    assert gpu.hmr_gpu_download(ctx, VP(o.ctypes.data), d, C.c_size_t(o.nbytes)) == 0
w, h = S - 8, R - 8
hv, wv = h // 8 * 8, w // 8 * 8
for f in range(16):
    fx, fy = f & 3, f >> 2
    want = np.zeros((h, w), np.int16)
    src = pic[0].ctypes.data + 2 * (4 * S + 4)
    for y0 in range(0, hv, 8):
        for x0 in range(0, wv, 8):
            ora.ora_mc_luma(VP(src + 2 * (y0 * S + x0)), S, VP(want.ctypes.data + 2 * (y0 * w + x0)), w, 8, 8, fx, fy, 0)
    a, b = out[0][4:4 + hv, f, 4:4 + wv], want[:hv, :wv].astype(np.uint8)
    bad = np.argwhere(a != b)
    print("luma", f, len(bad), [(int(y), int(x), int(a[y, x]), int(b[y, x])) for y, x in bad[:6]])
wc, hc = Sc - 8, Rc - 8
hv, wv = hc // 8 * 8, wc // 8 * 8
for comp in (1, 2):
    tot = 0
    for f in range(64):
        fx, fy = f & 7, f >> 3
        want = np.zeros((hc, wc), np.int16)
        src = pic[comp].ctypes.data + 2 * (4 * Sc + 4)
        for y0 in range(0, hv, 8):
            for x0 in range(0, wv, 8):
                ora.ora_mc_chroma(VP(src + 2 * (y0 * Sc + x0)), Sc, VP(want.ctypes.data + 2 * (y0 * wc + x0)), wc, 8, fx, fy, 0)
        a, b = out[comp][4:4 + hv, f, 4:4 + wv], want[:hv, :wv].astype(np.uint8)
        bad = np.argwhere(a != b)
        tot += len(bad)
        if len(bad):
            print("chroma", comp, f, len(bad), [(int(y), int(x), int(a[y, x]), int(b[y, x])) for y, x in bad[:4]])
    print("chroma", comp, "total mismatches", tot)
