#!/usr/bin/env python3
"""GPU box tool: the CTU records of frame 0 of a clip over and over (fresh encoder each time); the first run is the yardstick, differing CTUs are listed.
usage: tools/ctu_stress.py width height iterations [key=value ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import encoder_cases as ec  # noqa: E402


def main():
    w, h, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    keys = dict(k.split("=") for k in sys.argv[4:])
    lib = bench.load_lib()
    lib.hmr_gpu_enc_frame_ctus.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int] + [C.c_char_p] * 3 + [C.c_double, C.c_char_p]
    nx, ny = (w + 63) // 64, (h + 63) // 64
    planes = next(iter(ec.clip_frames(w, h, 1)))
    other = next(iter(ec.clip_frames(416, 240, 1)))
    first = None
    bad = 0
    for it in range(iters):
        # something else on the GPU in between, as in the test suite
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0
        cfg = ec.default_cfg(416, 240)
        assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0
        r0 = C.create_string_buffer(ec.REC * 7 * 4)
        assert lib.hmr_gpu_enc_frame_ctus(enc, *other, 0, None, None, None, -1.0, r0) > 0
        lib.hmr_gpu_enc_destroy(enc)
        ctx, enc = C.c_void_p(), C.c_void_p()
        assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0
        cfg = ec.default_cfg(w, h, **keys)
        assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
        recs = C.create_string_buffer(ec.REC * nx * ny)
        assert lib.hmr_gpu_enc_frame_ctus(enc, *planes, 0, None, None, None, -1.0, recs) > 0, lib.hmr_gpu_last_error()
        lib.hmr_gpu_enc_destroy(enc)
        a = np.frombuffer(recs.raw, dtype=np.uint8).reshape(nx * ny, ec.REC)
        if first is None:
            first = a.copy()
            continue
        diff = [n for n in range(nx * ny) if not np.array_equal(a[n], first[n])]
        if diff:
            bad += 1
            n = diff[0]
            offs = np.nonzero(a[n] != first[n])[0]
            print(f"iteration {it}: {len(diff)} CTUs differ, first {n} (row {n // nx}, col {n % nx}); rows touched {sorted(set(d // nx for d in diff))}; in CTU {n}: {len(offs)} bytes from offset {offs[0]} to {offs[-1]}", flush=True)
    print(f"{iters} iterations, {bad} differing")


if __name__ == "__main__":
    main()
