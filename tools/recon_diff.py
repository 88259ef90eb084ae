#!/usr/bin/env python3
"""GPU box tool: encode the synthetic clip with the device encoder and with the CPU checker build of the same core (oracle/libenc_cpu.so) and report where the
reconstructed pictures and the streams differ (frame, plane, CTU, sample).

usage: tools/recon_diff.py --width 416 --height 240 --frames 3 [key=value ...]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import encoder_cases as ec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("keys", nargs="*")
    a = ap.parse_args()
    keys = dict(k.split("=") for k in a.keys)
    w, h = a.width, a.height
    cfg = ec.default_cfg(w, h, **keys)
    gpu = C.CDLL(os.path.join(ROOT, "homerhevc_amd", "libhomer_gpu.so"))
    cpu = C.CDLL(os.path.join(ROOT, "oracle", "libenc_cpu.so"))
    gpu.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    gpu.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
    gpu.hmr_gpu_enc_encode.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int, C.c_char_p, C.c_long, C.POINTER(C.c_long), C.c_char_p]
    gpu.hmr_gpu_last_error.restype = C.c_char_p
    cpu.henc_cpu_create.restype = C.c_void_p
    cpu.henc_cpu_create.argtypes = [C.POINTER(ec.EncCfg)]
    cpu.henc_cpu_encode_frame.restype = C.c_long
    cpu.henc_cpu_encode_frame.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int, C.c_char_p, C.c_long, C.c_char_p]
    ctx, enc = C.c_void_p(), C.c_void_p()
    assert gpu.hmr_gpu_create(C.byref(ctx), 0, None) == 0
    assert gpu.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, gpu.hmr_gpu_last_error()
    hc = cpu.henc_cpu_create(C.byref(cfg))
    assert hc
    buf_g, buf_c = C.create_string_buffer(8 << 20), C.create_string_buffer(8 << 20)
    rec_g, rec_c = C.create_string_buffer(w * h * 3 // 2), C.create_string_buffer(w * h * 3 // 2)
    nb = C.c_long()
    for f, planes in enumerate(ec.clip_frames(w, h, a.frames)):
        st = gpu.hmr_gpu_enc_encode(enc, *planes, 0, buf_g, len(buf_g), C.byref(nb), rec_g)
        assert st in (1, 2), gpu.hmr_gpu_last_error()
        n = cpu.henc_cpu_encode_frame(hc, *planes, 0, buf_c, len(buf_c), rec_c)
        same_stream = buf_g.raw[:nb.value] == buf_c.raw[:n]
        g = np.frombuffer(rec_g.raw, np.uint8)
        c = np.frombuffer(rec_c.raw, np.uint8)
        print(f"frame {f}: stream {'identical' if same_stream else f'DIFFERENT ({nb.value} vs {n} bytes)'}; picture {'identical' if (g == c).all() else 'DIFFERENT'}")
        if os.environ.get("RECON_DUMP"):
            gy = g[:w * h].reshape(h, w)
            print("  gpu luma: unique values", np.unique(gy)[:10], "nonzero samples", int((gy != 0).sum()))
            ys, xs = np.nonzero(gy)
            if len(ys): print("  nonzero rows", sorted(set(ys.tolist()))[:40], "cols", sorted(set(xs.tolist()))[:40])
        o = 0
        for comp, (pw, ph) in enumerate([(w, h), (w // 2, h // 2), (w // 2, h // 2)]):
            gp, cp = g[o:o + pw * ph].reshape(ph, pw), c[o:o + pw * ph].reshape(ph, pw)
            o += pw * ph
            ys, xs = np.nonzero(gp != cp)
            if len(ys):
                sz = 64 >> (1 if comp else 0)
                ctus = sorted(set(zip((ys // sz).tolist(), (xs // sz).tolist())))
                print(f"  plane {comp}: {len(ys)} samples differ, CTUs (row, col) {ctus[:12]}, first at y={ys[0]} x={xs[0]}: gpu {gp[ys[0], xs[0]]} cpu {cp[ys[0], xs[0]]}; rows {sorted(set((ys % sz).tolist()))[:20]} cols {sorted(set((xs % sz).tolist()))[:20]}")


if __name__ == "__main__":
    main()
