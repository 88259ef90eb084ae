#!/usr/bin/env python3
"""GPU box tool: the per-CTU records of the single-sequence kernel against those of the batch kernel (HENC_FORCE_BATCH_KERNEL) on the same frames.
usage: tools/kernel_diff.py [width height frames wpp]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import encoder_cases as ec  # noqa: E402


def run(w, h, frames, wpp, out):
    lib = C.CDLL(os.path.join(ROOT, "homerhevc_amd", "libhomer_gpu.so"))
    lib.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(ec.EncCfg), C.POINTER(C.c_void_p)]
    lib.hmr_gpu_enc_frame_ctus.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int] + [C.c_char_p] * 3 + [C.c_double, C.c_char_p]
    lib.hmr_gpu_last_error.restype = C.c_char_p
    ctx, enc = C.c_void_p(), C.c_void_p()
    assert lib.hmr_gpu_create(C.byref(ctx), 0, None) == 0
    cfg = ec.default_cfg(w, h, wpp=wpp)
    assert lib.hmr_gpu_enc_create(ctx, C.byref(cfg), C.byref(enc)) == 0, lib.hmr_gpu_last_error()
    nctu = ((w + 63) // 64) * ((h + 63) // 64)
    recs = C.create_string_buffer(ec.REC * nctu)
    allrec = []
    for planes in ec.clip_frames(w, h, frames):
        assert lib.hmr_gpu_enc_frame_ctus(enc, *planes, 0, None, None, None, -1.0, recs) > 0, lib.hmr_gpu_last_error()
        allrec.append(recs.raw)
    np.save(out, np.frombuffer(b"".join(allrec), dtype=np.uint8))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
        sys.exit(0)
    w, h, frames, wpp = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (416, 240, 3, 4)
    for tag, env in (("single", {}), ("batch", {"HENC_FORCE_BATCH_KERNEL": "1"})):
        subprocess.run([sys.executable, __file__, "--child", str(w), str(h), str(frames), str(wpp), f"/tmp/kd_{tag}.npy"], check=True, env=dict(os.environ, **env))
    a, b = np.load("/tmp/kd_single.npy").tobytes(), np.load("/tmp/kd_batch.npy").tobytes()
    nctu = ((w + 63) // 64) * ((h + 63) // 64)
    nbad = 0
    for f in range(frames):
        for n in range(nctu):
            o = (f * nctu + n) * ec.REC
            ra, rb = ec.split(a[o:o + ec.REC]), ec.split(b[o:o + ec.REC])
            for name in ec.COMPARED:
                if not np.array_equal(ra[name], rb[name]):
                    idx = np.flatnonzero(ra[name] != rb[name])
                    nbad += 1
                    if nbad <= 12:
                        print(f"frame {f} ctu {n} {name}: {len(idx)} differ, first {idx[:6]} single {ra[name][idx[:6]]} batch {rb[name][idx[:6]]}")
    print("differing (ctu, field) pairs:", nbad)
