#!/usr/bin/env python3
"""Build-container tool: diff the CTU encoder core (one-lane checker build, oracle/libenc_cpu.so) against the compiled
reference CTU by CTU.  The reference runs in lockstep with oracle/_ref/ref_ctudump (per-CTU records) and writes its
reconstructed pictures; the core is then driven frame by frame with the reference's previous reconstruction as its
reference picture ("teacher forcing") or, with --free, with its own pictures once the in-loop filters are part of the run.

usage: tools/ctu_diff.py --width 416 --height 240 --frames 3 [key=value ...]   (keys of ref_lockstep)
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_yuv  # noqa: E402

FIELDS = [("hdr", 32, np.int32), ("cbf", 768, np.uint8), ("intra_mode", 512, np.uint8), ("inter_mode", 256, np.uint8), ("tr_idx", 256, np.uint8),
          ("pred_depth", 256, np.uint8), ("part_size_type", 256, np.uint8), ("pred_mode", 256, np.uint8), ("skipped", 256, np.uint8), ("merge", 256, np.uint8),
          ("merge_idx", 256, np.uint8), ("qp", 256, np.uint8), ("mv_ref_idx", 256, np.int8), ("mv_diff_ref_idx", 256, np.uint8), ("mv_ref", 2048, np.int32),
          ("mv_diff", 2048, np.int32), ("coeff", 12288, np.int16), ("recon", 12288, np.int16), ("mode_buffs", 2560, np.uint8)]
REC = sum(f[1] for f in FIELDS)


class HostCfg(C.Structure):
    _fields_ = [("size", C.c_int32), ("profile", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("frame_rate", C.c_float), ("cu_size", C.c_int32),
                ("max_pred_partition_depth", C.c_int32), ("max_intra_tr_depth", C.c_int32), ("max_inter_tr_depth", C.c_int32), ("intra_period", C.c_int32),
                ("gop_size", C.c_int32), ("num_b", C.c_int32), ("num_ref_frames", C.c_int32), ("motion_estimation_precision", C.c_int32), ("qp", C.c_int32),
                ("chroma_qp_offset", C.c_int32), ("num_enc_engines", C.c_int32), ("wfpp_enable", C.c_int32), ("wfpp_num_threads", C.c_int32),
                ("sign_hiding", C.c_int32), ("sample_adaptive_offset", C.c_int32), ("bitrate_mode", C.c_int32), ("bitrate", C.c_int32), ("vbv_size", C.c_int32),
                ("vbv_init", C.c_int32), ("reinit_gop_on_scene_change", C.c_int32), ("rd_mode", C.c_int32), ("performance_mode", C.c_int32)]


def default_cfg(width, height, **kw):
    c = HostCfg(size=C.sizeof(HostCfg), profile=1, width=width, height=height, frame_rate=25.0, cu_size=64, max_pred_partition_depth=4, max_intra_tr_depth=2,
                max_inter_tr_depth=1, intra_period=100, gop_size=1, num_b=0, num_ref_frames=1, motion_estimation_precision=2, qp=32, chroma_qp_offset=2,
                num_enc_engines=1, wfpp_enable=1, wfpp_num_threads=1, sign_hiding=1, sample_adaptive_offset=1, bitrate_mode=0, bitrate=20000, vbv_size=20000,
                vbv_init=7000, reinit_gop_on_scene_change=1, rd_mode=2, performance_mode=2)
    names = {"wpp": "wfpp_num_threads", "engines": "num_enc_engines", "perf": "performance_mode", "rd": "rd_mode", "sao": "sample_adaptive_offset", "intra_tr": "max_intra_tr_depth", "inter_tr": "max_inter_tr_depth", "me": "motion_estimation_precision", "cqo": "chroma_qp_offset"}
    for k, v in kw.items():
        setattr(c, names.get(k, k), int(v))
    return c


def split(rec):
    out, o = {}, 0
    for name, n, dt in FIELDS:
        out[name] = np.frombuffer(rec[o:o + n], dtype=dt)
        o += n
    return out


def load_cpu():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libenc_cpu.so"))
    lib.henc_cpu_create.restype = C.c_void_p
    lib.henc_cpu_create.argtypes = [C.POINTER(HostCfg)]
    lib.henc_cpu_frame_ctus.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int] + [C.c_char_p] * 3 + [C.c_double, C.c_int, C.c_int]
    lib.henc_cpu_records.restype = C.POINTER(C.c_uint8)
    lib.henc_cpu_records.argtypes = [C.c_void_p]
    lib.henc_cpu_avg_dist.restype = C.c_double
    lib.henc_cpu_avg_dist.argtypes = [C.c_void_p]
    lib.henc_cpu_set_trace.argtypes = [C.c_char_p]
    lib.henc_cpu_destroy.argtypes = [C.c_void_p]
    assert lib.henc_cpu_record_bytes() == REC, (lib.henc_cpu_record_bytes(), REC)
    return lib


def run_reference(tmp, width, height, frames, keys, force_intra=False, trace=False, seed=1234):
    yuv = os.path.join(tmp, "in.yuv")
    gen_yuv.write_clip(yuv, width, height, frames, seed)
    env = dict(os.environ, HOMER_CTUDUMP=os.path.join(tmp, "ctus.bin"))
    if int(keys.get("wpp", 1)) > 1 or int(keys.get("engines", 1)) > 1:
        env["HOMER_TURNSTILE"] = "1"      # (the deterministic schedule the streams are pinned on)
    if trace:
        env["HOMER_CUTRACE"] = os.path.join(tmp, "ref_trace.txt")
    cmd = [os.path.join(ROOT, "oracle", "_ref", "ref_ctudump"), yuv, os.path.join(tmp, "out.265"), str(width), str(height), str(frames),
           "recon=" + os.path.join(tmp, "rec.yuv")] + [f"{k}={v}" for k, v in keys.items()]
    if force_intra:
        cmd.append("force_intra=1")
    subprocess.run(cmd, check=True, env=env, stdout=subprocess.DEVNULL)
    return yuv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--force-intra", action="store_true")
    ap.add_argument("--seed", type=int, default=1234, help="clip seed (tools/gen_yuv.py)")
    ap.add_argument("--gpu", action="store_true", help="GPU box: the device encoder (hmr_gpu_enc_frame_ctus) instead of the checker build")
    ap.add_argument("--trace", action="store_true", help="write per-CU traces of both sides next to the dumps")
    ap.add_argument("--keep", default=None, help="directory to keep the artefacts in")
    ap.add_argument("--max-report", type=int, default=6)
    ap.add_argument("keys", nargs="*")
    a = ap.parse_args()
    keys = dict(k.split("=") for k in a.keys)
    tmp = a.keep or tempfile.mkdtemp(prefix="ctudiff_")
    os.makedirs(tmp, exist_ok=True)
    yuv = run_reference(tmp, a.width, a.height, a.frames, keys, a.force_intra, a.trace, a.seed)
    ref = open(os.path.join(tmp, "ctus.bin"), "rb").read()
    rec = open(os.path.join(tmp, "rec.yuv"), "rb").read()
    src = open(yuv, "rb").read()
    fsz = a.width * a.height * 3 // 2
    nctu = ((a.width + 63) // 64) * ((a.height + 63) // 64)
    assert len(ref) == REC * nctu * a.frames, (len(ref), REC, nctu)
    gpu = None
    if a.gpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import libs
        gpu = libs.load_gpu()
        gpu.hmr_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
        gpu.hmr_gpu_enc_create.argtypes = [C.c_void_p, C.POINTER(HostCfg), C.POINTER(C.c_void_p)]
        gpu.hmr_gpu_enc_frame_ctus.argtypes = [C.c_void_p] + [C.c_char_p] * 3 + [C.c_int] + [C.c_char_p] * 3 + [C.c_double, C.c_char_p]
        gpu.hmr_gpu_last_error.restype = C.c_char_p
        gctx, genc = C.c_void_p(), C.c_void_p()
        assert gpu.hmr_gpu_create(C.byref(gctx), 0, None) == 0, gpu.hmr_gpu_last_error()
        gcfg = default_cfg(a.width, a.height, **keys)
        assert gpu.hmr_gpu_enc_create(gctx, C.byref(gcfg), C.byref(genc)) == 0, gpu.hmr_gpu_last_error()
        grecs = C.create_string_buffer(REC * nctu)
    lib = load_cpu()
    cfg = default_cfg(a.width, a.height, **keys)
    h = lib.henc_cpu_create(C.byref(cfg))
    if a.trace:
        lib.henc_cpu_set_trace(os.path.join(tmp, "cpu_trace.txt").encode())
    ysz, csz = a.width * a.height, a.width * a.height // 4
    bad = 0
    for f in range(a.frames):
        fr = src[f * fsz:(f + 1) * fsz]
        planes = [fr[:ysz], fr[ysz:ysz + csz], fr[ysz + csz:]]
        if f > 0:
            pr = rec[(f - 1) * fsz:f * fsz]
            refs = [pr[:ysz], pr[ysz:ysz + csz], pr[ysz + csz:]]
        else:
            refs = [None, None, None]
        st = lib.henc_cpu_frame_ctus(h, *planes, 3 if a.force_intra else 0, *refs, -1.0, 0, -1)
        mine = C.string_at(lib.henc_cpu_records(h), REC * nctu)
        if gpu:         # (the checker runs alongside: its running state - avg_dist - is what the printout below shows)
            st = gpu.hmr_gpu_enc_frame_ctus(genc, *planes, 3 if a.force_intra else 0, *refs, -1.0, grecs)
            assert st > 0, gpu.hmr_gpu_last_error()
            mine = grecs.raw
        nbad = 0
        # (with several WPP threads the reference dumps its records in the order the CTUs finish: pair them by the CTU number in the header)
        by_num = {}
        for k in range(nctu):
            rr = ref[(f * nctu + k) * REC:(f * nctu + k + 1) * REC]
            by_num[int(np.frombuffer(rr[8:12], dtype=np.int32)[0])] = rr
        ref_f = b"".join(by_num.get(n, ref[(f * nctu + n) * REC:(f * nctu + n + 1) * REC]) for n in range(nctu))
        if a.gpu:       # (outside the picture the reference's windows hold what the thread's previous CTU left, the device's hold zeros: compare what is coded)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import encoder_cases as ec
            nx = (a.width + 63) // 64
            ref_f, mine = bytes(ec.crop_recon(ref_f, a.width, a.height, nx)), bytes(ec.crop_recon(mine, a.width, a.height, nx))
        for n in range(nctu):
            r = split(ref_f[n * REC:(n + 1) * REC])
            m = split(mine[n * REC:(n + 1) * REC])
            diffs = [name for name, _, _ in FIELDS if not np.array_equal(r[name], m[name]) and not (name == "mode_buffs" and int(keys.get("wpp", 1)) > 1)]
            if diffs:
                nbad += 1
                if nbad <= a.max_report:
                    print(f"frame {f} (slice {st}) ctu {n}: mismatch in {diffs}")
                    for name in diffs[:4 if not a.gpu else 12]:
                        idx = np.flatnonzero(r[name] != m[name])
                        print(f"   {name}: {len(idx)} entries differ, first at {idx[:6]}: ref {r[name][idx[:6]]} mine {m[name][idx[:6]]}")
        print(f"frame {f}: slice_type {st}, {nctu - nbad}/{nctu} CTUs identical, avg_dist after = {lib.henc_cpu_avg_dist(h):.4f}")
        bad += nbad
    if a.trace:
        lib.henc_cpu_set_trace(b"")
    print("artefacts in", tmp)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
