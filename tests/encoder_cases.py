"""Shared pieces of the frame-encoder tests: the configuration struct (HVENC_Cfg layout), the per-CTU record layout of
oracle/ref_ctudump.c, and the comparison of a run against the committed reference fixtures (tests/golden/ctus_*.npz)."""
import ctypes as C
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLDEN = os.path.join(ROOT, "tests", "golden")

FIELDS = [("hdr", 32, np.int32), ("cbf", 768, np.uint8), ("intra_mode", 512, np.uint8), ("inter_mode", 256, np.uint8), ("tr_idx", 256, np.uint8),
          ("pred_depth", 256, np.uint8), ("part_size_type", 256, np.uint8), ("pred_mode", 256, np.uint8), ("skipped", 256, np.uint8), ("merge", 256, np.uint8),
          ("merge_idx", 256, np.uint8), ("qp", 256, np.uint8), ("mv_ref_idx", 256, np.int8), ("mv_diff_ref_idx", 256, np.uint8), ("mv_ref", 2048, np.int32),
          ("mv_diff", 2048, np.int32), ("coeff", 12288, np.int16), ("recon", 12288, np.int16), ("mode_buffs", 2560, np.uint8)]
REC = sum(f[1] for f in FIELDS)
# what a run must reproduce: everything, including the single worker thread's mode buffers after each CTU (the device rebuilds that chain, enc_sched.h)
COMPARED = [f[0] for f in FIELDS]


class EncCfg(C.Structure):
    """HVENC_Cfg, homer_hevc_enc_api.h:138-167"""
    _fields_ = [("size", C.c_int32), ("profile", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("frame_rate", C.c_float), ("cu_size", C.c_int32),
                ("max_pred_partition_depth", C.c_int32), ("max_intra_tr_depth", C.c_int32), ("max_inter_tr_depth", C.c_int32), ("intra_period", C.c_int32),
                ("gop_size", C.c_int32), ("num_b", C.c_int32), ("num_ref_frames", C.c_int32), ("motion_estimation_precision", C.c_int32), ("qp", C.c_int32),
                ("chroma_qp_offset", C.c_int32), ("num_enc_engines", C.c_int32), ("wfpp_enable", C.c_int32), ("wfpp_num_threads", C.c_int32),
                ("sign_hiding", C.c_int32), ("sample_adaptive_offset", C.c_int32), ("bitrate_mode", C.c_int32), ("bitrate", C.c_int32), ("vbv_size", C.c_int32),
                ("vbv_init", C.c_int32), ("reinit_gop_on_scene_change", C.c_int32), ("rd_mode", C.c_int32), ("performance_mode", C.c_int32)]


KEY_NAMES = {"engines": "num_enc_engines", "wpp": "wfpp_num_threads", "perf": "performance_mode", "rd": "rd_mode", "sao": "sample_adaptive_offset", "intra_tr": "max_intra_tr_depth", "inter_tr": "max_inter_tr_depth",
             "me": "motion_estimation_precision", "cqo": "chroma_qp_offset"}


def default_cfg(width, height, **kw):
    """BASELINE.json configs[1], the defaults of oracle/ref_lockstep.c"""
    c = EncCfg(size=C.sizeof(EncCfg), profile=1, width=width, height=height, frame_rate=25.0, cu_size=64, max_pred_partition_depth=4, max_intra_tr_depth=2,
               max_inter_tr_depth=1, intra_period=100, gop_size=1, num_b=0, num_ref_frames=1, motion_estimation_precision=2, qp=32, chroma_qp_offset=2,
               num_enc_engines=1, wfpp_enable=1, wfpp_num_threads=1, sign_hiding=1, sample_adaptive_offset=1, bitrate_mode=0, bitrate=20000, vbv_size=20000,
               vbv_init=7000, reinit_gop_on_scene_change=1, rd_mode=2, performance_mode=2)
    for k, v in kw.items():
        setattr(c, KEY_NAMES.get(k, k), int(v))
    # oracle/ref_lockstep.c: the buffer follows the bit rate
    c.vbv_size = c.bitrate
    c.vbv_init = int(c.bitrate * 0.35)
    return c


def split(rec):
    out, o = {}, 0
    for name, n, dt in FIELDS:
        out[name] = np.frombuffer(rec[o:o + n], dtype=dt)
        o += n
    return out


def _abs2raster():
    t = np.zeros(256, dtype=np.int32)
    for a in range(256):
        x = y = 0
        for b in range(4):
            x |= ((a >> (2 * b)) & 1) << b
            y |= ((a >> (2 * b + 1)) & 1) << b
        t[a] = y * 16 + x
    return t


ABS2RASTER = _abs2raster()


def crop_recon(rec, width, height, nx):
    """zero what lies outside the picture in each CTU's reconstruction and levels: the reference's dump shows stale window content there
    (whatever the worker thread's previous CTU left), which is never coded"""
    rec = bytearray(rec)
    nctu = len(rec) // REC
    off_c = sum(f[1] for f in FIELDS[:16])
    off = sum(f[1] for f in FIELDS[:17])
    ux, uy = (ABS2RASTER % 16) * 4, (ABS2RASTER // 16) * 4
    for n in range(nctu):
        cx, cy = (n % nx) * 64, (n // nx) * 64
        a = np.frombuffer(rec, dtype=np.int16, count=6144, offset=n * REC + off)
        y = a[:4096].reshape(64, 64)
        y[max(0, height - cy):, :] = 0
        y[:, max(0, width - cx):] = 0
        for k in range(2):
            c = a[4096 + 1024 * k:5120 + 1024 * k].reshape(32, 32)
            c[max(0, height // 2 - cy // 2):, :] = 0
            c[:, max(0, width // 2 - cx // 2):] = 0
        outside = (cx + ux >= width) | (cy + uy >= height)
        if outside.any():
            q = np.frombuffer(rec, dtype=np.int16, count=6144, offset=n * REC + off_c)
            q[:4096].reshape(256, 16)[outside] = 0
            q[4096:5120].reshape(256, 4)[outside] = 0
            q[5120:].reshape(256, 4)[outside] = 0
    return bytes(rec)


def field_hashes(rec):
    r = split(rec)
    return np.array([zlib.crc32(r[name].tobytes()) for name in COMPARED], dtype=np.uint32)


def load_fixture(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def clip_frames(width, height, frames, cut_at=None, seed=1234):
    import gen_yuv
    return [tuple(p.tobytes() for p in planes) for planes in gen_yuv.gen_frames(width, height, frames, seed=seed, cut_at=cut_at)]


def check_frame_against_fixture(fx, f, records, width, height):
    """records: bytes of nctu records of frame f from the run under test.  Returns a list of mismatch descriptions."""
    nx = (width + 63) // 64
    nctu = nx * ((height + 63) // 64)
    records = crop_recon(records, width, height, nx)
    bad = []
    if f"f{f}_records" in fx:
        ref = fx[f"f{f}_records"].tobytes()
        for n in range(nctu):
            r, m = split(ref[n * REC:(n + 1) * REC]), split(records[n * REC:(n + 1) * REC])
            for name in COMPARED:
                if not np.array_equal(r[name], m[name]):
                    idx = np.flatnonzero(r[name] != m[name])
                    bad.append(f"frame {f} ctu {n} {name}: {len(idx)} differ, first {idx[:4]} ref {r[name][idx[:4]]} got {m[name][idx[:4]]}")
    else:
        ref = fx[f"f{f}_hashes"]
        for n in range(nctu):
            h = field_hashes(records[n * REC:(n + 1) * REC])
            for k, name in enumerate(COMPARED):
                if h[k] != ref[n, k]:
                    bad.append(f"frame {f} ctu {n} {name}: hash differs")
    return bad
