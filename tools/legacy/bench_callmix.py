#!/usr/bin/env python3
"""tools/legacy/bench_callmix.py - round 1's call-mix replay, kept as a KERNEL-LEVEL measurement tool (per-launch algorithmic rates of the batched
table kernels).  It is not the bench: a replay has no data dependencies between launches, the real encode does.  bench.py at the repo
root times the real cfg-2 encode through hmr_gpu_enc_encode.

Original description:
encoded-frames/sec of the MI355X hot path on BASELINE.json configs[1].

Workload ("cfg2-1080p-P-frame-replay"): one step = the complete hot-path work of one 1920x1080 P frame of the
reference's cfg-2 encode (IPPP gop_size=1, QP 32, quarter-pel ME, SAO on, wpp=1, engines=1):
  * every call the reference makes through low_level_funcs_t and to the off-table kernels during that frame -
    function, block size, stage flags and calling driver exactly as recorded from the compiled reference by
    oracle/ref_callmix.c (fixture tests/golden/callmix_1080p_cfg2.json, 4.5 M table-level calls per P frame) - issued as
    batched launches over device-resident synthetic planes laid out like the encoder's windows (seeded positions).  Call
    sequences that the reference issues from one per-block driver are replayed by that driver's fused kernel (sub-pel
    refinement, motion compensation, intra mode search, intra / inter / plain TU chains); --unfused replays every table
    call as its own job;
  * the frame-level in-loop passes (edge flags, deblock V+H, SAO statistics, SAO offset, border padding) over the whole
    picture with synthetic side-info.
The launches of a frame are described once as a C command list, captured into a hipGraph (independent launches on
parallel branches) and replayed per step.  Inputs are resident in HBM before the timed region; decisions, CABAC and
bitstream packing stay on the host and are not part of the step.  value = frames/s = steps / wall time
(max over ranks, all GPUs).  Other workloads (--workload): the same encode at 2160p, and the all-intra full-RDO 2160p
configuration of BASELINE configs[4].

Multi-GPU (--gpus N under torch.distributed.run): one encoder engine per GPU (num_enc_engines <-> GPUs, weak
scaling: every rank replays its own frames).  The only data-path exchange is the one the reference's engines
have: the reconstructed, padded reference picture goes from engine r to engine r+1 (mod N) once per frame, as
point-to-point send/recv over RCCL.

Extra objects on the JSON line: `roofline` for the launch with the largest HIP-event time (algorithmic bytes at ABI width per
SURVEY.md 8-d, counter-measured HBM traffic and VALU issue share from the committed profiles) and `cpu_baseline` (the
compiled reference encoder, oracle/_ref/ref_lockstep, timed on this host on a bounded sample of the same configuration;
rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # gpu_abi.py: the ctypes mirror of the batched ABI

W, H = 1920, 1080
HA = 1088                      # CTU-aligned height
PAD = 80                       # reference-frame margin (hmr_encoder_lib.c:1514)
REF_STRIDE = W + 2 * PAD
CREF_STRIDE = W // 2 + PAD
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
WORKLOADS = {   # name -> (width, height, recorded call mix); the metric is quoted on the first (BASELINE.json configs[1])
    "cfg2-1080p-P-frame-replay": (1920, 1080, "callmix_1080p_cfg2.json"),
    "cfg4-2160p-P-frame-replay": (3840, 2160, "callmix_2160p_cfg4.json"),   # configs[3] per engine: the same encode at 2160p
    "cfg5-2160p-all-intra-replay": (3840, 2160, "callmix_2160p_cfg5_intra.json"),   # configs[4]: all-intra, full RDO, intra TU depth 4 (use --callmix-frame 1)
}
REF_ARGS = {"cfg5-2160p-all-intra-replay": ("force_intra=1", "rd=1", "intra_tr=4", "perf=0")}     # lockstep-driver keys of the CPU baseline
CALLMIX = "callmix_1080p_cfg2.json"
WORKLOAD = "cfg2-1080p-P-frame-replay"


def set_workload(name):
    global W, H, HA, REF_STRIDE, CREF_STRIDE, CALLMIX, WORKLOAD
    W, H, CALLMIX = WORKLOADS[name]
    WORKLOAD = name
    HA = (H + 63) // 64 * 64
    REF_STRIDE = W + 2 * PAD
    CREF_STRIDE = W // 2 + PAD


def load_callmix(frame_index):
    with open(os.path.join(ROOT, "tests", "golden", CALLMIX)) as f:
        d = json.load(f)
    return d["frames"][frame_index]["calls"]


class Arena:
    """One int16 device tensor carved into planes/pools; job offsets are element offsets from its base."""

    def __init__(self):
        self.size = 0
        self.init = []      # (offset, numpy int16 array) host-side initial contents

    def alloc(self, n, fill=None):
        off = (self.size + 63) & ~63
        self.size = off + int(n)
        if fill is not None:
            self.init.append((off, fill))
        return off


from gpu_abi import (CHROMA_JOB_DTYPE, INTER_TU_JOB_DTYPE, INTRA_JOB_DTYPE, ITU_JOB_DTYPE, ITU_MODE_FROM_SEARCH, ME_JOB_DTYPE, TREE_JOB_DTYPE, TREE_NO_PARENT,  # noqa: E402
                               TU_JOB_DTYPE)


def build_groups(calls, rng, arena, fused=True, cu_driver=False, cu_rounds=True, chroma_driver=True, inter_source=True):
    """Turn the recorded call mix into batched launches.  Returns list of dict(name, fn, size, jobs, args, bytes)."""
    from gpu_abi import JOB_DTYPE

    # Data layout follows the reference (SURVEY.md §8 header, hmr_encoder_lib.c:1343-1395): per-CTU working windows - source CTU
    # (curr_mbs_wnd), prediction, residual, reconstruction at pitch 64, sub-pel / intermediate windows at pitch 80 - plus the
    # picture-sized padded reference frame.  Every job of CTU k works inside CTU k's windows, so the working set of a batch
    # walks the picture exactly like the encoder does.
    NCX, NCY = W // 64, HA // 64
    NCTU = NCX * NCY
    pix = lambda n: rng.integers(0, 256, n).astype(np.int16)   # noqa: E731
    ref = arena.alloc(REF_STRIDE * (HA + 2 * PAD), pix(REF_STRIDE * (HA + 2 * PAD)))
    ref0 = ref + PAD * REF_STRIDE + PAD
    recf = arena.alloc(REF_STRIDE * (HA + 2 * PAD), pix(REF_STRIDE * (HA + 2 * PAD)))   # frame being reconstructed (intra neighbours)
    recf0 = recf + PAD * REF_STRIDE + PAD
    P64, P80 = 64 * 64, 80 * 80
    # intra TU chains predict from a SMOOTH picture (recs) and code a source that equals it in the upper half of every CTU window and carries
    # strong noise in the lower half, so that the recorded share of TUs comes out coded
    yy, xx = np.mgrid[0:HA + 2 * PAD, 0:REF_STRIDE]
    smooth = np.clip(128 + 60 * np.sin(xx / 53.0) * np.cos(yy / 41.0), 0, 255).astype(np.int16)
    recs = arena.alloc(smooth.size, smooth.ravel())
    recs0 = recs + PAD * REF_STRIDE + PAD
    core = smooth[PAD:PAD + HA, PAD:PAD + W].reshape(HA // 64, 64, W // 64, 64).transpose(0, 2, 1, 3).reshape(-1, 64, 64).astype(np.int64)
    inoise = np.zeros_like(core)
    inoise[:, 32:, :] = rng.integers(-40, 41, (core.shape[0], 32, 64))
    isrc = arena.alloc(core.size, np.clip(core + inoise, 0, 255).astype(np.int16).ravel())
    srcw = arena.alloc(NCTU * P64, pix(NCTU * P64))
    predw = arena.alloc(NCTU * P64, pix(NCTU * P64))
    resw = arena.alloc(NCTU * P64, rng.integers(-255, 256, NCTU * P64).astype(np.int16))
    recw = arena.alloc(NCTU * P64, pix(NCTU * P64))
    tmpw = arena.alloc(NCTU * P80, rng.integers(-8192, 8129, NCTU * P80).astype(np.int16))   # first-stage interpolation output
    outw = arena.alloc(NCTU * P80)
    zero_row = arena.alloc(64, np.zeros(64, np.int16))
    cur_ctu = [None]

    def ctus(n):
        c = (np.arange(n, dtype=np.int64) * NCTU) // max(n, 1)
        cur_ctu[0] = c
        return c

    def wnd(n, base, pitch, bw, bh, align=1, ctu=None):
        """element offsets of an aligned bw x bh block inside the job's CTU window"""
        c = ctus(n) if ctu is None else ctu
        span = pitch                                   # windows are pitch x pitch
        x = rng.integers(0, max(span - bw, 0) // align + 1, n) * align
        y = rng.integers(0, max(span - bh, 0) // align + 1, n) * align
        return (base + c * (pitch * pitch) + y * pitch + x).astype(np.int64)

    def frame(n, base0, bw, bh, lo_x, lo_y, ctu=None):
        """element offsets inside a picture-sized padded frame, within [lo, 64 - lo) of the job's CTU (search window / neighbours)"""
        c = ctus(n) if ctu is None else ctu
        cx, cy = c % NCX, c // NCX
        x = cx * 64 + lo_x + rng.integers(0, max(64 - bw, 0) - 2 * lo_x + 1, n)
        y = cy * 64 + lo_y + rng.integers(0, max(64 - bh, 0) - 2 * lo_y + 1, n)
        x = np.clip(x, -64, W + 64 - bw)
        y = np.clip(y, -64, HA + 64 - bh)
        return (base0 + y * REF_STRIDE + x).astype(np.int64)

    def jobs(n):
        return np.zeros(n, JOB_DTYPE)

    # table calls made inside a chroma CU driver (encode_intra_chroma, keyed ...@chroma by the recorder): replayed as that driver's two launches when fused,
    # as plain table calls otherwise.  Its window copies (synchronize_motion_buffers_chroma) are replayed as copy jobs either way.
    chroma_leaf = {}
    plain = {}
    for key, n in calls.items():
        head, _, rest = key.partition(":")
        kind, _, origin = head.partition("@")
        if origin == "chroma" and (not (fused and chroma_driver) or kind.startswith("copy")):
            key = kind + (":" + rest if rest else "")
        elif origin == "chroma":
            chroma_leaf[key] = n
        plain[key] = plain.get(key, 0) + n
    calls = plain
    merged = {}

    def add(name, fn, size, jb, nbytes, extra=()):
        key = (name, size)
        ctu = cur_ctu[0] if cur_ctu[0] is not None and len(cur_ctu[0]) == len(jb) else np.zeros(len(jb), np.int64)
        if key in merged:
            g = merged[key]
            g["jobs"] = np.concatenate([g["jobs"], jb])
            g["ctu"] = np.concatenate([g["ctu"], ctu])
            g["bytes"] += nbytes
        else:
            merged[key] = {"name": name, "fn": fn, "size": size, "jobs": jb, "ctu": ctu, "bytes": nbytes, "extra": extra}

    for key, n in sorted(calls.items()):
        parts = key.split(":")
        kind, a = parts[0], [int(p) for p in parts[1:]]
        kind, _, origin = kind.partition("@")     # interpolation calls carry their caller: @planes (sub-pel plane builders) / @mc
        if fused and (origin or kind == "sad_direct"):
            continue                               # issued as fused sub-pel refinement / motion compensation jobs below
        if kind in ("mc_luma", "mc_chroma", "half_pel_planes", "quarter_pel_planes", "intra_search", "intra_tu", "inter_tu"):
            continue
        if kind in ("sad", "sad_direct"):
            N = a[0]
            jb = jobs(n)
            c = ctus(n)
            jb["a_off"] = wnd(n, srcw, 64, N, N, align=min(N, 8), ctu=c); jb["a_stride"] = 64
            if kind == "sad" or True:
                jb["b_off"] = frame(n, ref0, N, N, -32, -16, ctu=c); jb["b_stride"] = REF_STRIDE   # candidate block in the search window
            add("sad", "hmr_gpu_sad_batch", N, jb, n * (4 * N * N + 4))
        elif kind == "ssd16b":
            N, z = a
            jb = jobs(n)
            c = ctus(n)
            jb["a_off"] = wnd(n, srcw, 64, N, N, align=min(N, 8), ctu=c); jb["a_stride"] = 64
            if z:
                jb["b_off"] = zero_row; jb["b_stride"] = 0
            else:
                jb["b_off"] = wnd(n, recw, 64, N, N, align=min(N, 8), ctu=c); jb["b_stride"] = 64
            add("ssd16b", "hmr_gpu_ssd16b_batch", N, jb, n * (4 * N * N + 4))
        elif kind == "predict":
            N = a[0]
            jb = jobs(n)
            c = ctus(n)
            jb["a_off"] = wnd(n, srcw, 64, N, N, align=N, ctu=c); jb["a_stride"] = 64
            jb["b_off"] = wnd(n, predw, 64, N, N, align=N, ctu=c); jb["b_stride"] = 64
            jb["c_off"] = wnd(n, resw, 64, N, N, align=N, ctu=c); jb["c_stride"] = 64
            add("predict", "hmr_gpu_predict_batch", N, jb, n * 6 * N * N)
        elif kind == "reconst":
            N, z = a
            jb = jobs(n)
            c = ctus(n)
            jb["a_off"] = wnd(n, predw, 64, N, N, align=N, ctu=c); jb["a_stride"] = 64
            if z:
                jb["b_off"] = zero_row; jb["b_stride"] = 0
            else:
                jb["b_off"] = wnd(n, resw, 64, N, N, align=N, ctu=c); jb["b_stride"] = 64
            jb["c_off"] = wnd(n, recw, 64, N, N, align=N, ctu=c); jb["c_stride"] = 64
            add("reconst", "hmr_gpu_reconst_batch", N, jb, n * 6 * N * N)
        elif kind == "copy_16_16":
            h, w = a
            if w > W:
                continue   # whole-picture copies of the input path (3 per frame) are host-side I/O
            jb = jobs(n)
            c = ctus(n)
            jb["a_off"] = wnd(n, recw, 64, w, h, align=min(w, 8), ctu=c); jb["a_stride"] = 64
            jb["c_off"] = wnd(n, outw, 80, w, h, align=min(w, 8), ctu=c); jb["c_stride"] = 80
            jb["w"] = w; jb["h"] = h
            add("copy_16_16", "hmr_gpu_copy_batch", w if (h == w and w in (4, 8, 16, 32, 64)) else 0, jb, n * 4 * h * w)
        elif kind in ("intra_planar", "intra_angular"):
            N = a[0]
            mode, luma = (0, 1) if kind == "intra_planar" else (a[1], a[2])
            pool = arena.alloc(n * (4 * N + 1), pix(n * (4 * N + 1)))
            jb = jobs(n)
            jb["a_off"] = pool + np.arange(n, dtype=np.int64) * (4 * N + 1)
            jb["c_off"] = wnd(n, predw, 64, N, N, align=N); jb["c_stride"] = 64
            jb["p0"] = mode; jb["p1"] = luma
            add("intra_pred", "hmr_gpu_intra_pred_batch", N, jb, n * (2 * (4 * N + 1) + 2 * N * N))
        elif kind == "fill_reference_samples":
            N, chroma, filt = a
            pool = arena.alloc(2 * n * (4 * N + 1))
            jb = jobs(n)
            jb["a_off"] = frame(n, recf0, 2 * N + 1, 2 * N + 1, -1, -1); jb["a_stride"] = REF_STRIDE   # neighbours in the frame under reconstruction
            jb["c_off"] = pool + np.arange(n, dtype=np.int64) * 2 * (4 * N + 1)
            jb["b_off"] = jb["c_off"] + (4 * N + 1)
            avail = rng.integers(0, 16, n)     # left/top/bl/tr mix; bl implies left, tr implies top
            left, top = (avail & 1) | ((avail >> 2) & 1), ((avail >> 1) & 1) | ((avail >> 3) & 1)
            jb["p0"] = left | (top << 1) | (((avail >> 2) & 1) << 2) | (((avail >> 3) & 1) << 3) | (16 if filt else 0) | 32
            jb["p1"] = N | (N << 16)
            add("intra_refs", "hmr_gpu_intra_refs_batch", N, jb, n * (4 * N + 1) * 2 * (3 if filt else 2))
        elif kind in ("interp_luma", "interp_chroma"):
            w, h, fl, _ = a
            frac_nz, vert, first, last = fl & 1, (fl >> 1) & 1, (fl >> 2) & 1, (fl >> 3) & 1
            luma = kind == "interp_luma"
            taps = 8 if luma else 4
            jb = jobs(n)
            c = ctus(n)
            if first:
                jb["a_off"] = frame(n, ref0, w + 8, h + 8, -24, -12, ctu=c); jb["a_stride"] = REF_STRIDE
            else:
                jb["a_off"] = wnd(n, tmpw, 80, min(w + 8, 80), min(h + 8, 76), ctu=c) + 4 * 80 + 4; jb["a_stride"] = 80
            jb["c_off"] = wnd(n, outw, 80, w, h, ctu=c); jb["c_stride"] = 80
            jb["w"] = w; jb["h"] = h
            jb["p0"] = rng.integers(1, 4 if luma else 8, n) if frac_nz else 0
            jb["p1"] = vert | (first << 1) | (last << 2)
            rd = (w * (h + taps - 1) if vert else (w + taps - 1) * h) if frac_nz else w * h
            items = ((w + 3) // 4) * ((h + 3) // 4) if (vert and frac_nz) else (((w + 7) // 8) * h if frac_nz else ((w + 3) // 4) * h)
            lanes = next(g for g in (4, 8, 16, 32, 64) if items <= g or g == 64)
            add(kind, "hmr_gpu_interpolate_batch", (1 if luma else 0) | (lanes << 8), jb, n * 2 * (rd + w * h))
        elif kind in ("transform", "itransform"):
            N, is_dst = a
            pool = arena.alloc(n * N * N, rng.integers(-200, 201, n * N * N).astype(np.int16) if kind == "itransform" else None)
            jb = jobs(n)
            lin = pool + np.arange(n, dtype=np.int64) * N * N
            blk = wnd(n, resw, 64, N, N, align=N)
            if kind == "transform":
                jb["a_off"] = blk; jb["a_stride"] = 64; jb["c_off"] = lin
            else:
                jb["a_off"] = lin; jb["c_off"] = blk; jb["c_stride"] = 64
            jb["p0"] = is_dst
            add(kind, "hmr_gpu_%s_batch" % kind, N, jb, n * 4 * N * N)
        elif kind in ("quant", "inv_quant"):
            N, comp, intra = a
            # transform-coefficient statistics: energy falls off with frequency and most blocks are weak (two thirds of the
            # reference's quant calls produce an all-zero block: inv_quant / quant call ratio of the recorded mix)
            fall = (1.0 / (1.0 + 0.6 * np.add.outer(np.arange(N), np.arange(N))) ** 1.5).ravel()
            scale = 1500.0 * rng.random(n) ** 3
            init = (rng.standard_normal((n, N * N)) * scale[:, None] * fall[None, :]).astype(np.int16).ravel()
            pin = arena.alloc(n * N * N, init if kind == "quant" else (init // 64).astype(np.int16))
            pout = arena.alloc(n * N * N)
            ctus(n)
            jb = jobs(n)
            jb["a_off"] = pin + np.arange(n, dtype=np.int64) * N * N
            jb["c_off"] = pout + np.arange(n, dtype=np.int64) * N * N
            jb["p0"] = 3 | (comp << 2) | (intra << 4) | (0 << 5) | (1 << 6)   # diagonal scan, P slice, sign hiding on
            jb["p1"] = 5 | (2 << 8)                                             # QP 32: per 5, rem 2
            add(kind, "hmr_gpu_%s_batch" % kind, N, jb, n * 4 * N * N)
        # half_pel_planes / quarter_pel_planes are drivers whose interpolation calls are already counted;
        # deblock_ctu / sao_* / pad_ctu are issued as the frame-level passes below.
    if fused:
        # Sub-pel refinement (hmr_half/quarter_pixel_estimation_luma_hm + the direct sad calls of hmr_motion_estimation) is ONE job per
        # PU of the motion-estimation kernel (action = half | quarter): the 16 sub-pel planes never leave the chip.  Algorithmic bytes
        # = those of the interpolation and sad calls it stands for.
        def interp_bytes(origin, pred):
            tot_b = 0
            for key, n in calls.items():
                kp = key.split(":")
                if "@" not in kp[0] or kp[0].split("@")[1] != origin:
                    continue
                luma = kp[0].startswith("interp_luma")
                w, h, fl = int(kp[1]), int(kp[2]), int(kp[3])
                if not pred(luma, w, h):
                    continue
                taps = 8 if luma else 4
                rd = (w * (h + taps - 1) if (fl >> 1) & 1 else (w + taps - 1) * h) if fl & 1 else w * h
                tot_b += n * 2 * (rd + w * h)
            return tot_b

        for key, n in sorted(calls.items()):
            kp = key.split(":")
            if kp[0] != "half_pel_planes":
                continue
            N = int(kp[1])
            jb = np.zeros(n, ME_JOB_DTYPE)
            c = ctus(n)
            jb["corr"] = 32 * 0.5
            jb["orig_off"] = wnd(n, srcw, 64, N, N, align=min(N, 8), ctu=c); jb["orig_stride"] = 64
            blk = frame(n, ref0, N, N, 0, 0, ctu=c)
            jb["ref_off"] = blk; jb["ref_stride"] = REF_STRIDE
            rel = blk - ref0
            jb["gx"] = rel % REF_STRIDE; jb["gy"] = rel // REF_STRIDE
            jb["init_x"] = rng.integers(-24, 25, n); jb["init_y"] = rng.integers(-12, 13, n)
            jb["n_amvp"] = 1
            jb["amvp"][:, 0, 0] = rng.integers(-64, 65, n); jb["amvp"][:, 0, 1] = rng.integers(-32, 33, n)
            jb["action"] = 6
            nb = interp_bytes("planes", lambda luma, w, h, N=N: luma and w in (N, N + 1) and h in (N, N + 1, N + 7, N + 8))
            nb += sum(v for k, v in calls.items() if k == "sad_direct:%d" % N) * (4 * N * N + 4)
            merged[("me_subpel", N)] = {"name": "me_subpel", "fn": "hmr_gpu_motion_estimation_batch", "size": N, "jobs": jb, "ctu": c, "bytes": nb, "extra": ()}
        # Intra mode search (homer_loop1_motion_intra): one job per PU instead of one reference build + up to 13 {prediction, SAD} pairs
        # calls made inside a luma CU driver (intra_cu:N = encode_intra_luma with a one-level tree) are issued as that driver's chain further down:
        # per CU one search at N, one parent TU at N (none for N = 64) and four child TUs at N / 2
        cu_n = {int(k2.split(":")[1]): v for k2, v in calls.items() if k2.split(":")[0] == "intra_cu"} if cu_driver else {}
        search_total = {int(k2.split(":")[1]): v for k2, v in calls.items() if k2.split(":")[0] == "intra_search"}
        itu_total = {int(k2.split(":")[1]): v for k2, v in calls.items() if k2.split(":")[0] == "intra_tu"}
        cu_bytes = {}
        for key, n_all in sorted(calls.items()):
            kp = key.split(":")
            if kp[0] != "intra_search":
                continue
            N = int(kp[1])
            n = n_all - cu_n.get(N, 0)
            jb = np.zeros(n, INTRA_JOB_DTYPE)
            c = ctus(n)
            jb["sqrt_lambda"] = 7.5
            jb["orig_off"] = wnd(n, srcw, 64, N, N, align=N, ctu=c); jb["orig_stride"] = 64
            jb["dec_off"] = frame(n, recf0, 2 * N + 1, 2 * N + 1, -1, -1, ctu=c); jb["dec_stride"] = REF_STRIDE
            pool = arena.alloc(n * 2 * (4 * N + 4))
            jb["adi_off"] = pool + np.arange(n, dtype=np.int64) * 2 * (4 * N + 4); jb["adif_off"] = jb["adi_off"] + 4 * N + 4
            jb["pred_off"] = wnd(n, predw, 64, N, N, align=N, ctu=c); jb["pred_stride"] = 64
            avail = np.where(rng.random(n) < 0.8, 15, rng.integers(0, 16, n))
            left, top = (avail & 1) | ((avail >> 2) & 1), ((avail >> 1) & 1) | ((avail >> 3) & 1)
            jb["flags"] = left | (top << 1) | (((avail >> 2) & 1) << 2) | (((avail >> 3) & 1) << 3) | 32
            jb["sizes"] = N | (N << 16)
            jb["preds"] = np.stack([rng.integers(2, 35, n), np.zeros(n, np.int64), np.ones(n, np.int64)], 1)
            jb["pred_bits"] = 1; jb["other_bits"] = 12       # RD_FAST
            nb = 0
            for k2, v in calls.items():
                q = k2.split(":")
                if "@search" in q[0] and int(q[1]) == N:
                    nb += v * {"fill_reference_samples@search": (4 * N + 1) * 2 * 3, "sad@search": 4 * N * N + 4}.get(q[0], 2 * (4 * N + 1) + 2 * N * N)
            cu_bytes[("search", N)] = nb / n_all
            if n:
                merged[("intra_search", N)] = {"name": "intra_search", "fn": "hmr_gpu_intra_search_batch", "size": N, "jobs": jb, "ctu": c, "bytes": nb * n // n_all,
                                               "extra": ()}
        # Intra TU chain (encode_intra_cu): neighbour array + prediction + the seven-call TU chain as one job per luma intra TU
        itu_coded = {}
        for key, n_all in sorted(calls.items()):
            kp = key.split(":")
            if kp[0] != "intra_tu":
                continue
            N = int(kp[1])
            n = n_all - (cu_n.get(N, 0) if N <= 32 else 0) - 4 * cu_n.get(2 * N, 0)
            assert n >= 0, (key, n)
            inner = {k2.split(":")[0]: 0 for k2 in calls if "@itu" in k2}
            nb = 0
            for k2, v in calls.items():
                q = k2.split(":")
                if "@itu" in q[0] and int(q[1]) == N:
                    inner[q[0]] += v
                    nb += v * {"fill_reference_samples@itu": (4 * N + 1) * 2 * (3 if int(q[3]) else 2) if q[0].startswith("fill") else 0,
                               "intra_planar@itu": 2 * (4 * N + 1) + 2 * N * N, "intra_angular@itu": 2 * (4 * N + 1) + 2 * N * N,
                               "predict@itu": 6 * N * N, "reconst@itu": 6 * N * N, "ssd16b@itu": 4 * N * N + 4}.get(q[0], 4 * N * N)
            coded_frac = inner.get("inv_quant@itu", 0) / max(inner.get("quant@itu", 1), 1)
            itu_coded[N] = coded_frac
            cu_bytes[("tu", N)] = nb / n_all
            if not n:
                continue
            nb = nb * n // n_all
            jb = np.zeros(n, ITU_JOB_DTYPE)
            c = ctus(n)
            coded = rng.random(n) < coded_frac
            # source = the frame under reconstruction's texture at the TU (+ strong noise in the lower half of the CTU window for coded TUs)
            half = max(32 - N, 0) // N + 1
            x = rng.integers(0, (64 - N) // N + 1, n) * N
            y = np.minimum(rng.integers(0, half, n) * N + np.where(coded, 32, 0), 64 - N)
            pos_in = c * P64 + y * 64 + x
            jb["orig_off"] = isrc + pos_in; jb["orig_stride"] = 64
            jb["pred_off"] = predw + pos_in; jb["pred_stride"] = 64
            jb["rec_off"] = recw + pos_in; jb["rec_stride"] = 64
            cx, cy = c % NCX, c // NCX
            jb["dec_off"] = recs0 + (cy * 64 + y - 1) * REF_STRIDE + cx * 64 + x - 1; jb["dec_stride"] = REF_STRIDE
            lev_pool = arena.alloc(n * N * N)
            jb["lev_off"] = lev_pool + np.arange(n, dtype=np.int64) * N * N
            mode = rng.integers(0, 35, n)
            thr = {4: 10, 8: 7, 16: 1, 32: 0}[N]
            filt = ((mode != 1) & (np.minimum(np.abs(mode - 10), np.abs(mode - 26)) > thr)).astype(np.int64)
            jb["flags"] = 15 | 32 | (filt << 6) | (1 << 7); jb["sizes"] = N | (N << 16); jb["mode"] = mode
            jb["p0"] = 3 | (0 << 2) | (1 << 4) | (0 << 5) | (1 << 6) | ((1 if N == 4 else 0) << 7)     # diagonal scan, luma, P slice, sign hiding on, DST for 4x4
            jb["p1"] = 5 | (2 << 8)
            merged[("intra_tu", N)] = {"name": "intra_tu", "fn": "hmr_gpu_intra_tu_chain_batch", "size": N, "jobs": jb, "ctu": c, "bytes": nb, "extra": ()}
        # Chroma CU driver (encode_intra_chroma): per chroma CU size one search launch (five candidates on U and V; a 64x64 CU is searched on its first
        # quadrant) and the TUs of the winner - unsplit CUs one launch of TUs of that size, split CUs one launch of four rounds at half the size - with the
        # mode handed over on the device.  Every CU owns a pair of (2S+1)^2 planes cut from the smooth picture and a source pair equal to it plus noise.
        leaf_bytes = {"fill_reference_samples": lambda q: (4 * q[0] + 1) * 2 * 2, "intra_planar": lambda q: 2 * (4 * q[0] + 1) + 2 * q[0] * q[0],
                      "intra_angular": lambda q: 2 * (4 * q[0] + 1) + 2 * q[0] * q[0], "sad": lambda q: 4 * q[0] * q[0] + 4, "predict": lambda q: 6 * q[0] * q[0],
                      "reconst": lambda q: 6 * q[0] * q[0], "ssd16b": lambda q: 4 * q[0] * q[0] + 4}
        cb = {}                                       # (phase, block size) -> algorithmic bytes of the table calls
        cnt = {}
        for key, v in chroma_leaf.items():
            head, _, rest = key.partition(":")
            kind = head.partition("@")[0]
            q = [int(t) for t in rest.split(":")] if rest else [0]
            cnt[(kind, q[0])] = cnt.get((kind, q[0]), 0) + v
        for (kind, n_), v in cnt.items():
            per_call = leaf_bytes.get(kind, lambda q: 4 * q[0] * q[0])([n_])
            if kind in ("fill_reference_samples", "intra_planar", "intra_angular"):      # one per candidate and component in the search, one per TU and component after
                tu_share = cnt.get(("predict", n_), 0) / max(cnt.get(("predict", n_), 0) + cnt.get(("sad", n_), 0), 1)
                cb[("tu", n_)] = cb.get(("tu", n_), 0) + v * per_call * tu_share
                cb[("search", n_)] = cb.get(("search", n_), 0) + v * per_call * (1 - tu_share)
            else:
                ph = "search" if kind == "sad" else "tu"
                cb[(ph, n_)] = cb.get((ph, n_), 0) + v * per_call
        ccu = {}
        for key, v in calls.items():
            kp = key.split(":")
            if kp[0] == "intra_chroma_cu" and chroma_driver:
                ccu[(int(kp[1]), int(kp[2]))] = v
        coded_c = cnt.get(("inv_quant", 4), 0) + cnt.get(("inv_quant", 8), 0) + cnt.get(("inv_quant", 16), 0)
        coded_c = coded_c / max(cnt.get(("quant", 4), 0) + cnt.get(("quant", 8), 0) + cnt.get(("quant", 16), 0), 1)
        n_search = {}
        n_tu = {}
        for (S, sp), m in ccu.items():
            n_search[min(S, 16)] = n_search.get(min(S, 16), 0) + m
            tn = S // 2 if sp else S
            n_tu[tn] = n_tu.get(tn, 0) + m * (8 if sp else 2)
        for ss in sorted({min(S, 16) for S, _ in ccu}):
            # one search launch per kernel size (the CUs of both TU shapes, and the 64x64 CUs searched at 16), then one TU launch per (CU size, shape)
            entries = [(S, sp, m) for (S, sp), m in sorted(ccu.items()) if min(S, 16) == ss]
            chain = "chroma%d" % ss
            sjs, off, ssd_off = [], 0, 0
            for S, sp, m in entries:
                ring = 2 * S + 1
                c = ctus(m)
                yy, xx = np.mgrid[0:ring, 0:ring]
                pools, srcs = [], []
                for comp in range(2):
                    ty = rng.integers(0, HA + 2 * PAD - ring, m); tx = rng.integers(0, REF_STRIDE - ring, m)
                    tiles = smooth[ty[:, None, None] + yy, tx[:, None, None] + xx]
                    pools.append(arena.alloc(m * ring * ring, tiles.ravel()) + np.arange(m, dtype=np.int64) * ring * ring)
                    coded = rng.random(m) < coded_c
                    src = tiles[:, 1:S + 1, 1:S + 1].astype(np.int64) + np.where(coded[:, None, None], rng.integers(-40, 41, (m, S, S)), 0)
                    srcs.append(arena.alloc(m * S * S, np.clip(src, 0, 255).astype(np.int16).ravel()) + np.arange(m, dtype=np.int64) * S * S)
                o_pred = [arena.alloc(m * S * S) + np.arange(m, dtype=np.int64) * S * S for _ in range(2)]
                o_lev = [arena.alloc(m * S * S) + np.arange(m, dtype=np.int64) * S * S for _ in range(2)]
                sj = np.zeros(m, CHROMA_JOB_DTYPE)
                sj["sqrt_lambda"] = 7.5
                sj["orig_u_off"] = srcs[0]; sj["orig_v_off"] = srcs[1]; sj["orig_stride"] = S
                sj["dec_u_off"] = pools[0]; sj["dec_v_off"] = pools[1]; sj["dec_stride"] = ring
                sj["flags"] = 15; sj["sizes"] = ss | (ss << 16)
                sj["luma_mode"] = rng.integers(0, 35, m)
                sjs.append(sj)
                tn, rounds = (S // 2, 4) if sp else (S, 1)
                nbf = [15, 3 | 8, 3 | 8 | 4, 3] if sp else [15]
                t = np.zeros((rounds, 2 * m), ITU_JOB_DTYPE)
                for r_ in range(rounds):
                    x0, y0 = ((r_ & 1) * tn, (r_ >> 1) * tn) if sp else (0, 0)
                    for comp in range(2):
                        q = t[r_, comp::2]
                        q["orig_off"] = srcs[comp] + y0 * S + x0; q["orig_stride"] = S
                        q["pred_off"] = o_pred[comp] + y0 * S + x0; q["pred_stride"] = S
                        q["dec_off"] = pools[comp] + y0 * ring + x0; q["dec_stride"] = ring
                        q["rec_off"] = q["dec_off"] + ring + 1; q["rec_stride"] = ring
                        q["lev_off"] = o_lev[comp] + r_ * tn * tn
                        q["flags"] = nbf[r_] | ITU_MODE_FROM_SEARCH; q["sizes"] = tn | (tn << 16)
                        q["mode"] = off + np.arange(m)
                        q["p0"] = ((comp + 1) << 2) | (1 << 4) | (1 << 6); q["p1"] = 5 | (2 << 8)
                merged[("chroma_tus%ds%d" % (S, sp), S)] = {"name": "chroma_tus%ds%d" % (S, sp), "fn": "hmr_gpu_intra_tu_chain_modes_batch", "size": tn, "jobs": t.reshape(-1),
                                                            "ctu": np.arange(rounds * 2 * m), "bytes": int(cb.get(("tu", tn), 0) * m * (8 if sp else 2) / max(n_tu[tn], 1)),
                                                            "extra": (), "chain": chain, "level": 0, "njobs": 2 * m, "rounds": rounds, "ssd_off": ssd_off}
                off += m
                ssd_off += rounds * 2 * m
            sj = np.concatenate(sjs)
            # the search goes first in the chain: re-insert the TU groups after it
            tus = {k_: merged.pop(k_) for k_ in [k2 for k2, g2 in merged.items() if g2.get("chain") == chain]}
            merged[("chroma_search", ss)] = {"name": "chroma_search", "fn": "hmr_gpu_chroma_search_batch", "size": ss, "jobs": sj, "ctu": np.arange(len(sj)),
                                             "bytes": int(cb.get(("search", ss), 0)), "extra": (), "chain": chain, "level": -1}
            merged.update(tus)
        # Luma intra CU driver (encode_intra_luma, one-level tree): search -> parent TUs -> children 0..3 -> consolidation as seven ordered launches per
        # CU size, the mode handed from the search to the TU launches on the device.  Every CU owns a pair of (2N+1)^2 planes (parent / child level) cut
        # from the smooth picture - its neighbourhood - and a source block equal to it, plus noise where the recorded share of TUs is coded.
        for N, m in sorted(cu_n.items()):
            h, ring = N // 2, 2 * N + 1
            c = ctus(m)
            ty = rng.integers(0, HA + 2 * PAD - ring, m); tx = rng.integers(0, REF_STRIDE - ring, m)
            yy, xx = np.mgrid[0:ring, 0:ring]
            tiles = smooth[ty[:, None, None] + yy, tx[:, None, None] + xx]                       # m x ring x ring
            planes_pool = arena.alloc(2 * m * ring * ring, np.repeat(tiles.reshape(m, 1, -1), 2, 1).ravel())
            coded = rng.random(m) < itu_coded.get(N if N <= 32 else h, 0.5)
            src = tiles[:, 1:N + 1, 1:N + 1].astype(np.int64) + np.where(coded[:, None, None], rng.integers(-40, 41, (m, N, N)), 0)
            src_pool = arena.alloc(m * N * N, np.clip(src, 0, 255).astype(np.int16).ravel())
            o_pp = planes_pool + np.arange(m, dtype=np.int64) * 2 * ring * ring; o_pc = o_pp + ring * ring
            o_src = src_pool + np.arange(m, dtype=np.int64) * N * N
            o_pred = arena.alloc(m * N * N) + np.arange(m, dtype=np.int64) * N * N
            lev_pool = arena.alloc(2 * m * N * N)
            o_lp = lev_pool + np.arange(m, dtype=np.int64) * 2 * N * N; o_lc = o_lp + N * N
            adi_pool = arena.alloc(m * 2 * (4 * N + 4))
            sj = np.zeros(m, INTRA_JOB_DTYPE)
            sj["sqrt_lambda"] = 7.5
            sj["orig_off"] = o_src; sj["orig_stride"] = N; sj["dec_off"] = o_pp; sj["dec_stride"] = ring
            sj["adi_off"] = adi_pool + np.arange(m, dtype=np.int64) * 2 * (4 * N + 4); sj["adif_off"] = sj["adi_off"] + 4 * N + 4
            sj["pred_off"] = o_pred; sj["pred_stride"] = N
            sj["flags"] = 15 | 32; sj["sizes"] = N | (N << 16)
            sj["preds"] = np.stack([rng.integers(2, 35, m), np.zeros(m, np.int64), np.ones(m, np.int64)], 1)
            sj["pred_bits"] = 1; sj["other_bits"] = 12       # RD_FAST
            chain = "cu%d" % N
            merged[("cu_search", N)] = {"name": "cu_search", "fn": "hmr_gpu_intra_search_batch", "size": N, "jobs": sj, "ctu": c, "bytes": int(cu_bytes[("search", N)] * m),
                                        "extra": (), "chain": chain, "level": -1}
            gx, gy, gs = [0, 0, h, 0, h], [0, 0, 0, h, h], [N, h, h, h, h]
            nbf = [15, 15, 3 | 8, 3 | 8 | 4, 3]      # neighbour flags of the CU and of its quadrants in an interior position
            children = []
            for k in range(0 if N <= 32 else 1, 5):
                t = np.zeros(m, ITU_JOB_DTYPE)
                plane = o_pc if k else o_pp
                t["orig_off"] = o_src + gy[k] * N + gx[k]; t["orig_stride"] = N
                t["pred_off"] = o_pred + gy[k] * N + gx[k]; t["pred_stride"] = N
                t["dec_off"] = plane + gy[k] * ring + gx[k]; t["dec_stride"] = ring
                t["rec_off"] = t["dec_off"] + ring + 1; t["rec_stride"] = ring
                t["lev_off"] = (o_lc + (k - 1) * h * h) if k else o_lp
                t["flags"] = nbf[k] | 32 | (1 << 7) | ITU_MODE_FROM_SEARCH; t["sizes"] = gs[k] | (gs[k] << 16)
                t["mode"] = np.arange(m)
                t["p0"] = (1 << 4) | (1 << 6) | ((1 if gs[k] == 4 else 0) << 7); t["p1"] = 5 | (2 << 8)
                if k and cu_rounds:
                    children.append(t)
                    continue
                merged[("cu_tu%d" % k, gs[k])] = {"name": "cu_tu%d" % k, "fn": "hmr_gpu_intra_tu_chain_modes_batch", "size": gs[k], "jobs": t, "ctu": c,
                                                  "bytes": int(cu_bytes[("tu", gs[k])] * m), "extra": (), "chain": chain, "level": k}
            if children:      # the four children of every CU back to back in one launch (four rounds over the same lanes)
                merged[("cu_children", h)] = {"name": "cu_children", "fn": "hmr_gpu_intra_tu_chain_modes_batch", "size": h, "jobs": np.concatenate(children),
                                              "ctu": np.arange(4 * m), "bytes": int(cu_bytes[("tu", h)] * 4 * m), "extra": (), "chain": chain, "level": 1,
                                              "njobs": m, "rounds": 4}
            dj = np.zeros(m, TREE_JOB_DTYPE)
            dj["parent"] = np.arange(m) if N <= 32 else TREE_NO_PARENT
            for k in range(4):
                dj["child"][:, k] = (k + 1) * m + np.arange(m)
            dj["par_rec_off"] = o_pp + ring + 1; dj["par_rec_stride"] = ring; dj["chl_rec_off"] = o_pc + ring + 1; dj["chl_rec_stride"] = ring
            dj["par_lev_off"] = o_lp; dj["chl_lev_off"] = o_lc; dj["size"] = N; dj["rule"] = 1
            # the consolidation's copies are the copy_16_16 calls of synchronize_motion_buffers_luma / wnd_copy, which the mix already replays as copy
            # jobs: priced there, not twice
            merged[("cu_decide", N)] = {"name": "cu_decide", "fn": "hmr_gpu_tree_decide_batch", "size": N, "jobs": dj, "ctu": c, "bytes": 0, "extra": (),
                                        "chain": chain, "level": 5}
        # Inter TU chain (encode_inter_cu / _chroma): DCT + quant + keep-or-drop decision + reconstruction as one job per inter TU.  The recorded
        # mix gives the shares: coded = inv_quant / quant, kept = reconst with a residual / coded.
        eres = np.zeros((NCTU, 64, 64), np.int64)
        eres[:, 32:, :] = rng.integers(-40, 41, (NCTU, 32, 64))
        eresw = arena.alloc(NCTU * P64, eres.astype(np.int16).ravel())
        # inter_source: the TU jobs address the source block and form the residual themselves (what the CU-level `predict` calls ahead of encode_inter write
        # out in the reference): prediction window in 40..215, source = prediction + the residual above, so the same residuals reach the transform
        epred = rng.integers(40, 216, (NCTU, 64, 64))
        epredw = arena.alloc(NCTU * P64, epred.astype(np.int16).ravel()) if inter_source else predw
        esrcw = arena.alloc(NCTU * P64, (epred + eres).astype(np.int16).ravel()) if inter_source else eresw
        for N in (4, 8, 16, 32):
            keys = {k2: v for k2, v in calls.items() if k2.split(":")[0] == "inter_tu" and int(k2.split(":")[1]) == N}
            if not keys:
                continue
            inner, nb = {}, 0
            for k2, v in calls.items():
                q = k2.split(":")
                if "@etu" in q[0] and int(q[1]) == N:
                    inner[q[0]] = inner.get(q[0], 0) + v
                    nb += v * {"reconst@etu": 6 * N * N, "ssd16b@etu": 4 * N * N + 4}.get(q[0], 4 * N * N)
            n_coded = inner.get("inv_quant@etu", 0)
            n_kept = calls.get("reconst@etu:%d:0" % N, 0)
            parts = []
            for k2, n in sorted(keys.items()):
                comp = int(k2.split(":")[2])
                jb = np.zeros(n, INTER_TU_JOB_DTYPE)
                c = ctus(n)
                coded = rng.random(n) < n_coded / max(inner.get("quant@etu", 1), 1)
                kept = rng.random(n) < n_kept / max(n_coded, 1)
                half = max(32 - N, 0) // N + 1
                x = rng.integers(0, (64 - N) // N + 1, n) * N
                y = np.minimum(rng.integers(0, half, n) * N + np.where(coded, 32, 0), 64 - N)
                pos_in = c * P64 + y * 64 + x
                jb["orig_off"] = esrcw + pos_in; jb["orig_stride"] = 64
                jb["pred_off"] = epredw + pos_in; jb["pred_stride"] = 64
                jb["reserved"] = 1 if inter_source else 0
                jb["rec_off"] = recw + pos_in; jb["rec_stride"] = 64
                jb["p0"] = 3 | (comp << 2) | (0 << 4) | (0 << 5) | (1 << 6)
                jb["p1"] = 5 | (2 << 8)
                jb["weight"] = 1.0 if comp == 0 else 2.0 ** (2 / 3.0)
                jb["zero_thr"] = np.where(kept, 1.0, 20000.0)
                parts.append((jb, c))
            jobs_all = np.concatenate([q[0] for q in parts])
            lev_pool = arena.alloc(len(jobs_all) * N * N)
            jobs_all["lev_off"] = lev_pool + np.arange(len(jobs_all), dtype=np.int64) * N * N
            merged[("inter_tu", N)] = {"name": "inter_tu", "fn": "hmr_gpu_inter_tu_chain_batch", "size": N, "jobs": jobs_all,
                                       "ctu": np.concatenate([q[1] for q in parts]), "bytes": nb, "extra": ()}
        # Motion compensation (hmr_motion_compensation_luma / _chroma): one job per PU and component instead of one or two
        # interpolation calls through the 80-pitch intermediate window.
        refc = arena.alloc(2 * CREF_STRIDE * (HA // 2 + PAD), pix(2 * CREF_STRIDE * (HA // 2 + PAD)))
        for key, n in sorted(calls.items()):
            kp = key.split(":")
            if kp[0] not in ("mc_luma", "mc_chroma"):
                continue
            luma = kp[0] == "mc_luma"
            if luma:
                w, h, fx, fy = (int(v) for v in kp[1:5])
            else:
                w = h = int(kp[1]); fx, fy = int(kp[2]), int(kp[3])
            taps, fbits = (8, 2) if luma else (4, 3)
            jb = jobs(n)
            c = ctus(n)
            if luma:
                jb["a_off"] = frame(n, ref0, w, h, 0, 0, ctu=c); jb["a_stride"] = REF_STRIDE
            else:
                cx, cy = c % NCX, c // NCX
                plane = rng.integers(0, 2, n)
                x = cx * 32 + rng.integers(0, 32 - w + 1, n); y = cy * 32 + rng.integers(0, 32 - h + 1, n)
                jb["a_off"] = refc + plane * (CREF_STRIDE * (HA // 2 + PAD)) + (y + PAD // 2) * CREF_STRIDE + PAD // 2 + x; jb["a_stride"] = CREF_STRIDE
            jb["c_off"] = wnd(n, predw, 64, w, h, align=min(w, 8), ctu=c); jb["c_stride"] = 64
            jb["w"] = w; jb["h"] = h
            mvx = (rng.integers(-10, 11, n) << fbits) + (rng.integers(1, 1 << fbits, n) if fx else 0)
            mvy = (rng.integers(-6, 7, n) << fbits) + (rng.integers(1, 1 << fbits, n) if fy else 0)
            jb["p0"] = mvx.astype(np.int32).view(np.uint32); jb["p1"] = mvy.astype(np.int32).view(np.uint32)
            # lanes per block: about two first-stage work items (four outputs each) per lane, and the first stage must fit the block's
            # share of the wave's LDS tile
            items = (w // 4) * (h + taps - 1) if w >= 4 else w * h
            tile = (32 + taps - 1) * 32
            lanes = next(g for g in (4, 8, 16, 32, 64) if (2 * g >= items and min(w, 32) * (min(h, 32) + taps - 1) <= tile * g // 64) or g == 64)
            if fx and fy:
                nb = 2 * ((w + taps - 1) * (h + taps - 1) + w * (h + taps - 1)) + 2 * (w * (h + taps - 1) + w * h)
            elif fx or fy:
                nb = 2 * (w * (h + taps - 1) + w * h)
            else:
                nb = 4 * w * h
            add("mc_luma" if luma else "mc_chroma", "hmr_gpu_mc_batch", (1 if luma else 0) | (lanes << 8), jb, n * nb)
        # The per-TU sequence predict -> transform -> quant -> [inv_quant -> itransform] -> reconst -> ssd16b (encode_intra_cu /
        # encode_inter_cu) is issued as ONE fused launch per TU size.  Counts come from the recorded mix: one chain per quant call;
        # the share of coded TUs is the recorded inv_quant / quant ratio; ssd16b calls beyond the chains stay separate jobs.
        TU_DTYPE = TU_JOB_DTYPE
        tot = lambda prefix, N: sum(v for k, v in calls.items() if k.split(":")[0] == prefix and int(k.split(":")[1]) == N)   # noqa: E731
        # prediction window = source window, plus strong noise in its lower half: TUs placed there are coded, TUs in the upper half are not
        noise = np.zeros((NCTU, 64, 64), np.int64)
        noise[:, 32:, :] = rng.integers(-40, 41, (NCTU, 32, 64))
        src_init = next(d for o, d in arena.init if o == srcw)
        pred_fused = arena.alloc(NCTU * P64, np.clip(src_init.reshape(NCTU, 64, 64) + noise, 0, 255).astype(np.int16).ravel())
        for N in (4, 8, 16, 32):
            nq = tot("quant", N)
            if not nq:
                continue
            assert nq == tot("transform", N) == tot("reconst", N), (N, nq, tot("transform", N), tot("reconst", N))
            assert tot("predict", N) == nq + sum(v for k2, v in calls.items() if k2.split(":")[0] == "inter_tu" and int(k2.split(":")[1]) == N), N
            coded_frac = tot("inv_quant", N) / nq
            parts = []
            for key, n in sorted(calls.items()):
                kp = key.split(":")
                if kp[0] != "quant" or int(kp[1]) != N:
                    continue
                comp, intra = int(kp[2]), int(kp[3])
                jb = np.zeros(n, TU_DTYPE)
                c = ctus(n)
                coded = rng.random(n) < coded_frac
                half = max(32 - N, 0) // N + 1
                x = rng.integers(0, (64 - N) // N + 1, n) * N
                y = rng.integers(0, half, n) * N + np.where(coded, 32 if N < 64 else 0, 0)
                y = np.minimum(y, 64 - N)
                pos_in = c * P64 + y * 64 + x
                jb["orig_off"] = srcw + pos_in; jb["orig_stride"] = 64
                jb["pred_off"] = pred_fused + pos_in; jb["pred_stride"] = 64
                jb["rec_off"] = recw + pos_in; jb["rec_stride"] = 64
                jb["p0"] = 3 | (comp << 2) | (intra << 4) | (0 << 5) | (1 << 6) | ((1 if (N == 4 and comp == 0 and intra) else 0) << 7)
                jb["p1"] = 5 | (2 << 8)
                parts.append((jb, c))
            jobs_all = np.concatenate([p[0] for p in parts])
            ctu_all = np.concatenate([p[1] for p in parts])
            lev_pool = arena.alloc(len(jobs_all) * N * N)
            jobs_all["lev_off"] = lev_pool + np.arange(len(jobs_all), dtype=np.int64) * N * N
            # algorithmic bytes of the seven calls the chain stands for (ABI width)
            n_coded = tot("inv_quant", N)
            nbytes = nq * (6 + 4 + 4 + 6 + 4) * N * N + n_coded * (4 + 4) * N * N
            merged[("tu_chain", N)] = {"name": "tu_chain", "fn": "hmr_gpu_tu_chain_batch", "size": N, "jobs": jobs_all, "ctu": ctu_all, "bytes": nbytes, "extra": ()}
            # stand-alone ssd16b calls that are not the tail of a chain
            extra_ssd = tot("ssd16b", N) - nq
            g = merged.get(("ssd16b", N))
            if g is not None:
                keep = max(extra_ssd, 0)
                g["jobs"], g["ctu"] = g["jobs"][:keep], g["ctu"][:keep]
                g["bytes"] = keep * (4 * N * N + 4)
                if keep == 0:
                    del merged[("ssd16b", N)]
            for name in ("transform", "quant", "inv_quant", "itransform", "reconst"):
                merged.pop((name, N), None)
            # predict also runs once per inter TU ahead of encode_inter_cu (the inter chain starts from the residual): those stay predict jobs
            g = merged.get(("predict", N))
            if g is not None:
                keep = tot("predict", N) - nq
                g["jobs"], g["ctu"] = g["jobs"][:keep], g["ctu"][:keep]
                g["bytes"] = keep * 6 * N * N
                if keep == 0:
                    del merged[("predict", N)]
    if fused and inter_source:
        # the predict calls left over are the CU-level residuals ahead of encode_inter: the inter TU jobs form them themselves and are priced with their bytes
        left = [k for k in merged if k[0] == "predict"]
        pred_bytes = sum(merged[k]["bytes"] for k in left)
        tus_ = [g for k, g in merged.items() if k[0] == "inter_tu"]
        if tus_ and left:
            total = sum(g["bytes"] for g in tus_)
            for g in tus_:
                g["bytes"] += int(pred_bytes * g["bytes"] / total)
            for k in left:
                del merged[k]
    for g in merged.values():      # a batch is issued in CTU order, like the host would enumerate it
        order = np.argsort(g["ctu"], kind="stable")
        g["jobs"] = np.ascontiguousarray(g["jobs"][order])
    return list(merged.values()), {"ref0": ref0}


def frame_side_info(rng):
    """Synthetic coding tree / motion side-info for the frame-level passes (shape of a cfg-2 P frame)."""
    W4, H4 = W // 4, HA // 4
    depth = np.kron(rng.integers(1, 4, (HA // 64 * 2, W // 64 * 2)), np.ones((8, 8), np.int64)).astype(np.uint8)[:H4, :W4]
    tr = (rng.random((H4 // 2, W4 // 2)) < 0.3).astype(np.uint8)
    tr = np.kron(tr, np.ones((2, 2), np.uint8))[:H4, :W4]
    cu = np.kron(rng.random((H4 // 2, W4 // 2)), np.ones((2, 2)))[:H4, :W4]
    intra = (cu < 0.08)
    cbf = (np.kron(rng.random((H4 // 2, W4 // 2)), np.ones((2, 2)))[:H4, :W4] < 0.5)
    flags = (intra * 1 + cbf * 2).astype(np.uint8)
    mvx = np.kron(rng.integers(-40, 41, (H4 // 4, W4 // 4)), np.ones((4, 4), np.int64)).astype(np.int16)[:H4, :W4]
    mvy = np.kron(rng.integers(-24, 25, (H4 // 4, W4 // 4)), np.ones((4, 4), np.int64)).astype(np.int16)[:H4, :W4]
    ref_idx = np.where(intra, -1, 0).astype(np.int8)
    qp = np.full((H4, W4), 32, np.uint8)
    n_ctu = (W // 64) * (HA // 64)
    params = np.zeros((n_ctu, 3, 34), np.int32)
    for c in range(n_ctu):
        for comp in range(3):
            params[c, comp, 0] = int(rng.random() < 0.6)
            t = int(rng.integers(0, 5))
            params[c, comp, 1] = t
            if t == 4:
                b = int(rng.integers(0, 28))
                params[c, comp, 2 + b:6 + b] = rng.integers(-4, 5, 4)
            else:
                params[c, comp, 2:7] = [3, 1, 0, -1, -3]
    return {"pred_depth": depth, "tr_idx": tr, "flags": flags, "mvx": mvx, "mvy": mvy, "ref_idx": ref_idx, "qp": qp, "sao_params": params}


def cpu_baseline(frames=None):
    """Reference encoder (compiled by oracle/Makefile in the build container, shipped in oracle/_ref) on this host's cores."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_lockstep")
    if not os.path.exists(exe):
        return None
    extra_cfg = REF_ARGS.get(WORKLOAD, ())
    if frames is None:
        frames = 1 if extra_cfg else (64 if W <= 1920 else 16)      # about 10-35 s of single-core encoding
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_yuv
    with tempfile.TemporaryDirectory() as td:
        clip = os.path.join(td, "clip.yuv")
        gen_yuv.write_clip(clip, W, H, frames)
        def run(extra):
            try:
                out = subprocess.run([exe, clip, "-", str(W), str(H), str(frames), *extra_cfg, *extra], capture_output=True, text=True, timeout=900).stdout
            except Exception:
                return None
            for line in out.splitlines():
                if line.startswith("LOCKSTEP"):
                    return dict(p.split("=") for p in line.split()[1:])
            return None
        kv = run(())
        # the reference's throughput mode inside one engine: one WPP thread per CTU row, as many as the host has cores for (not deterministic)
        threads = max(1, min((H + 63) // 64, os.cpu_count() or 1, 32))      # the reference caps WPP threads at 32 (hmr_private.h:1232)
        kv_mt = run((f"wpp={threads}",)) if threads > 1 else None
    if not kv:
        return None
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        cpu = "unknown"
    res = {"value": float(kv["fps"]), "unit": "frames/s", "cores": 1, "kind": "reference", "host_cpu": cpu,
           "sample": f"{kv['frames']} frames {W}x{H} " + (" ".join(extra_cfg) if extra_cfg else "cfg2 (IPPP QP32 qpel SAO)") + f", wpp=1 engines=1, {kv['seconds']} s, oracle/_ref/ref_lockstep"}
    if kv_mt:
        res["wpp_threads"] = {"value": float(kv_mt["fps"]), "cores": threads, "seconds": float(kv_mt["seconds"])}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--callmix-frame", type=int, default=2, help="which recorded P frame to replay")
    ap.add_argument("--workload", choices=list(WORKLOADS), default="cfg2-1080p-P-frame-replay")
    ap.add_argument("--mode", choices=["eager", "graph"], default="graph",
                    help="eager: C command list with an event pair around every launch inside the timed region (per-kernel roofline numbers are live); "
                         "graph: the same command list captured once into a hipGraph and replayed (no per-launch host cost), per-kernel numbers from an eager pass after the timed region")
    ap.add_argument("--unfused", action="store_true",
                    help="replay predict/transform/quant/inv_quant/itransform/reconst/ssd16b as seven separate batches per TU size instead of the fused TU-chain kernel")
    ap.add_argument("--issue-order", choices=["heavy-first", "recorded"], default="recorded",
                    help="order of the launches in the command list (issue order inside a graph branch): by descending algorithmic bytes, or in the order the groups were built")
    ap.add_argument("--schedule", choices=["time", "bytes"], default="bytes",
                    help="how the independent launches are dealt to the graph branches: by their measured isolated duration (one eager pass while the frame is set up) or by algorithmic bytes")
    ap.add_argument("--tu-multi", choices=["all", "upto16", "small", "off"], default="off",
                    help="fused TU chain batches (given prediction / intra / inter, all TU sizes) as segments of one launch: every size, sizes 4-16 (a launch with a "
                         "32x32 segment reserves that body's 52 KB of LDS for all), or one launch per batch")
    ap.add_argument("--multi-max-mb", type=float, default=100.0,
                    help="only batches of at most this many algorithmic MB become segments of a multi launch: merging pays for short launches (they cost queue slots, "
                         "not arithmetic); a batch that fills the GPU on its own is better off alone")
    ap.add_argument("--sao-offsets", action="store_true",
                    help="add the SAO offset derivation (hmr_gpu_sao_offsets_frame, a launch between SAO statistics and SAO apply) to the frame: not a table call of the "
                         "recorded mix, the device-side part of the SAO decision")
    ap.add_argument("--no-multi", action="store_true", help="one launch per (pixel kernel, block size) instead of one multi-segment launch per pixel kernel")
    ap.add_argument("--no-inter-source", action="store_true",
                    help="replay the CU-level `predict` calls ahead of encode_inter as their own jobs and feed the inter TU chains from the residual plane, instead of "
                         "letting the inter TU jobs form the residual from source and prediction")
    ap.add_argument("--no-chroma-driver", action="store_true",
                    help="replay the table calls of the chroma CU drivers (encode_intra_chroma) one by one instead of as search + TU launches per chroma CU size")
    ap.add_argument("--cu-driver", action="store_true",
                    help="issue the luma intra CU drivers (encode_intra_luma: search + transform tree + consolidation) as ordered device-side chains - search -> parent TUs -> "
                         "the four children in one launch -> consolidation, the mode handed over on the device - instead of independent search / TU batches")
    ap.add_argument("--cu-child-launches", action="store_true", help="issue the four children of the luma CU drivers as four launches instead of one launch with four rounds")
    ap.add_argument("--chain-branches", type=int, default=1, help="1: every luma CU driver chain runs on a graph branch of its own; 0: chains are balanced like the other launches")
    ap.add_argument("--branches", type=int, default=8, help="graph mode: number of parallel graph branches the independent launches are dealt to (1 = one serial chain)")
    ap.add_argument("--engines-per-gpu", type=int, default=1,
                    help="encoder engines (frames in flight) per GPU, each with its own stream, planes and command list; a step encodes that many frames. "
                         "The reference runs up to 8 engines on consecutive frames (num_enc_engines); an IPPP chain keeps about 3 usefully in flight at 1080p (SURVEY.md 8-e)")
    ap.add_argument("--launch-order", default=None, help="write the per-step kernel launch order (JSON) for tools/pmc_summary.py")
    args = ap.parse_args()
    set_workload(args.workload)

    import torch
    import torch.distributed as dist
    from homerhevc_amd.engines import exchange_reference
    from gpu_abi import Context, Frame, Units

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the MI355X backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    P = C.c_void_p

    class Segment(C.Structure):      # hmr_gpu_segment
        _fields_ = [("jobs", P), ("out", P), ("njobs", C.c_int), ("size", C.c_int)]

    class TuSegment(C.Structure):    # hmr_gpu_tu_segment
        _fields_ = [("jobs", P), ("ssd", P), ("ac_sum", P), ("modes", P), ("njobs", C.c_int), ("size", C.c_int), ("kind", C.c_int), ("rounds", C.c_int)]

    class Cmd(C.Structure):
        _fields_ = [("op", C.c_int), ("njobs", C.c_int), ("size", C.c_int), ("p", C.c_int * 4), ("jobs", P), ("a", P), ("b", P), ("c", P), ("out", P), ("p64", P * 3), ("branch", C.c_int)]

    calls = load_callmix(args.callmix_frame)

    def make_engine(e):
        """One encoder engine = one frame in flight: its own stream, context, planes, job arrays and command list."""
        stream = torch.cuda.Stream(device=dev)
        ctx = Context(device=local_rank, stream=stream.cuda_stream)

        rng = np.random.default_rng(1234 + rank + 1000 * e)
        arena = Arena()
        groups, planes = build_groups(calls, rng, arena, fused=not args.unfused, cu_driver=args.cu_driver, cu_rounds=not args.cu_child_launches, chroma_driver=not args.no_chroma_driver, inter_source=not args.no_inter_source)
        if os.environ.get("HOMER_BENCH_DROP"):      # experiments only (what would a launch cost if it were free?): the record is marked
            drop = set(os.environ["HOMER_BENCH_DROP"].split(","))
            groups = [g for g in groups if g["name"] not in drop and f"{g['name']}:{g['size']}" not in drop]
        info = frame_side_info(rng)

        with torch.cuda.stream(stream):
            host = np.zeros(arena.size, np.int16)
            for off, data in arena.init:
                host[off:off + data.size] = data
            d_arena = torch.from_numpy(host).to(dev)
            base = d_arena.data_ptr()
            for g in groups:
                g["d_jobs"] = torch.from_numpy(g["jobs"].view(np.uint8)).to(dev)
                g["d_out"] = torch.zeros(len(g["jobs"]), dtype=torch.int32, device=dev)
            if not args.no_multi:
                # batches of one pixel kernel that differ only in the block size go out as segments of ONE launch (hmr_gpu_pixel_multi): each is a
                # short launch that cannot fill the GPU on its own
                PIXEL_OPS = {"hmr_gpu_sad_batch": 1, "hmr_gpu_ssd16b_batch": 2, "hmr_gpu_predict_batch": 3, "hmr_gpu_reconst_batch": 4, "hmr_gpu_copy_batch": 5}
                fam = {}
                for g in groups:
                    if g["fn"] in PIXEL_OPS and g["size"] in (4, 8, 16, 32, 64) and not g.get("chain") and g["bytes"] <= args.multi_max_mb * 1e6:
                        fam.setdefault(g["fn"], []).append(g)
                # ... and the fused TU chains (given prediction / intra / inter, every TU size) as segments of one launch (hmr_gpu_tu_chain_multi)
                TU_KIND = {"hmr_gpu_tu_chain_batch": 0, "hmr_gpu_intra_tu_chain_batch": 1, "hmr_gpu_inter_tu_chain_batch": 2}
                tus = [g for g in groups if g["fn"] in TU_KIND and not g.get("chain") and (g["size"] < 32 or args.tu_multi == "all") and
                       (args.tu_multi != "small" or g["size"] == 4 or g["fn"] == "hmr_gpu_tu_chain_batch")]
                if args.tu_multi != "off" and 2 <= len(tus) <= 8:
                    tus.sort(key=lambda g: -g["bytes"])          # blocks are dispatched in segment order: the long batches first, the short ones fill the tail
                    for g in tus:
                        g["d_ac"] = torch.zeros(len(g["jobs"]), dtype=torch.int32, device=dev)
                    tsegs = (TuSegment * len(tus))(*[TuSegment(m["d_jobs"].data_ptr(), m["d_out"].data_ptr(), m["d_ac"].data_ptr(), None, len(m["jobs"]), m["size"],
                                                                 TU_KIND[m["fn"]], 0) for m in tus])
                    merged_g = {"name": "tu_chains", "fn": "hmr_gpu_tu_chain_multi", "size": "multi", "segs": tsegs, "members": tus,
                                "jobs": np.zeros(sum(len(m["jobs"]) for m in tus), np.uint8), "bytes": sum(m["bytes"] for m in tus), "extra": (), "d_jobs": None,
                                "d_out": tus[0]["d_out"]}
                    groups[groups.index(tus[0])] = merged_g
                    for m in tus[1:]:
                        groups.remove(m)
                for fn, members in fam.items():
                    if len(members) < 2:
                        continue
                    segs = (Segment * len(members))(*[Segment(m["d_jobs"].data_ptr(), m["d_out"].data_ptr(), len(m["jobs"]), m["size"]) for m in members])
                    merged_g = {"name": members[0]["name"], "fn": "hmr_gpu_pixel_multi", "size": "multi", "pixel_op": PIXEL_OPS[fn], "segs": segs, "members": members,
                                "jobs": np.concatenate([m["jobs"] for m in members]), "bytes": sum(m["bytes"] for m in members), "extra": (),
                                "d_jobs": None, "d_out": members[0]["d_out"]}
                    groups[groups.index(members[0])] = merged_g
                    for m in members[1:]:
                        groups.remove(m)
            # frame-level state: original + reconstruction (padded) + SAO destination, side-info
            def padded_plane(w, h, pad):
                return torch.from_numpy(rng.integers(0, 256, ((h + 2 * pad), (w + 2 * pad))).astype(np.int16)).to(dev)
            rec_pl = [padded_plane(W, H, PAD), padded_plane(W // 2, H // 2, PAD // 2), padded_plane(W // 2, H // 2, PAD // 2)]
            org_pl = [padded_plane(W, H, PAD), padded_plane(W // 2, H // 2, PAD // 2), padded_plane(W // 2, H // 2, PAD // 2)]
            # the SAO output / next reference picture: its three padded planes live in ONE buffer, so the engine-to-engine exchange is a single
            # send and a single receive per picture
            sizes = [t.numel() for t in rec_pl]
            dst_flat = torch.cat([t.reshape(-1) for t in rec_pl])
            nxt_flat = torch.empty_like(dst_flat)              # reference picture received from the previous engine
            dst_pl = [v.view_as(t) for v, t in zip(torch.split(dst_flat, sizes), rec_pl)]
            d_info = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in info.items()}
            n_ctu = info["sao_params"].shape[0]
            d_stats = torch.zeros(n_ctu * 3 * 5 * 2 * 32, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

        def frame_of(pl):
            f = Frame()
            f.width, f.height = W, H
            f.y = pl[0].data_ptr() + 2 * (PAD * (W + 2 * PAD) + PAD)
            f.u = pl[1].data_ptr() + 2 * ((PAD // 2) * (W // 2 + PAD) + PAD // 2)
            f.v = pl[2].data_ptr() + 2 * ((PAD // 2) * (W // 2 + PAD) + PAD // 2)
            f.stride_y, f.stride_c = W + 2 * PAD, W // 2 + PAD
            return f

        f_rec, f_org, f_dst = frame_of(rec_pl), frame_of(org_pl), frame_of(dst_pl)
        units = Units(W // 4, d_info["mvx"].data_ptr(), d_info["mvy"].data_ptr(), d_info["ref_idx"].data_ptr(), d_info["qp"].data_ptr(), d_info["flags"].data_ptr())
        OPS = {"hmr_gpu_sad_batch": 1, "hmr_gpu_ssd16b_batch": 2, "hmr_gpu_predict_batch": 3, "hmr_gpu_reconst_batch": 4, "hmr_gpu_copy_batch": 5,
               "hmr_gpu_intra_pred_batch": 7, "hmr_gpu_intra_refs_batch": 8, "hmr_gpu_interpolate_batch": 9, "hmr_gpu_transform_batch": 11,
               "hmr_gpu_itransform_batch": 12, "hmr_gpu_quant_batch": 13, "hmr_gpu_inv_quant_batch": 14, "hmr_gpu_mc_batch": 15,
               "hmr_gpu_motion_estimation_batch": 16, "hmr_gpu_tu_chain_batch": 22, "hmr_gpu_intra_search_batch": 23, "hmr_gpu_intra_tu_chain_batch": 24, "hmr_gpu_inter_tu_chain_batch": 25,
               "hmr_gpu_intra_tu_chain_modes_batch": 24, "hmr_gpu_tree_decide_batch": 26, "hmr_gpu_chroma_search_batch": 29}
        OP_EDGE, OP_DEBLOCK, OP_STATS, OP_APPLY, OP_PAD = 17, 18, 19, 20, 21
        cmds, names = [], []
        if args.issue_order == "heavy-first":
            # list order is issue order inside a branch and the order the graph's root nodes go out: the long launches first, so that the GPU is full
            # while the short ones ramp up and the tail of the frame is made of short launches.  Chains keep their internal order.
            unit_bytes = {}
            for g in groups:
                key = g.get("chain") or id(g)
                unit_bytes[key] = unit_bytes.get(key, 0) + g["bytes"]
            order_ = sorted(range(len(groups)), key=lambda i: (-unit_bytes[groups[i].get("chain") or id(groups[i])], i))
            groups[:] = [groups[i] for i in order_]
        # a luma CU driver chain shares its search results (the modes), the SSD / sum arrays of its five TU levels and the consolidation results
        chains = {}
        for g in groups:
            if g.get("chain") and g["chain"] not in chains:
                m = len(g["jobs"])
                chains[g["chain"]] = {"m": m, "modes": torch.zeros(4 * m, dtype=torch.int32, device=dev), "ssd": torch.zeros(8 * m, dtype=torch.int32, device=dev),
                                      "ac": torch.zeros(8 * m, dtype=torch.int32, device=dev), "res": torch.zeros(4 * m, dtype=torch.int32, device=dev)}
        for g in groups:
            if g["fn"] == "hmr_gpu_tu_chain_multi":
                cm = Cmd(op=28, njobs=len(g["segs"]), jobs=C.addressof(g["segs"]), a=base, b=base, c=base)
                cm.p64 = (P * 3)(base, None, None)
                cmds.append(cm)
                names.append(f"{g['name']}:multi")
                continue
            if g["fn"] == "hmr_gpu_pixel_multi":
                cmds.append(Cmd(op=27, njobs=len(g["segs"]), size=g["pixel_op"], jobs=C.addressof(g["segs"]), a=base, b=base, c=base))
                names.append(f"{g['name']}:multi")
                continue
            cm = Cmd(op=OPS[g["fn"]], njobs=g.get("njobs", len(g["jobs"])), size=g["size"], jobs=g["d_jobs"].data_ptr(), a=base, b=base, c=base, out=g["d_out"].data_ptr())
            if g["fn"] == "hmr_gpu_copy_batch":
                cm.size = g["size"] << 8            # kind 0 (int16) | uniform square size hint
            if g["fn"] == "hmr_gpu_quant_batch":
                cm.b = None                          # deltaU is scratch in the reference; not returned
            if g["fn"] == "hmr_gpu_motion_estimation_batch":
                g["d_out"] = torch.zeros(5 * len(g["jobs"]), dtype=torch.int32, device=dev)     # hmr_gpu_me_result per PU
                cm.out = g["d_out"].data_ptr()
                cm.p = (C.c_int * 4)(128, 64, W, HA)     # MOTION_SEARCH_RANGE_X/Y, picture size
            if g["fn"] == "hmr_gpu_intra_search_batch":
                g["d_out"] = chains[g["chain"]]["modes"] if g.get("chain") else torch.zeros(4 * len(g["jobs"]), dtype=torch.int32, device=dev)   # hmr_gpu_intra_result per PU
                cm.out = g["d_out"].data_ptr()
            if g["fn"] == "hmr_gpu_chroma_search_batch":
                cm.out = chains[g["chain"]]["modes"].data_ptr()
                cm.p64 = (P * 3)(None, None, None)       # luma modes: given in the jobs here
            if g["fn"] == "hmr_gpu_intra_tu_chain_modes_batch":
                ch, k = chains[g["chain"]], g["level"]
                so_ = 4 * g.get("ssd_off", k * ch["m"])
                cm.out = ch["ssd"].data_ptr() + so_
                cm.p64 = (P * 3)(base, ch["ac"].data_ptr() + so_, ch["modes"].data_ptr())
                cm.p = (C.c_int * 4)(g.get("rounds", 1), 0, 0, 0)
            if g["fn"] == "hmr_gpu_tree_decide_batch":
                ch = chains[g["chain"]]
                cm.a, cm.b, cm.c, cm.out = ch["ssd"].data_ptr(), ch["ac"].data_ptr(), base, ch["res"].data_ptr()
                cm.p64 = (P * 3)(base, None, None)
            if g["fn"] in ("hmr_gpu_intra_tu_chain_batch", "hmr_gpu_inter_tu_chain_batch"):
                g["d_ac"] = torch.zeros(len(g["jobs"]), dtype=torch.int32, device=dev)
                cm.p64 = (P * 3)(base, g["d_ac"].data_ptr(), None)   # reconstruction / prediction base, ac_sum
            if g["fn"] == "hmr_gpu_tu_chain_batch":
                g["d_ac"] = torch.zeros(len(g["jobs"]), dtype=torch.int32, device=dev)
                cm.p64 = (P * 3)(base, g["d_ac"].data_ptr(), None)   # reconstruction base, ac_sum
            cmds.append(cm)
            names.append(f"{g['name']}:{g['size']}")
        # The launches of one replayed frame have no data dependencies on each other except the in-loop filter chain (edge flags ->
        # deblock -> SAO stats -> SAO apply -> pad), which stays on branch 0 in order.  The batched groups are dealt to `--branches`
        # graph branches (longest first, by algorithmic bytes) so the ramp-up / tail of one kernel overlaps the body of another.
        load = [0.0] * max(args.branches, 1)
        load[0] = 2.0e8                                  # the frame-level chain
        sched = {}
        for i, g in enumerate(groups):                   # the launches of a CU driver chain depend on each other: one branch, list order
            sched.setdefault(g.get("chain") or i, []).append(i)
        extra_branch = len(load)
        for unit in sorted(sched.values(), key=lambda u: -sum(groups[i]["bytes"] for i in u)):
            if len(unit) > 1 and args.chain_branches:    # a chain is seven short dependent launches: latency, not bytes - it gets a branch of its own
                b = extra_branch
                extra_branch += 1
            else:
                b = load.index(min(load))
                load[b] += sum(groups[i]["bytes"] for i in unit)
            for i in unit:
                cmds[i].branch = b
        frame_bytes = {
            "deblock": 2 * 2 * 6144 * n_ctu, "sao_stats": (2 * 2 * 6144 + 5 * 3 * 512) * n_ctu, "sao_apply": 2 * 2 * 6144 * n_ctu,
            "pad": 2 * 2 * ((W + 2 * PAD) * (H + 2 * PAD) - W * H) * 3 // 2, "edge_flags": 3 * (W // 4) * (H // 4), "sao_offsets": (960 * 4 + 3 * 8 + 15 * (128 + 4 + 8)) * n_ctu,
        }
        cmds.append(Cmd(op=OP_EDGE, p=(C.c_int * 4)(W, H, W // 4, 0), a=d_info["pred_depth"].data_ptr(), b=d_info["tr_idx"].data_ptr(), c=d_info["flags"].data_ptr()))
        cmds.append(Cmd(op=OP_DEBLOCK, p=(C.c_int * 4)(2, 2, 0, 0), a=C.addressof(f_rec), b=C.addressof(units)))
        cmds.append(Cmd(op=OP_STATS, a=C.addressof(f_org), b=C.addressof(f_rec), out=d_stats.data_ptr()))
        if not (not args.sao_offsets):
            # SAO offset derivation of every (CTU, component, type) from the statistics just produced (hmr_sao.c:480-659): the device-side part of the SAO decision
            sao_lambdas = torch.full((n_ctu, 3), 56.0, dtype=torch.float64, device=dev)       # 0.4624 * 1.4^((32 - 12) / 1.4), hmr_wpp_sao_ctu
            sao_off = torch.zeros(n_ctu * 15 * 32, dtype=torch.int32, device=dev); sao_aux = torch.zeros(n_ctu * 15, dtype=torch.int32, device=dev)
            sao_dist = torch.zeros(n_ctu * 15, dtype=torch.int64, device=dev)
            cm = Cmd(op=30, njobs=n_ctu, a=d_stats.data_ptr(), b=sao_lambdas.data_ptr(), c=sao_off.data_ptr(), out=sao_aux.data_ptr())
            cm.p64 = (P * 3)(sao_dist.data_ptr(), None, None)
            cmds.append(cm)
            chains["sao_offsets_buffers"] = [sao_lambdas, sao_off, sao_aux, sao_dist]
        cmds.append(Cmd(op=OP_APPLY, a=C.addressof(f_rec), b=C.addressof(f_dst), c=d_info["sao_params"].data_ptr()))
        cmds.append(Cmd(op=OP_PAD, p=(C.c_int * 4)(PAD, PAD, 0, 0), a=C.addressof(f_dst)))
        names_tail = ["edge_flags", "deblock", "sao_stats"] + ([] if (not args.sao_offsets) else ["sao_offsets"]) + ["sao_apply", "pad"]
        names += names_tail
        if args.schedule == "time" and args.mode == "graph" and args.branches > 1:
            # measured schedule: one eager pass with an event pair per command gives each launch's isolated duration; the launches are then dealt to the
            # branches longest first by TIME (the runtime runs about four kernels at once, so what shares a queue matters more than bytes).
            # Part of setting the frame up - it happens once, before the warm-up.
            tmp_arr = (Cmd * len(cmds))(*cmds)
            for c_ in tmp_arr:
                c_.branch = 0
            tmp = P()
            ctx.call("hmr_gpu_cmdlist_create", tmp_arr, len(cmds), C.byref(tmp))
            n_c = len(cmds)
            with torch.cuda.stream(stream):
                evs = [(P * (2 * n_c))(*[ctx.event().value for _ in range(2 * n_c)]) for _ in range(4)]
                for r_ in range(4):
                    ctx.call("hmr_gpu_cmdlist_run", tmp, evs[r_])
                torch.cuda.synchronize()
            t_ms = [sorted(ctx.elapsed(P(evs[r_][2 * k]), P(evs[r_][2 * k + 1])) for r_ in range(1, 4))[1] for k in range(n_c)]
            ctx.lib.hmr_gpu_cmdlist_destroy.restype = None
            ctx.lib.hmr_gpu_cmdlist_destroy(tmp)
            tload = [0.0] * args.branches
            tload[0] = sum(t_ms[len(groups):])               # the frame-level chain stays on branch 0
            for unit in sorted(sched.values(), key=lambda u: -sum(t_ms[i] for i in u)):
                b = tload.index(min(tload))
                tload[b] += sum(t_ms[i] for i in unit)
                for i in unit:
                    cmds[i].branch = b
        cmd_arr = (Cmd * len(cmds))(*cmds)
        clist = P()
        ctx.call("hmr_gpu_cmdlist_create", cmd_arr, len(cmds), C.byref(clist))
        n_cmd = len(cmds)
        return {"ctx": ctx, "stream": stream, "clist": clist, "groups": groups, "names": names, "frame_bytes": frame_bytes, "dst_pl": [dst_flat], "nxt_pl": [nxt_flat],
                "n_cmd": n_cmd, "keep": [d_arena, rec_pl, org_pl, d_info, d_stats, f_rec, f_org, f_dst, units, cmd_arr, cmds, chains]}

    engines = [make_engine(e) for e in range(max(args.engines_per_gpu, 1))]
    E0 = engines[0]
    ctx, stream, clist, groups, names, frame_bytes, n_cmd = E0["ctx"], E0["stream"], E0["clist"], E0["groups"], E0["names"], E0["frame_bytes"], E0["n_cmd"]
    # one event pair per command and timed step (engine 0)
    ev = [(P * (2 * n_cmd))(*[ctx.event().value for _ in range(2 * n_cmd)]) for _ in range(args.steps)]

    def step(idx, timed):
        for k, eng in enumerate(engines):
            if args.mode == "graph":
                eng["ctx"].call("hmr_gpu_cmdlist_replay", eng["clist"])
            else:
                eng["ctx"].call("hmr_gpu_cmdlist_run", eng["clist"], ev[idx] if (timed and k == 0) else None)
        # reconstructed reference picture: engine r -> engine r+1, point-to-point over RCCL/xGMI, posted AFTER the frame's work on this stream (the
        # transfer is ordered behind what is already on the stream: the offsets and padding that write the picture) and waited for before the next step
        reqs = exchange_reference(E0["dst_pl"], E0["nxt_pl"], rank, world)
        for r in reqs:
            r.wait()

    if args.launch_order and rank == 0:
        order = [f"{g['name']}:{g['size']}" for g in groups] + ["edge_flags", "deblock", "deblock", "sao_stats"] + ([] if (not args.sao_offsets) else ["sao_offsets"]) + ["sao_apply", "pad", "pad", "pad"]   # kernels, not commands
        with open(args.launch_order, "w") as f:
            json.dump(order, f)

    with torch.cuda.stream(stream):
        for i in range(args.warmup):
            step(-1 - i, False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # device-to-device copy rate of this GPU (confirm the nominal HBM peak on the box): 1 GiB read + 1 GiB written
    with torch.cuda.stream(stream):
        src_probe = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst_probe = torch.empty_like(src_probe)
        dst_probe.copy_(src_probe)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            dst_probe.copy_(src_probe)
        e1.record(stream)
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src_probe, dst_probe

    # VALU issue rate of this GPU, measured: 16 wavefronts per SIMD issuing nothing but independent packed dot products (hmr_gpu_valu_probe)
    with torch.cuda.stream(stream):
        probe_out = torch.zeros(4, dtype=torch.int32, device=dev)
        blocks, iters = 4096, 4096
        rates = {}
        for kind, nm in ((0, "v_dot2_i32_i16"), (1, "v_sad_u16")):
            ctx.call("hmr_gpu_valu_probe", kind, blocks, 64, probe_out.data_ptr())
            pe = [ctx.event() for _ in range(2)]
            ctx.record(pe[0]); ctx.call("hmr_gpu_valu_probe", kind, blocks, iters, probe_out.data_ptr()); ctx.record(pe[1])
            torch.cuda.synchronize()
            rates[nm] = blocks * 4 * iters * 8 / (ctx.elapsed(pe[0], pe[1]) * 1e-3) / 1e9       # G wavefront-instructions / s
    valu_probe = {"Ginst_per_s": {k: round(v, 1) for k, v in rates.items()}, "nominal_Ginst_per_s": 1024 * 2.4 / 4,
                  "note": "wave64 VALU instructions per second over the whole GPU; nominal = 1024 SIMDs x 2.4 GHz / 4 cycles"}

    # what an event pair around a launch measures beyond the kernel: empty launches timed the same way
    with torch.cuda.stream(stream):
        evn = [ctx.event() for _ in range(42)]
        for i in range(0, 42, 2):
            ctx.record(evn[i]); ctx.call("hmr_gpu_nop"); ctx.record(evn[i + 1])
        torch.cuda.synchronize()
    nop_ms = sorted(ctx.elapsed(evn[i], evn[i + 1]) for i in range(2, 42, 2))[10]     # median of 20

    # per-kernel durations from the event pairs around every command
    if args.mode == "graph":      # the graph has no event nodes: one eager pass of the same steps right after the timed region
        with torch.cuda.stream(stream):
            for i in range(args.steps):
                ctx.call("hmr_gpu_cmdlist_run", clist, ev[i])
            torch.cuda.synchronize()
    # median over the steps: one disturbed pass (a host hiccup between two launches) must not pick the launch the roofline object describes
    samples = {n: [] for n in names}
    for i in range(args.steps):
        for k, n in enumerate(names):
            samples[n].append(ctx.elapsed(P(ev[i][2 * k]), P(ev[i][2 * k + 1])))
    per = {k: sorted(v)[len(v) // 2] for k, v in samples.items()}
    nbytes = {f"{g['name']}:{g['size']}": g["bytes"] for g in groups}
    nbytes.update(frame_bytes)
    dom = max(per, key=per.get)
    achieved = nbytes[dom] / (per[dom] * 1e-3) / 1e9 if per[dom] > 0 else 0.0
    # HBM traffic of the dominant kernel from the committed PMC passes of this same command (tools/pmc_summary.py), if present
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    profiled = WORKLOAD == "cfg2-1080p-P-frame-replay" and args.callmix_frame == 2 and not args.unfused   # what the committed counter passes ran
    if os.path.exists(tpath) and profiled:
        with open(tpath) as f:
            traffic = json.load(f).get("groups", {}).get(dom, {}).get("hbm_bytes")
    # VALU issue floor of the dominant kernel from the committed SQ counter passes (tools/pmc_sq.sh): a wave64 VALU instruction holds
    # its SIMD16 for 4 cycles, the chip has 256 CUs x 4 SIMDs at 2.4 GHz
    valu = None
    spath = os.path.join(ROOT, "profiles", "sq_summary.csv")
    if os.path.exists(spath) and os.path.exists(tpath) and profiled:
        import csv
        with open(tpath) as f:
            kname = json.load(f).get("groups", {}).get(dom, {}).get("kernel", "").replace("void ", "")
        for r in csv.DictReader(open(spath)):
            if r["kernel"] == kname and r.get("SQ_INSTS_VALU") and r.get("SQ_WAVES"):
                insts = float(r["SQ_INSTS_VALU"])
                floor_ms = insts * 4 / (256 * 4) / 2.4e9 * 1e3
                valu = {"valu_insts_per_launch": int(insts), "waves_per_launch": int(float(r["SQ_WAVES"])), "issue_floor_ms": round(floor_ms, 5),
                        "frac_of_issue_peak": round(floor_ms / per[dom], 4) if per[dom] > 0 else None,
                        "frac_of_measured_issue_rate": round(insts / (rates["v_dot2_i32_i16"] * 1e9) * 1e3 / per[dom], 4) if per[dom] > 0 else None}
    total_alg = sum(nbytes.values())

    if rank == 0:
        fps = args.steps * world * len(engines) / elapsed
        line = {
            "metric": f"encoded frames/sec, {H}p YUV420 fixed-QP IPPP (hot-path replay of the reference's per-frame call mix)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "config": {"workload": WORKLOAD, "width": W, "height": H, "qp": 32, "gop": "IPPP gop_size=1", "me": "quarter-pel", "sao": 1,
                       "calls_per_frame": int(sum(len(g["jobs"]) for g in groups)), "launches_per_frame": len(groups) + 8 + (0 if (not args.sao_offsets) else 1),
                       "replay": {"fused_call_sequences": not args.unfused, "chroma_cu_drivers": not args.no_chroma_driver and not args.unfused,
                                  "inter_residual_in_kernel": not args.no_inter_source and not args.unfused, "luma_cu_driver_chains": bool(args.cu_driver),
                                  "multi_segment_launches_up_to_MB": None if args.no_multi else args.multi_max_mb, "sao_offsets_launch": bool(args.sao_offsets)}, **({"EXPERIMENT_dropped_groups": os.environ["HOMER_BENCH_DROP"]} if os.environ.get("HOMER_BENCH_DROP") else {}),
                       "callmix_frame": args.callmix_frame, "parallelism": f"{len(engines)} engine(s) per gpu x{world}", "frames_per_step": len(engines) * world, "launch_mode": args.mode, "graph_branches": args.branches if args.mode == "graph" else 1, "tu_chain": "7 separate batches" if args.unfused else "fused kernel",
                       "algorithmic_MB_per_frame_abi_width": round(total_alg / 1e6, 2), "compulsory_MB_per_frame": round(10.5 * W * H / 1e6, 2)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_frac": round(traffic / (per[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if (traffic and per[dom] > 0) else None,
                         "measured_copy_GBps": round(copy_gbs, 1), "valu_issue": valu, "valu_probe": valu_probe,
                         "timing": ("HIP event pairs around every launch inside the timed region" if args.mode == "eager" else
                                    "HIP event pairs around every launch, eager replay of the same K steps right after the timed graph replays "
                                    "(event nodes inside a hipGraph cannot be read back on ROCm 7.2)"),
                         "bytes_per_launch": int(nbytes[dom]), "ms_per_launch": round(per[dom], 5), "ms_event_pair_empty_launch": round(nop_ms, 5),
                         "frame_level_frac": round(10.5 * W * H * fps / world / 1e9 / HBM_PEAK_GBS, 6)},
            "kernels_ms": {k: round(v, 4) for k, v in sorted(per.items(), key=lambda kv: -kv[1])},
            "kernels_gbs": {k: round(nbytes[k] / (v * 1e-3) / 1e9, 1) for k, v in sorted(per.items(), key=lambda kv: -kv[1]) if v > 0},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline()
            line["cpu_baseline"] = cb if cb else {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference", "sample": "oracle/_ref/ref_lockstep not shipped"}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
