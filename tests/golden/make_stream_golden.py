#!/usr/bin/env python3
"""Mint the bitstream fixtures from the COMPILED REFERENCE (build container only): oracle/_ref/ref_lockstep encodes the synthetic
clip of tools/gen_yuv.py (wpp = 1, engines = 1: the deterministic mode) and this script records, per case, the md5 of the .265
stream, the NAL unit sizes and the md5 of every reconstructed picture.  tests/golden/streams.json is what the free-running encoder
(checker build on the CPU, the device path under -m gpu) has to reproduce byte for byte."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_yuv  # noqa: E402
import stream_diff  # noqa: E402

CASES = [
    ("200x136", 200, 136, 3, {}),
    ("416x240", 416, 240, 5, {}),
    ("416x240_nosao", 416, 240, 3, {"sao": 0}),
    ("416x240_qp22_perf0", 416, 240, 3, {"qp": 22, "perf": 0}),
    ("328x264_qp38_nosbh", 328, 264, 3, {"qp": 38, "sign_hiding": 0}),
    ("416x240_intra_period1", 416, 240, 2, {"intra_period": 1}),   # the reference clamps intra_period to gop_size + 1 = 2 (hmr_encoder_lib.c:743): I, P
    ("832x480", 832, 480, 3, {}),
    # found by tools/encoder_fuzz.py: two identical merge candidates, the first one's coded evaluation without levels and best - the second one's no-residual evaluation
    # is then the first of that prediction and wins (enc_ctu.h check_rd_cost_merge)
    ("392x136_qp22_clip931814", 392, 136, 4, {"qp": 22, "clip_seed": 931814}),
    # found by tools/encoder_fuzz.py --gpu: a merge candidate 122 samples outside the right picture edge is evaluated on what the THREAD's prediction window holds (Q12)
    ("400x104_qp22_perf0_nosao_wpp_rows_clip657909", 400, 104, 3, {"qp": 22, "perf": 0, "sao": 0, "wpp": 2, "clip_seed": 657909}),
    ("1920x1080_cfg2", 1920, 1080, 8, {}),       # BASELINE.json configs[1]
    ("1280x720_intra_period1", 1280, 720, 2, {"intra_period": 1}),   # I, P at 720p (see above)
    # forced intra pictures (encoder_in_out_t.image_type = IMAGE_I on every frame, homer_hevc_enc_api.h:112, honoured hmr_encoder_lib.c:311-313): consecutive I frames
    ("1280x720_force_intra", 1280, 720, 4, {"force_intra": 1}),   # BASELINE.json configs[0]: 720p all-intra, QP 32, one thread
    ("416x240_force_intra", 416, 240, 4, {"force_intra": 1}),
    ("416x240_force_intra_wpp_rows", 416, 240, 4, {"force_intra": 1, "wpp": 4}),
    # BASELINE.json configs[4]: all-intra, rd = 1 (RD_FULL: the mode and transform-tree decisions price their syntax with the CABAC bit counter), intra TU depth 4
    ("416x240_force_intra_rdfull_tr4_wpp_rows", 416, 240, 3, {"force_intra": 1, "rd": 1, "intra_tr": 4, "wpp": 4}),
    ("416x240_force_intra_rdfull_wpp_rows", 416, 240, 2, {"force_intra": 1, "rd": 1, "wpp": 4}),
    ("328x264_force_intra_rdfull_tr3_perf0_wpp_rows", 328, 264, 2, {"force_intra": 1, "rd": 1, "intra_tr": 3, "perf": 0, "wpp": 5}),
    ("3840x2160_force_intra_rdfull_tr4", 3840, 2160, 2, {"force_intra": 1, "rd": 1, "intra_tr": 4, "wpp": 32}),
    ("416x240_rdfull_wpp_rows", 416, 240, 5, {"rd": 1, "wpp": 4}),      # IPPP with RD_FULL: the intra CUs of P frames price their syntax too
    ("832x480_rdfull_tr3_wpp_rows", 832, 480, 4, {"rd": 1, "intra_tr": 3, "wpp": 8}),
    ("3840x2160_cfg2", 3840, 2160, 2, {}),       # configs[3], one engine's share: the cfg-2 encode at 2160p (I + P)
    ("200x136_scene_cut", 200, 136, 27, {"cut_at": 24}),     # new scene at frame 24: in-frame scene-change detection (hmr_motion_inter.c:3791), frames 25-26 after it
    ("416x240_scene_cut", 416, 240, 25, {"cut_at": 23}),
    # wfpp_num_threads = CTU rows: the reference is not deterministic with several threads; oracle/ref_ctudump.c's HOMER_TURNSTILE forces the synchronous-wavefront
    # schedule on it (one legal interleaving, the one a row-parallel device executes) and these fixtures are what it then produces
    ("416x240_wpp_rows", 416, 240, 5, {"wpp": 4}),
    ("416x240_scene_cut_wpp_rows", 416, 240, 25, {"cut_at": 23, "wpp": 4}),
    ("1920x1080_cfg2_wpp_rows", 1920, 1080, 8, {"wpp": 17}),
    ("832x480_wpp_rows", 832, 480, 3, {"wpp": 8}),
    ("416x240_qp22_perf0_wpp_rows", 416, 240, 3, {"qp": 22, "perf": 0, "wpp": 4}),
    ("416x240_nosao_wpp_rows", 416, 240, 3, {"sao": 0, "wpp": 4}),
    # fewer threads than rows (2 x threads >= CTU columns keeps the synchronous wavefront one of the reference's own interleavings): threads own rows k, k + N
    ("328x264_wpp3", 328, 264, 4, {"wpp": 3}),
    ("200x136_wpp2", 200, 136, 4, {"wpp": 2}),
    ("3840x2160_cfg2_wpp32", 3840, 2160, 3, {"wpp": 32}),
    # num_enc_engines > 1: the reference's frame pipeline is not deterministic either; oracle/ref_ctudump.c's engine turnstile pins it (frame n starts from the avg_dist frame n - E
    # left and from engine n mod E's own state, on the complete reconstruction of frame n - 1) - with one WPP thread, and combined with the synchronous wavefront
    ("416x240_eng2", 416, 240, 10, {"engines": 2}),
    ("416x240_eng3_wpp_rows", 416, 240, 10, {"engines": 3, "wpp": 4}),
    ("832x480_eng2_wpp_rows", 832, 480, 6, {"engines": 2, "wpp": 8}),
    ("416x240_eng4_wpp_rows", 416, 240, 14, {"engines": 4, "wpp": 4}),
    ("416x240_scene_cut_eng2_wpp_rows", 416, 240, 27, {"cut_at": 23, "engines": 2, "wpp": 4}),
    ("1920x1080_cfg2_eng2", 1920, 1080, 8, {"engines": 2, "wpp": 17}),
    ("1920x1080_cfg2_eng3", 1920, 1080, 8, {"engines": 3, "wpp": 17}),
    ("3840x2160_cfg2_eng8", 3840, 2160, 10, {"engines": 8, "wpp": 32}),
    # rate control (hmr_rate_control.c): the QP of a CTU follows the bits of the CTUs entropy coded so far - in the single-thread order, and in the synchronous
    # wavefront (the turnstile serialises the threads' post-decision sections, so every CTU of a step sees the bits as of the end of the step before)
    ("416x240_cbr400_perf1", 416, 240, 8, {"bitrate_mode": 1, "bitrate": 400, "perf": 1}),
    # ... and the bit estimates of RD_FULL, rate control at 2160p, the all-intra RD_FULL encode of BASELINE.json configs[4] as BASELINE.md states it (performance_mode 0):
    # the reference's deterministic single-thread mode itself (ref_lockstep, no turnstile) - on the device one CTU at a time, the pool's raster schedule
    ("416x240_force_intra_rdfull_tr4", 416, 240, 3, {"force_intra": 1, "rd": 1, "intra_tr": 4}),
    ("416x240_rdfull", 416, 240, 4, {"rd": 1}),
    ("328x264_force_intra_rdfull_tr3_perf0", 328, 264, 2, {"force_intra": 1, "rd": 1, "intra_tr": 3, "perf": 0}),
    ("416x240_vbr400", 416, 240, 6, {"bitrate_mode": 2, "bitrate": 400}),
    ("3840x2160_cbr20000_perf1", 3840, 2160, 2, {"bitrate_mode": 1, "bitrate": 20000, "perf": 1}),
    ("3840x2160_force_intra_rdfull_tr4_perf0", 3840, 2160, 1, {"force_intra": 1, "rd": 1, "intra_tr": 4, "perf": 0}),
    # rate control with several engines (the engine turnstile: a frame decides with the rate-control state its engine copied when the frame was fed)
    ("416x240_cbr400_perf1_eng2_wpp_rows", 416, 240, 12, {"bitrate_mode": 1, "bitrate": 400, "perf": 1, "engines": 2, "wpp": 4}),
    ("416x240_vbr400_eng3_wpp_rows", 416, 240, 12, {"bitrate_mode": 2, "bitrate": 400, "engines": 3, "wpp": 4}),
    ("416x240_cbr300_eng2", 416, 240, 10, {"bitrate_mode": 1, "bitrate": 300, "engines": 2}),
    ("832x480_cbr1500_perf1_eng4_wpp_rows", 832, 480, 12, {"bitrate_mode": 1, "bitrate": 1500, "perf": 1, "engines": 4, "wpp": 8}),
    ("416x240_cbr400_perf1_wpp_rows", 416, 240, 8, {"bitrate_mode": 1, "bitrate": 400, "perf": 1, "wpp": 4}),
    ("416x240_vbr400_wpp_rows", 416, 240, 8, {"bitrate_mode": 2, "bitrate": 400, "wpp": 4}),
    ("832x480_cbr1500_perf1_wpp_rows", 832, 480, 6, {"bitrate_mode": 1, "bitrate": 1500, "perf": 1, "wpp": 8}),
    ("416x240_cbr300_nosao_wpp_rows", 416, 240, 6, {"bitrate_mode": 1, "bitrate": 300, "sao": 0, "wpp": 4}),
    ("1920x1080_cbr5000_perf1_wpp_rows", 1920, 1080, 6, {"bitrate_mode": 1, "bitrate": 5000, "perf": 1, "wpp": 17}),
    # a very low QP without WPP: the picture is ONE sub-stream of several hundred bytes per CTU (the device's entropy stage writes it into the whole allocation, not a row's share)
    ("416x240_qp4", 416, 240, 2, {"qp": 4}),
    # performance_mode 3 (PERF_FASTEST_COMPUTATION): inter CUs from depth 2 on, and in the intra walk (I pictures, the CTUs after a scene cut) the variance
    # pre-analysis (analyse_recursive_info_cu, hmr_motion_intra.c:1660) decides which partitions are evaluated as a whole and where the recursion ends
    ("416x240_perf3", 416, 240, 5, {"perf": 3}),
    ("416x240_perf3_wpp_rows", 416, 240, 5, {"perf": 3, "wpp": 4}),
    ("416x240_force_intra_perf3_wpp_rows", 416, 240, 3, {"perf": 3, "force_intra": 1, "wpp": 4}),
    ("416x240_scene_cut_perf3_wpp_rows", 416, 240, 26, {"perf": 3, "cut_at": 23, "wpp": 4}),
    ("832x480_qp26_perf3_rdfull_wpp_rows", 832, 480, 3, {"perf": 3, "qp": 26, "rd": 1, "wpp": 8}),
    ("3840x2160_cbr20000_perf1_wpp32", 3840, 2160, 4, {"bitrate_mode": 1, "bitrate": 20000, "perf": 1, "wpp": 32}),       # BASELINE.json configs[2]: 2160p IPPP, CBR 20000 kbps, performance_mode 1       # BASELINE.json configs[3]: 2160p, n_enc_engines = 8     # the 2160p picture of the metric with the reference's maximum of 32 WPP threads for 34 CTU rows (I + P + P)
]


def run(width, height, frames, keys):
    keys = dict(keys)
    cut_at = keys.pop("cut_at", None)
    clip_seed = keys.pop("clip_seed", None)
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "in.yuv")
        gen_yuv.write_clip(yuv, width, height, frames, seed=clip_seed or 1234, cut_at=cut_at)
        turnstile = int(keys.get("wpp", 1)) > 1 or int(keys.get("engines", 1)) > 1
        cmd = [os.path.join(ROOT, "oracle", "_ref", "ref_ctudump" if turnstile else "ref_lockstep"), yuv, os.path.join(tmp, "out.265"), str(width), str(height), str(frames),
               "recon=" + os.path.join(tmp, "rec.yuv")] + [f"{k}={v}" for k, v in keys.items()]
        subprocess.run(cmd, check=True, timeout=900, stdout=subprocess.DEVNULL, env=dict(os.environ, HOMER_TURNSTILE="1") if turnstile else None)
        stream = open(os.path.join(tmp, "out.265"), "rb").read()
        rec = open(os.path.join(tmp, "rec.yuv"), "rb").read()
    fsz = width * height * 3 // 2
    if cut_at is not None:
        keys["cut_at"] = cut_at
    if clip_seed is not None:
        keys["clip_seed"] = clip_seed
    return {"width": width, "height": height, "frames": frames, "keys": keys, "stream_md5": hashlib.md5(stream).hexdigest(), "stream_bytes": len(stream),
            "nal_sizes": [len(x) for x in stream_diff.split_nals(stream)],
            "recon_md5": [hashlib.md5(rec[f * fsz:(f + 1) * fsz]).hexdigest() for f in range(frames)]}


if __name__ == "__main__":
    # no arguments: every case; with case names: only those, merged into the existing file
    path = os.path.join(HERE, "streams.json")
    only = set(sys.argv[1:])
    out = json.load(open(path)) if only and os.path.exists(path) else {}
    out.update({name: run(w, h, f, keys) for name, w, h, f, keys in CASES if not only or name in only})
    out = {name: out[name] for name, *_ in CASES if name in out}
    json.dump(out, open(os.path.join(HERE, "streams.json"), "w"), indent=1)
    for k, v in out.items():
        print(k, v["stream_md5"], v["stream_bytes"])
