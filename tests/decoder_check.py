"""The decoder-side check: a stream is decoded by oracle/hevcdec - a small HEVC decoder written from the standard, test infrastructure, no code shared with
the encoder - and the decoded pictures are compared with the reconstruction the compiled reference encoder dumped when the fixture was minted (`recon_md5` in
tests/golden/streams.json).  While decoding, hevcdec checks what a stream has to get right beyond its syntax: every CABAC sub-stream ends on a terminating bin, the stop bit and
zero bits to the byte boundary, and the next one starts where the slice header's entry point says.

Two sets of fixtures are EXPECTED to differ, and the tests hold them to exactly that (DESIGN.md section 6 - defects of the reference encoder that the product reproduces because
its bar is byte identity with the reference):
  R1  under rate control the reference deblocks coding units whose QP the stream does not carry (no coded cu_qp_delta yet in their CTU, or none at all) with its rate-control QP,
      or with the predicted one, depending on how far its lagged filter / entropy pipeline has got; a decoder always uses the predicted QP (H.265 8.6.1).  The decoded pictures
      then differ from the encoder's reconstruction by a few samples beside block edges - and from there on the encoder predicts from pictures no decoder has.  Evidence that this
      and nothing else is the difference: every fixed-QP fixture decodes identically (incl. all five BASELINE.json configurations at full size, tools/decoder_sweep.py); in the
      first picture that differs it is 5 ... 71 luma samples, by 4 at most (416x240 ... 3840x2160); and for four of the six fixtures `--ref-deblock-qp` (every CU of a CTU with a coded delta
      deblocked with the CTU's QP) makes the pictures identical.
  R2  quirk Q12: a merge candidate far outside the picture is predicted by the reference from a stale window (the library counts these: hmr_gpu_enc_stale_predictions)."""
import hashlib
import os
import subprocess
import tempfile

import libs

HEVCDEC = os.path.join(libs.ORACLE_DIR, "hevcdec")
R1_DEBLOCK_QP = {"416x240_cbr400_perf1_eng2_wpp_rows", "416x240_cbr300_eng2", "832x480_cbr1500_perf1_eng4_wpp_rows", "832x480_cbr1500_perf1_wpp_rows",
                 "1920x1080_cbr5000_perf1_wpp_rows", "3840x2160_cbr20000_perf1_wpp32"}
R2_STALE_WINDOW = {"400x104_qp22_perf0_nosao_wpp_rows_clip657909"}


def build():
    subprocess.check_call(["make", "-s", "-C", libs.ORACLE_DIR, HEVCDEC])


def decode(stream, width, height, ref_deblock_qp=False, keep=False):
    """-> (md5 of every decoded picture, the decoder's summary as a dict[, the pictures]); raises on any violation the decoder finds"""
    build()
    with tempfile.TemporaryDirectory() as tmp:
        src, out = os.path.join(tmp, "in.265"), os.path.join(tmp, "out.yuv")
        open(src, "wb").write(stream)
        r = subprocess.run([HEVCDEC, src, out] + (["--ref-deblock-qp"] if ref_deblock_qp else []), capture_output=True, text=True, timeout=600)
        if r.returncode:
            raise AssertionError(f"hevcdec exit code {r.returncode}: {r.stderr.strip()}")
        data = open(out, "rb").read()
    info = dict(kv.split("=") for kv in r.stdout.split()[1:])
    fsz = width * height * 3 // 2
    assert len(data) % fsz == 0
    out = [hashlib.md5(data[i:i + fsz]).hexdigest() for i in range(0, len(data), fsz)], {k: int(v) for k, v in info.items()}
    return out + (data,) if keep else out


def check(stream, gold, case, recon_frames=None):
    """the stream of fixture `case` (already known to equal the reference's bytes) against the reference's own reconstruction (md5s in the fixture; `recon_frames`: the same
    pictures as bytes, from the encoder under test, for the two sets that are expected to differ)"""
    import numpy as np
    w, h, frames = gold["width"], gold["height"], gold["frames"]
    md5s, info, data = decode(stream, w, h, keep=True)
    rows = (h + 63) // 64
    assert info["pictures"] == frames and info["width"] == w and info["height"] == h
    assert info["substreams"] == frames * rows and info["entry_points"] == frames * (rows - 1)      # (the reference always writes one sub-stream per CTU row)
    same = [a == b for a, b in zip(md5s, gold["recon_md5"])]
    if case in R2_STALE_WINDOW:
        assert same[0] and not all(same)
    elif case in R1_DEBLOCK_QP:
        assert not all(same), "listed as drifting under rate control, but the decoder agrees with the reference's reconstruction"
        k = same.index(False)
        assert k >= 1      # (the first picture is an I picture at the slice QP everywhere)
        if recon_frames is not None:
            fsz = w * h * 3 // 2
            a = np.frombuffer(recon_frames[k], dtype=np.uint8)[:w * h].astype(int)
            b = np.frombuffer(data[k * fsz:k * fsz + w * h], dtype=np.uint8).astype(int)
            assert hashlib.md5(recon_frames[k]).hexdigest() == gold["recon_md5"][k]
            d = np.abs(a - b)
            assert 0 < int((d > 0).sum()) <= 200 and int(d.max()) <= 8, (int((d > 0).sum()), int(d.max()))      # (measured: 5 ... 71 samples, by 4 at most)
    else:
        assert all(same), same
