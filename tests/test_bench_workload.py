"""Host-side checks of the bench workload definition (no GPU): the recorded call mixes, the clip generator and the way bench.py turns a
call mix into launches."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tools", "legacy"))
import bench_callmix as bench      # noqa: E402
import gen_yuv    # noqa: E402


def test_clip_generator_matches_published_md5():
    """md5 of the generated clips (720p x 8 here; the 1080p / 2160p clips use the same code)."""
    md5 = hashlib.md5()
    for planes in gen_yuv.gen_frames(1280, 720, 8):
        for p in planes:
            md5.update(p.tobytes())
    assert md5.hexdigest().startswith("eba88043")
    # long clips (CPU baseline sample): the texture window wraps instead of running off its margin
    frames = list(gen_yuv.gen_frames(320, 192, 40))
    assert len(frames) == 40 and frames[39][0].shape == (192, 320)


@pytest.mark.parametrize("workload", list(bench.WORKLOADS))
def test_callmix_fixture_is_consistent(workload):
    bench.set_workload(workload)
    with open(os.path.join(ROOT, "tests", "golden", bench.CALLMIX)) as f:
        mix = json.load(f)
    assert (mix["width"], mix["height"]) == (bench.W, bench.H) and mix["qp"] == 32
    n_ctu = ((bench.W + 63) // 64) * ((bench.H + 63) // 64)
    for fr in mix["frames"][1:]:
        c = fr["calls"]
        tot = lambda k: sum(v for key, v in c.items() if key.split(":")[0] == k)   # noqa: E731
        # the per-TU chain: one predict / transform / quant / reconst per TU, inverse path only for coded TUs
        assert tot("transform") == tot("quant") == tot("reconst") and tot("predict") == tot("quant") + tot("inter_tu")
        assert tot("inv_quant") == tot("itransform") <= tot("quant")
        # inter TUs (encode_inter_cu / _chroma): one DCT / quant / reconst each, two SSDs when coded, one otherwise
        assert tot("transform@etu") == tot("quant@etu") == tot("reconst@etu") == tot("inter_tu")
        assert tot("ssd16b@etu") == tot("inter_tu") + tot("inv_quant@etu") and tot("inv_quant@etu") == tot("itransform@etu")
        # luma intra TUs (encode_intra_cu)
        assert tot("fill_reference_samples@itu") == tot("predict@itu") == tot("quant@itu") == tot("ssd16b@itu") == tot("intra_tu")
        assert tot("intra_planar@itu") + tot("intra_angular@itu") == tot("intra_tu")
        # drivers and the calls they make
        assert tot("half_pel_planes") == tot("quarter_pel_planes")
        assert tot("interp_luma@planes") == 16 * tot("half_pel_planes") and tot("sad_direct") == 18 * tot("half_pel_planes")
        assert tot("mc_chroma") == 2 * tot("mc_luma")
        assert tot("fill_reference_samples@search") == tot("intra_search") == tot("intra_planar@search")
        assert tot("sad@search") == tot("intra_planar@search") + tot("intra_angular@search")
        # chroma CU drivers (encode_intra_chroma): ten {reference build, prediction, SAD} per CU in the search, then one TU per component (eight when split)
        n_cu = sum(v for key, v in c.items() if key.startswith("intra_chroma_cu:"))
        n_tu = sum(v * (8 if key.endswith(":1") else 2) for key, v in c.items() if key.startswith("intra_chroma_cu:"))
        assert tot("sad@chroma") == 10 * n_cu and tot("predict@chroma") == tot("quant@chroma") == tot("reconst@chroma") == tot("ssd16b@chroma") == n_tu
        assert tot("fill_reference_samples@chroma") == tot("intra_planar@chroma") + tot("intra_angular@chroma") == 10 * n_cu + n_tu
        assert tot("sao_stats_ctu") == tot("sao_offset_ctu") == n_ctu and tot("deblock_ctu") == tot("pad_ctu") == 2 * n_ctu
    bench.set_workload("cfg2-1080p-P-frame-replay")


def test_fused_and_unfused_replays_account_for_the_same_work():
    bench.set_workload("cfg2-1080p-P-frame-replay")
    calls = bench.load_callmix(2)
    res = {}
    for fused in (True, False):
        groups, _ = bench.build_groups(calls, np.random.default_rng(7), bench.Arena(), fused=fused)
        res[fused] = (sum(g["bytes"] for g in groups), sum(len(g["jobs"]) for g in groups), {g["name"] for g in groups})
    # algorithmic bytes are those of the table calls either way (fused jobs are priced as the calls they stand for)
    assert abs(res[True][0] - res[False][0]) / res[False][0] < 0.02
    assert res[False][1] == sum(v for k, v in calls.items() if k.split(":")[0].split("@")[0] in (
        "sad", "sad_direct", "ssd16b", "predict", "reconst", "copy_16_16", "intra_planar", "intra_angular", "fill_reference_samples", "interp_luma",
        "interp_chroma", "transform", "itransform", "quant", "inv_quant") and not (k.startswith("copy_16_16") and int(k.split(":")[2]) > bench.W))
    # in this P frame every table-level TU chain and every loose reference build / prediction comes from the chroma CU drivers, which go out as search + TU launches
    assert {"inter_tu", "intra_tu", "me_subpel", "mc_luma", "mc_chroma", "intra_search", "chroma_search", "chroma_tus8s0", "chroma_tus4s0"} <= res[True][2]
    assert not ({"tu_chain", "intra_refs", "intra_pred"} & res[True][2])
    groups, _ = bench.build_groups(calls, np.random.default_rng(7), bench.Arena(), fused=True, chroma_driver=False)
    assert {"tu_chain", "intra_refs", "intra_pred"} <= {g["name"] for g in groups} and not any(g["name"].startswith("chroma_") for g in groups)
    assert abs(sum(g["bytes"] for g in groups) - res[True][0]) / res[True][0] < 0.005
    assert not ({"tu_chain", "inter_tu", "intra_tu", "me_subpel", "mc_luma", "intra_search"} & res[False][2])
    # --cu-driver: in this P frame every intra CU goes through the one-level tree, so all its searches and TUs become luma CU driver chains
    groups, _ = bench.build_groups(calls, np.random.default_rng(7), bench.Arena(), fused=True, cu_driver=True)
    assert {"cu_search", "cu_tu0", "cu_children", "cu_decide"} <= {g["name"] for g in groups} and not ({"intra_search", "intra_tu"} & {g["name"] for g in groups})
    assert abs(sum(g["bytes"] for g in groups) - res[True][0]) / res[True][0] < 0.001
    by = {(g["name"], g["size"]): len(g["jobs"]) for g in groups}
    n_cu = {int(k.split(":")[1]): v for k, v in calls.items() if k.startswith("intra_cu:")}
    tot = lambda kind, n: sum(v for k, v in calls.items() if k.split(":")[0] == kind and int(k.split(":")[1]) == n)   # noqa: E731
    for n, m in n_cu.items():
        assert by[("cu_search", n)] == by[("cu_tu0", n)] == by[("cu_decide", n)] == m and by[("cu_children", n // 2)] == 4 * m
    for n in (4, 8, 16, 32, 64):      # chains + what is left of the plain batches = the recorded calls
        assert by.get(("intra_search", n), 0) + n_cu.get(n, 0) == tot("intra_search", n)
        assert by.get(("intra_tu", n), 0) + (n_cu.get(n, 0) if n <= 32 else 0) + 4 * n_cu.get(2 * n, 0) == tot("intra_tu", n)
    # every job's operands stay inside the arena
    for cu in (False, True):
        arena = bench.Arena()
        groups, _ = bench.build_groups(calls, np.random.default_rng(7), arena, fused=True, cu_driver=cu)
        for g in groups:
            for field in g["jobs"].dtype.names:
                if field.endswith("_off"):
                    assert int(g["jobs"][field].max()) < arena.size, (g["name"], field)
