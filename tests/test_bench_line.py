"""bench.py's record (no GPU): the LAST stdout line is a compact JSON object the driver can parse - under 4 KB whatever the sections hold - and tools/bench_record.py puts the
detail lines back together."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench           # noqa: E402
import bench_record    # noqa: E402


def stub_record():
    """The shape of a full N = 1 record, with every free-text field long (round 5's line was 25.8 KB and the driver could not parse it)."""
    long = "x" * 3000
    section = {"value": 100.4, "unit": "frames/s", "ms_per_step": 1912.0, "steps": 4, "warmup": 2, "stream_md5": "0" * 32, "stream_matches_reference": True,
               "frames_checked_against_reference": 6, "clips": {"by_seed": {str(s): {"stream_md5": "0" * 32, "note": long} for s in range(8)}},
               "config": {"sequences_per_gpu": 192, "timed_region": long}, "single_sequence": {"value": 1.84, "ms_per_step": 543.0, "stream_matches_reference": True},
               "cpu_baseline": {"value": 11.0, "cores": 12.2, "sample": long, "one_process": {"value": 0.785}, "throughput": {"note": long}}}
    return {
        "metric": "encoded frames/sec, 1080p & 2160p YUV420 fixed-QP IPPP, 1/2/4/8 MI355X", "value": 414.0123, "unit": "frames/s", "n_gpus": 1, "steps": 20, "warmup": 3,
        "ms_per_step": 618.3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
        "config": {"workload": "cfg2-1080p-encode", "sequences_per_gpu": 256, "width": 1920, "height": 1080, "wfpp_num_threads": 17, "qp": 32, "gop": "IPPP intra_period=100",
                   "stream_matches_reference": True, "frames_checked_against_reference": 23, "timed_region": long, "call": long, "parallelism": long},
        "stream_md5": "0" * 32, "stream_matches_reference": True, "frames_checked_against_reference": 23, "evaluations_on_a_stale_prediction_window": 0,
        "clips": {"by_seed": {str(s): {"stream_md5": "0" * 32, "note": long} for s in range(8)}},
        "schedule": {"ctu_stage_ms_per_frame": [597.7] * 20, "passes_per_frame": [1] * 20},
        "roofline": {"bound": "hbm", "kernel": "k_encode_pool", "achieved": 9.33, "peak": 8000.0, "unit": "GB/s", "frac": 0.00117, "traffic": 383000000000, "traffic_source": long,
                     "traffic_build": "abcdef012345", "this_build": "abcdef012345", "traffic_is_of_this_build": True, "launches": 20, "algorithmic_bytes_per_launch": 5573836800,
                     "algorithmic_bytes_model": "10.5 W H per frame", "ms_per_launch": 597.7, "note": long,
                     "issue_bound": {"wave_instructions_per_frame": 1000000000, "salu_over_valu": 0.91, "valu_issue_frac": 0.2, "wait_share_of_wave_cycles": 0.675,
                                     "valu_lane_utilisation": 0.5, "build": "abcdef012345", "probe": {"rows": [{"note": long}] * 11}, "verdict": long},
                     "subpel_planes": {"achieved": 1980.0, "frac": 0.2475, "ms_per_picture": 0.0664, "kernels": long}},
        "single_sequence": {"value": 4.19, "stream_matches_reference": True, "schedule": {"x": [1.0] * 500}, "note": long,
                            "engines_overlapped": [{"workload": f"cfg2-1080p-encode-engines{e}", "frames_per_s_full_chains": 11.6, "stream_matches_reference": True, "chain": long} for e in (3, 8, 8)]},
        "single_thread_order": dict(section), "at_2160p": dict(section), "cfg3_2160p_cbr": dict(section), "cfg5_2160p_intra_rdfull": dict(section),
        "cpu_baseline": {"value": 60.8, "unit": "frames/s", "cores": 11.1, "kind": "reference", "sample": long, "throughput": {"note": long},
                         "one_process": {"value": 5.49, "frames": 24}, "one_thread_per_ctu_row": {"value": 26.8, "threads": 17, "note": long}},
        "note": long,
    }


def test_last_line_is_compact_and_parses(tmp_path):
    out = stub_record()
    text = io.StringIO()
    with redirect_stdout(text):
        bench.emit(out)
    lines = text.getvalue().strip().splitlines()
    last = lines[-1]
    assert len(last) < 4096
    head = json.loads(last)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in head
    assert head["value"] == out["value"] and head["config"]["workload"] == "cfg2-1080p-encode" and head["config"]["frames_checked"] == 23
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "ms_per_launch", "algorithmic_bytes_per_launch", "traffic_build"):
        assert key in head["roofline"]
    for key in ("value", "unit", "cores", "kind", "sample", "one_process", "rows_threads"):
        assert key in head["cpu_baseline"]
    assert head["single_sequence"]["value"] == 4.19 and head["cfg5_2160p_intra_rdfull"]["alone"] == 1.84
    # every earlier line parses too, and the reader restores the detailed sections
    for line in lines[:-1]:
        assert set(json.loads(line)) == {"detail", "content"}
    path = tmp_path / "record.json"
    path.write_text(text.getvalue())
    full = bench_record.load(str(path))
    assert full["schedule"]["ctu_stage_ms_per_frame"] == [597.7] * 20 and full["headline"]["value"] == out["value"] and full["roofline"]["traffic_source"]


def test_ring_line_is_compact():
    """The N > 1 record (run_engine_ring's keys)."""
    out = {"metric": "m", "value": 1.0, "unit": "frames/s", "n_gpus": 8, "steps": 20, "warmup": 3, "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int16", "data": "synthetic", "config": {"workload": "cfg2-1080p-encode-engines8", "sequences_per_gpu": 160, "sequences": 1280, "num_enc_engines": 8, "width": 1920,
                                                             "height": 1080, "parallelism": "y" * 2000, "stream_matches_reference": True, "access_units_checked_against_reference": 100},
           "stream_matches_reference": True, "access_units_checked_against_reference": 100, "access_units_produced": 100, "access_units_differing": 0,
           "first_differences_on_rank_0": [], "exchange": {"bytes_per_rank_and_step": 1}, "roofline": None, "cpu_baseline": None}
    text = io.StringIO()
    with redirect_stdout(text):
        bench.emit(out)
    head = json.loads(text.getvalue().strip().splitlines()[-1])
    assert head["n_gpus"] == 8 and head["config"]["frames_checked"] == 100 and head["access_units_differing"] == 0 and head["roofline"] is None


def test_describe_keys():
    assert bench.describe_keys({}) == "IPPP QP32"
    assert bench.describe_keys(bench.WORKLOADS["cfg3-2160p-cbr"][2]) == "IPPP CBR 20000 kbps perf=1"
    assert bench.describe_keys(bench.WORKLOADS["cfg5-2160p-intra-rdfull"][2]) == "all-intra QP32 rd=1 intra_tr=4 perf=0"
