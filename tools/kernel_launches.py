#!/usr/bin/env python3
"""Per-launch durations of k_encode_pool (the CTU stage) from a rocprofv3 --kernel-trace csv beside the HIP-event times bench.py measured in the same run.

    tools/kernel_launches.py gpurun_out/prof_r02/bench_kernel_trace.csv gpurun_out/r02_bench_under_rocprof.json > profiles/r02_k_encode_ctus_launches.json
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_record

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_encode_pool" in r["Kernel_Name"]]
# the bench's headline is the batch of sequences (a launch of as many workgroups as the GPU has CUs); the single-sequence run beside it (17 workgroups) is not listed here
gkey = next((k for k in ("Grid_Size", "Grid_Size_X", "Workgroup_Count") if rows and k in rows[0]), None)
if gkey:
    big = max(int(r[gkey]) for r in rows)
    rows = [r for r in rows if int(r[gkey]) == big]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ms = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2) for r in rows]
b = bench_record.load(sys.argv[2])
timed = b["schedule"]["ctu_stage_ms_per_frame"]
json.dump({"rocprofv3_kernel_trace_ms_per_launch": ms, "warmup_launches": b["warmup"], "timed_launches_rocprofv3_ms": ms[b["warmup"]:],
           "timed_launches_hip_events_ms": timed, "mean_rocprofv3_ms": round(sum(ms[b["warmup"]:]) / max(len(ms[b["warmup"]:]), 1), 2),
           "mean_hip_events_ms": round(sum(timed) / max(len(timed), 1), 2),
           "note": "the kernel statistics csv averages ALL launches of the run (the I frame and the warm-up P frames included); the roofline object of bench.py uses the timed launches"},
          sys.stdout, indent=1)
print()
