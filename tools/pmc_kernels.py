#!/usr/bin/env python3
"""Sum the rocprofv3 --pmc passes written by tools/profile_bench.sh per kernel and derive the shares DESIGN.md quotes.

    tools/pmc_kernels.py gpurun_out r03 "command line" frames > profiles/r03_pmc_kernels.json

Every pass is its own run of the same command (MI355X_MICROARCH.md, HBM / rocprofv3: counter sets that do not fit one pass are collected
separately, never together with a trace domain).  HBM bytes = TCC_EA0_RDREQ / WRREQ requests x 64 B; the write figure is an upper bound
(narrow writes are counted as full requests)."""
import csv
import glob
import json
import os
import sys


def main():
    root, tag = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    frames = int(sys.argv[4]) if len(sys.argv) > 4 else None      # frames the command encodes (all launches of k_encode_pool together)
    kernels = {}
    for path in sorted(glob.glob(os.path.join(root, f"pmc_{tag}_*", "**", "*counter_collection.csv"), recursive=True)):
        launches = {}
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1].strip()
            k = kernels.setdefault(name, {})
            k[row["Counter_Name"]] = k.get(row["Counter_Name"], 0) + int(float(row["Counter_Value"]))
            launches.setdefault(name, set()).add(row["Dispatch_Id"])
        for name, ids in launches.items():
            kernels[name]["launches"] = len(ids)
    for name, k in kernels.items():
        d = {}
        if "SQ_WAVE_CYCLES" in k and k["SQ_WAVE_CYCLES"]:
            d["wait_share_of_wave_cycles"] = round(k.get("SQ_WAIT_ANY", 0) / k["SQ_WAVE_CYCLES"], 3)
            d["issue_share_of_wave_cycles"] = round(k.get("SQ_ACTIVE_INST_ANY", 0) / k["SQ_WAVE_CYCLES"], 3)
        if k.get("SQ_THREAD_CYCLES_VALU") and k.get("SQ_ACTIVE_INST_VALU"):
            # lanes a vector instruction keeps busy: thread-cycles over 64 x the cycles vector instructions were executing (both in quad-cycles)
            d["valu_lane_utilisation"] = round(k["SQ_THREAD_CYCLES_VALU"] / (64.0 * k["SQ_ACTIVE_INST_VALU"]), 4)
        if "TCC_EA0_RDREQ_sum" in k:
            d["hbm_read_bytes_TCC_EA0_RDREQ_x64"] = k["TCC_EA0_RDREQ_sum"] * 64
            d["hbm_write_bytes_TCC_EA0_WRREQ_x64_upper_bound"] = k.get("TCC_EA0_WRREQ_sum", 0) * 64
            # by request width (tools/tcc_calibrate.py: reads 64 / 128 bytes, writes 64 bytes for a full line and 32 for a partial one): the bytes that cross the fabric
            if "TCC_EA0_RDREQ_128B_sum" in k:
                r32, r64, r128 = k.get("TCC_EA0_RDREQ_32B_sum", 0), k.get("TCC_EA0_RDREQ_64B_sum", 0), k["TCC_EA0_RDREQ_128B_sum"]
                d["hbm_read_bytes"] = 32 * r32 + 64 * r64 + 128 * r128 + 64 * max(k["TCC_EA0_RDREQ_sum"] - r32 - r64 - r128, 0)
            if "TCC_EA0_WRREQ_64B_sum" in k:
                d["hbm_write_bytes"] = 64 * k["TCC_EA0_WRREQ_64B_sum"] + 32 * (k.get("TCC_EA0_WRREQ_sum", 0) - k["TCC_EA0_WRREQ_64B_sum"])
                d["write_requests_full_line_share"] = round(k["TCC_EA0_WRREQ_64B_sum"] / max(k.get("TCC_EA0_WRREQ_sum", 0), 1), 3)
            tot = k.get("TCC_HIT_sum", 0) + k.get("TCC_MISS_sum", 0)
            if tot:
                d["l2_hit_rate"] = round(k["TCC_HIT_sum"] / tot, 4)
        if name.startswith("k_encode_ctus") or name.startswith("k_encode_pool"):
            d["note"] = (f"rocprofv3 --pmc passes of `{cmd}`; sums over all launches of the kernel (two wavefronts per workgroup: the row worker and its "
                         "helper, whose polling counts as waiting); SQ cycle counters in quad-cycles")
            if k.get("SQ_WAVE_CYCLES"):
                d["valu_share_of_issue"] = round(k.get("SQ_INSTS_VALU", 0) / max(k.get("SQ_INSTS_VALU", 0) + k.get("SQ_INSTS_SALU", 0), 1), 3)
        k["derived"] = d
    if frames:
        # (`frames` counts every picture the command encodes; those of launches with at most one worker per CU - one picture each here - go through the latency kernel
        # k_encode_pool_lat and are not k_encode_pool's)
        lat = kernels.get("k_encode_pool_lat", {}).get("launches", 0)      # (launches of ONE pass, as every count in this summary)
        kernels["frames_encoded_by_k_encode_pool"] = frames - lat
        if lat:
            kernels["frames_encoded_by_k_encode_pool_lat"] = lat
    try:      # the build the passes ran on (the tree the GPU box was given: HEAD when the working tree is clean)
        import subprocess
        kernels["build_commit"] = subprocess.run(["git", "-C", os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        kernels["build_commit"] = None
    try:      # the digest of the sources in the tree the passes ran on (what the GPU box does have)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from homerhevc_amd.build import source_digest
        kernels["source_digest"] = source_digest()
    except Exception:      # noqa: BLE001
        kernels["source_digest"] = None
    json.dump(kernels, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
