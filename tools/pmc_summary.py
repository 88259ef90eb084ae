#!/usr/bin/env python3
"""Per-launch HBM traffic of the bench's kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o f -- python3 bench.py --launch-order out/order.json ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -o w -- python3 bench.py ...
    tools/pmc_summary.py out/f_counter_collection.csv out/w_counter_collection.csv out/order.json > profiles/hbm_traffic.json

Counters are per dispatch, in KiB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE tallies 128-byte requests at 64 bytes,
so the read side is doubled; WRITE_SIZE is taken as reported.  Dispatches are matched to bench groups by launch order.
"""
import csv
import json
import sys


def ours(path):
    rows = [r for r in csv.DictReader(open(path)) if ("hmr_gpu_job" in r["Kernel_Name"] or "(anonymous namespace)::k_" in r["Kernel_Name"])
            and "k_nop" not in r["Kernel_Name"] and "k_valu_probe" not in r["Kernel_Name"]]      # the bench's calibration launches are not frame work
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def main():
    fetch, write, order = ours(sys.argv[1]), ours(sys.argv[2]), json.load(open(sys.argv[3]))
    L = len(order)
    out = {}
    for rows, key in ((fetch, "fetch_kib"), (write, "write_kib")):
        assert len(rows) % L == 0, (len(rows), L)
        for i, r in enumerate(rows):
            g = out.setdefault(order[i % L], {"fetch_kib": 0.0, "write_kib": 0.0, "n": 0, "kernel": r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]})
            g[key] += float(r["Counter_Value"])
            if key == "fetch_kib":
                g["n"] += 1
    for g in out.values():
        n = max(g.pop("n"), 1)
        g["fetch_kib"] = round(g["fetch_kib"] / n, 1)
        g["write_kib"] = round(g["write_kib"] / n, 1)
        g["hbm_bytes"] = int((2 * g["fetch_kib"] + g["write_kib"]) * 1024)
    json.dump({"note": "per launch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count correction)", "groups": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
