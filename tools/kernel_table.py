#!/usr/bin/env python3
"""Per-kernel table of one bench record: time per launch, algorithmic rate, measured HBM traffic, VALU issue share.
usage: tools/kernel_table.py profiles/r01_bench.json profiles/hbm_traffic.json profiles/sq_summary.csv > profiles/r01_kernel_table.md"""
import csv
import json
import sys


def main():
    bench = json.load(open(sys.argv[1]))
    traffic = json.load(open(sys.argv[2]))["groups"]
    sq = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[3]))}
    ms, gbs = bench["kernels_ms"], bench["kernels_gbs"]
    print(f"Per-launch numbers of `{bench['config']['workload']}` ({bench['value']} frames/s, {bench['ms_per_step']} ms per frame; sum of launches "
          f"{sum(ms.values()):.3f} ms, overlapped on {bench['config'].get('graph_branches', 1)} graph branches).\n")
    print("| launch | ms | algorithmic GB/s | HBM MB (counters) | HBM GB/s | VALU Minst | share of VALU issue peak |")
    print("|---|---|---|---|---|---|---|")
    for k, t in ms.items():
        tr = traffic.get(k, {})
        hb = tr.get("hbm_bytes")
        kern = tr.get("kernel", "").replace("void ", "")
        r = sq.get(kern)
        # the counter file averages over all launches of a kernel template; only unambiguous when one launch per frame uses it
        insts = float(r["SQ_INSTS_VALU"]) if r and r.get("SQ_INSTS_VALU") and int(r["dispatches"]) <= 6 else None
        floor_ms = insts * 4 / 1024 / 2.4e9 * 1e3 if insts else None
        print(f"| {k} | {t:.4f} | {gbs.get(k, 0):.0f} | {hb / 1e6:.1f} | {hb / 1e6 / t:.0f} |" if hb else f"| {k} | {t:.4f} | {gbs.get(k, 0):.0f} | | |", end="")
        print(f" {insts / 1e6:.2f} | {floor_ms / t:.2f} |" if insts else " | |")


main()
