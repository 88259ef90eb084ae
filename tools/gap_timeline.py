#!/usr/bin/env python3
"""GPU box tool: what runs between two launches of k_encode_pool.  Input: the kernel trace csv of `rocprofv3 --kernel-trace` on the bench command.
Prints, for each gap between the end of a pool launch and the start of the next, its length and the kernels inside it (first start, last end, count, summed time)."""
import csv
import sys


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]))
    rows.sort()
    pools = [r for r in rows if r[2].startswith("k_encode_pool")]
    pools = [p for p in pools if p[1] - p[0] > float(sys.argv[2]) * 1e6] if len(sys.argv) > 2 else pools        # (ms: only launches longer than this, e.g. the batch launches)
    out = []
    for a, b in zip(pools, pools[1:]):
        gap0, gap1 = a[1], b[0]
        if gap1 - gap0 > 2e9:
            continue
        inside = [r for r in rows if r[0] >= gap0 - 1 and r[1] <= gap1 + 1 and not r[2].startswith("k_encode_pool")]
        names = {}
        for s, e, n in inside:
            d = names.setdefault(n, [s, e, 0, 0])
            d[0], d[1], d[2], d[3] = min(d[0], s), max(d[1], e), d[2] + 1, d[3] + (e - s)
        out.append((gap1 - gap0, names, gap0))
    for gap, names, g0 in out:
        print(f"gap {gap / 1e6:.1f} ms")
        for n, (s, e, c, tot) in sorted(names.items(), key=lambda kv: kv[1][0]):
            print(f"   {n[:44]:44s} x{c:4d}  from {(s - g0) / 1e6:7.2f} to {(e - g0) / 1e6:7.2f} ms   summed {tot / 1e6:8.2f} ms")


if __name__ == "__main__":
    main()
