#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as a per-kernel table (markdown/CSV-ish text).

usage: tools/rocprof_summary.py gpurun_out/prof/r01_results.db [steps] > profiles/r01_kernel_stats.txt
`steps` (timed + warmup launches of the bench) turns totals into per-frame figures.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
    print(f"# total kernel time {total / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches" + (f"; {total / 1e3 / steps:.3f} ms per frame over {steps} frames" if steps else ""))
    print("calls,total_us,avg_us,pct,name")
    for name, calls, tot, avg, pct in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0]
        print(f"{calls},{tot:.1f},{avg:.2f},{pct:.2f},{short}")


if __name__ == "__main__":
    main()
