#!/usr/bin/env python3
"""Timeline of ONE replayed frame from a rocprofv3 --kernel-trace CSV of `bench.py` (graph mode): when every kernel of the frame started and ended relative
to the first, how many kernels were in flight, and how much of the frame had fewer than two running.
usage: python tools/timeline.py kernel_trace.csv [launches_per_frame] > timeline.md"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
per = int(sys.argv[2]) if len(sys.argv) > 2 else 36
# frames are separated by idle gaps (host replay): split where the gap to every earlier kernel's end exceeds 20 us
frames, cur, cur_end = [], [], 0
for s, e, k in rows:
    if cur and s - cur_end > 20000:
        frames.append(cur); cur = []
    cur.append((s, e, k)); cur_end = max(cur_end, e)
frames.append(cur)
cands = [f for f in frames if abs(len(f) - per) <= 12 and not any("copy" in k.lower() and "Dtod" in k for _, _, k in f)]
f = cands[len(cands) // 2] if cands else max(frames, key=len)
t0 = min(s for s, _, _ in f); t1 = max(e for _, e, _ in f)
print(f"frame of {len(f)} kernels, {(t1 - t0) / 1000:.1f} us from first start to last end; sum of durations {sum(e - s for s, e, _ in f) / 1000:.1f} us\n")
print("| start us | end us | dur us | kernel |\n|---|---|---|---|")
for s, e, k in f:
    name = re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
    print(f"| {(s - t0) / 1000:.1f} | {(e - t0) / 1000:.1f} | {(e - s) / 1000:.1f} | {name} |")
ev = sorted([(s, 1) for s, _, _ in f] + [(e, -1) for _, e, _ in f])
hist, n, last = {}, 0, t0
for t, d in ev:
    hist[n] = hist.get(n, 0) + (t - last); last = t; n += d
print("\n| kernels in flight | us | share |\n|---|---|---|")
for k in sorted(hist):
    print(f"| {k} | {hist[k] / 1000:.1f} | {hist[k] / (t1 - t0):.2f} |")
