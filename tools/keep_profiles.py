#!/usr/bin/env python3
"""Build container: copy the summaries a profiling call left in gpurun_out/ into profiles/ (tracked) and stamp them with the commit they were measured on - the GPU box
gets a snapshot of the tree without .git, so the tools there cannot know it.     usage: tools/keep_profiles.py r05 [commit [name-part ...]]
(name parts: only the files whose name contains one of them - a call that re-ran the bench alone must not re-stamp the counter passes of an earlier build)"""
import json
import os
import shutil
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_record

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
only = sys.argv[3:]
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
kept = []
for name in sorted(os.listdir(src)):
    if not name.startswith(tag + "_") or not name.endswith((".json", ".csv", ".md")) or (only and not any(o in name for o in only)):
        continue
    path = os.path.join(src, name)
    if name.endswith(".json"):
        try:
            text = open(path).read().strip()
            try:
                d = json.loads(text)
            except ValueError:
                d = bench_record.load(path)      # (a bench record: detail lines, then the headline)
        except (ValueError, IndexError):
            continue
        if isinstance(d, dict):
            d["build_commit"] = commit
        json.dump(d, open(os.path.join(dst, name), "w"), indent=1)
    else:
        shutil.copy(path, os.path.join(dst, name))
    kept.append(name)
print("\n".join(kept))
