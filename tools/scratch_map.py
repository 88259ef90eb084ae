#!/usr/bin/env python3
"""Where the encoder kernel touches private memory (scratch): compiles homerhevc_amd/csrc/k_encode.hip for gfx950 (device only, line tables) and lists

  * every function's stack frame (`-Rpass-analysis=stack-frame-layout`): spill slots and variables that were not promoted to registers - a local array indexed at run time,
    a struct handed on by address, a by-value copy of a descriptor;
  * the scratch_load / scratch_store instructions by function, loop depth and source line (spills apart from variables).

Private memory is ordinary memory: every access is a trip through L1 / L2, a store is written back at the next release fence of any worker of the XCD, and a function's
callee-saved registers are saved there at every call.  Round 5 found 17 % of the kernel's store instructions this way (the 32 x 32 inverse transform's accumulators, the motion
search's candidate arrays, the merge evaluation's call frame, the coder's views): 2080 -> 584 bytes per lane, 53 -> 0 spilled vector registers.

usage: tools/scratch_map.py [extra compiler flags ...]       (runs here: no GPU needed; about two minutes)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "homerhevc_amd", "csrc", "k_encode.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip", "--cuda-device-only"]


def main():
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-gline-tables-only", "-Rpass-analysis=(stack-frame-layout|kernel-resource-usage)", "-S", "-o", asm, SRC],
                           capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        remarks = r.stderr
        lines = open(asm).read().split("\n")
    print("== kernels: private memory per lane, spills")
    for blk in remarks.split("Function Name: ")[1:]:
        name = blk.split()[0]
        if "k_encode" in name or "k_post" in name:
            f = dict(re.findall(r"(ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill): (\d+)", blk))
            print(f"  {name[:60]:60s} {f}")
    print("== stack frames (functions that have one)")
    for blk in remarks.split("Function: ")[1:]:
        name = blk.split("\n")[0].split(" [")[0]
        spills = len(re.findall(r"Type: Spill", blk))
        var = [int(x) for x in re.findall(r"Type: Variable, Align: \d+, Size: (\d+)", blk)]
        if spills or any(v > 4 for v in var):
            print(f"  {name[:72]:72s} spill slots {spills:3d}  variables {var}")
    files, func, loc, depth = {}, None, None, 0
    cnt = collections.Counter()
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func, depth = m.group(1), 0
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            loc = f"{files.get(int(m.group(1)), m.group(1))}:{m.group(2)}"
            continue
        if re.match(r"^\.LBB", l) or l.startswith("; %bb."):
            mm = re.search(r"Depth=(\d+)", l)
            depth = int(mm.group(1)) if mm else 0
        if "scratch_store" in l or "scratch_load" in l:
            kind = "spill" if ("Spill" in l or "Reload" in l) else "variable"
            cnt[(func[:48], "store" if "store" in l else "load", kind, depth, loc)] += 1
    print("== scratch instructions: count, function, kind, loop depth, source line")
    for k, v in sorted(cnt.items(), key=lambda x: (-x[1], x[0]))[:60]:
        print(f"  {v:4d}  {k[0]:48s} {k[1]:5s} {k[2]:8s} depth {k[3]}  {k[4]}")
    print(f"  total: {sum(v for k, v in cnt.items() if k[1] == 'store')} stores, {sum(v for k, v in cnt.items() if k[1] == 'load')} loads (static)")


if __name__ == "__main__":
    main()
