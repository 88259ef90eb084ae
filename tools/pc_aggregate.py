#!/usr/bin/env python3
"""Aggregate a rocprofv3 pc_sampling_host_trap csv per source line (Instruction_Comment) and per instruction mnemonic."""
import collections
import csv
import json
import sys

lines, insts, by_line_inst = collections.Counter(), collections.Counter(), collections.Counter()
total = 0
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        total += 1
        where = row.get("Instruction_Comment", "") or "?"
        ins = (row.get("Instruction", "") or "?").split(" ")[0]
        lines[where] += 1
        insts[ins] += 1
        by_line_inst[(where, ins)] += 1
json.dump({"samples": total, "lines": lines.most_common(1500), "instructions": insts.most_common(60),
           "line_instruction": [[k[0], k[1], v] for k, v in by_line_inst.most_common(400)]}, sys.stdout)
