#!/usr/bin/env python3
"""Read what bench.py printed: `{"detail": name, "content": ...}` lines followed by the compact headline as the last line; load() puts them back together
(the headline's keys, with every detailed section in place of its short form).     usage: tools/bench_record.py file > full.json"""
import json
import sys


def load(path):
    head, details = None, {}
    for line in open(path).read().splitlines():
        line = line.strip()
        if not line.startswith("{"):
            continue
        try:
            d = json.loads(line)
        except ValueError:
            continue
        if set(d) == {"detail", "content"}:
            details[d["detail"]] = d["content"]
        else:
            head = d
    if head is None:
        raise ValueError(f"{path}: no bench line")
    full = dict(head)
    full["headline"] = head
    full.update(details)
    return full


if __name__ == "__main__":
    json.dump(load(sys.argv[1]), sys.stdout, indent=1)
    print()
