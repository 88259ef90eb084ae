#!/usr/bin/env python3
"""Join the known byte counts of tools/ubench/tcc_probe with the TCC_EA0 request counters of its rocprofv3 --pmc run.

    tools/tcc_calibrate.py probe_stdout.txt counter_collection.csv [more passes ...] > profiles/r03_tcc_calibration.json

What it shows on gfx950 (profiles/r03_tcc_calibration.json): a read request is 64 or 128 bytes (TCC_EA0_RDREQ_64B / _128B; a wide coalesced read is all
128-byte requests, which is the guide's "FETCH_SIZE is half"), a write request is 64 bytes (TCC_EA0_WRREQ_64B: a full line) or 32 bytes (the rest: a
partial line, however few of its bytes are written) - so  read bytes = 64 RD64 + 128 RD128 + 32 RD32,  write bytes = 64 WR64 + 32 (WR - WR64)  are the bytes
that cross the fabric, and the useful share of them is the pattern's: 16-byte rows of an 8 x 8 block of int16 use a quarter / half of their requests' bytes.
"""
import csv
import json
import sys


def main():
    known = {}
    for line in open(sys.argv[1]):
        parts = line.split()
        if len(parts) == 2:
            known[parts[0]] = float(parts[1])
    counts = {}
    for path in sys.argv[2:]:
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].strip()
            counts.setdefault(name, {})
            counts[name][row["Counter_Name"]] = counts[name].get(row["Counter_Name"], 0) + float(row["Counter_Value"])
    out = {}
    for name, nbytes in known.items():
        c = counts.get(name, {})
        rd, wr, wr64 = c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_WRREQ_sum", 0), c.get("TCC_EA0_WRREQ_64B_sum")
        e = {"known_bytes": int(nbytes), "TCC_EA0_RDREQ": int(rd), "TCC_EA0_WRREQ": int(wr)}
        if wr64 is not None:
            e["TCC_EA0_WRREQ_64B"] = int(wr64)
        req = wr if name.startswith("w_") else rd
        if req:
            e["bytes_per_request"] = round(nbytes / req, 2)
            e["counter_x64_over_known"] = round(req * 64 / nbytes, 3)
        if name.startswith("w_") and wr64 is not None and wr:
            e["fabric_bytes_64_WR64_plus_32_rest"] = int((wr - wr64) * 32 + wr64 * 64)
            e["fabric_bytes_over_known"] = round(((wr - wr64) * 32 + wr64 * 64) / nbytes, 3)
        if name.startswith("r_") and "TCC_EA0_RDREQ_128B_sum" in c:
            r32, r64, r128 = c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0)
            e.update({"TCC_EA0_RDREQ_32B": int(r32), "TCC_EA0_RDREQ_64B": int(r64), "TCC_EA0_RDREQ_128B": int(r128)})
            fab = 32 * r32 + 64 * r64 + 128 * r128 + 64 * max(rd - r32 - r64 - r128, 0)
            e["fabric_bytes_by_request_width"] = int(fab)
            e["fabric_bytes_over_known"] = round(fab / nbytes, 3)
        out[name] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
