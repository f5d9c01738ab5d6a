#!/usr/bin/env python3
"""Summarises two rocprofv3 counter passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes) into the per-kernel HBM traffic file bench.py reads:
    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB and summed over the XCDs by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per
128-B request of a wide streaming read, so the read bytes are doubled (guide, HBM section)."""
import csv
import hashlib
import json
import os
import statistics
import sys


def kernel_source_digest():
    """the same digest bench.py computes: which kernel sources these counters belong to"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "fem-shell_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".h", ".cpp")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    rows = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            name = name[:name.index("(")] if "(" in name else name
            if name.startswith("void "):
                name = name[5:]
            rows.setdefault(name, {}).setdefault(r["Dispatch_Id"], 0.0)
            rows[name][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in rows.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    fk, wk = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {}
    for name in fk:
        if not name.startswith("femshell::") or name not in wk:
            continue
        f_kb, w_kb = statistics.median(fk[name]), statistics.median(wk[name])
        res[name] = {
            "FETCH_SIZE_KB_median": f_kb, "launches_FETCH_SIZE": len(fk[name]),
            "WRITE_SIZE_KB_median": w_kb, "launches_WRITE_SIZE": len(wk[name]),
            "read_bytes_raw": f_kb * 1024.0, "read_bytes_x2_gfx950": 2.0 * f_kb * 1024.0, "write_bytes": w_kb * 1024.0,
        }
    res["_meta"] = {"kernel_source_digest": kernel_source_digest(), "command": "bench.py --steps 3 --warmup 1 --profile"}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    res.pop("_meta")
    for k, v in res.items():
        print("%-45s read %8.1f MB  write %8.1f MB  (%d launches)" % (k, v["read_bytes_x2_gfx950"] / 1e6, v["write_bytes"] / 1e6,
                                                                    v["launches_FETCH_SIZE"]))


if __name__ == "__main__":
    main()
