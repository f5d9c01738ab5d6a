#!/usr/bin/env python3
"""HBM traffic of every (kernel, grid size) of a run, from two rocprofv3 counter passes collected separately
(--pmc FETCH_SIZE, --pmc WRITE_SIZE; MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request on gfx950, so the
read bytes are doubled), joined with the durations of a --kernel-trace pass when one is given:
    pmc_by_grid.py <fetch_counter_collection.csv> <write_counter_collection.csv> [kernel_trace.csv] > out.txt
The grid size tells the level of the multigrid hierarchy a launch worked on (tools/kernel_trace_by_grid.py)."""
import csv
import re
import statistics
import sys
from collections import defaultdict


def clean(name):
    return re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "").replace("femshell::", "")


def counters(path, counter):
    per = defaultdict(lambda: defaultdict(float))
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            wg = max(int(r.get("Workgroup_Size", 1) or 1), 1)
            key = (clean(r["Kernel_Name"]), int(r.get("Grid_Size", 0) or 0) // wg)
            per[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in per.items()}


def durations(path):
    per = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
            wg = max(int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1), 1)
            per[(clean(r["Kernel_Name"]), grid // wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return per


def main():
    fk, wk = counters(sys.argv[1], "FETCH_SIZE"), counters(sys.argv[2], "WRITE_SIZE")
    dur = durations(sys.argv[3]) if len(sys.argv) > 3 else {}
    print("%-34s %8s %7s %10s %10s %10s %9s %8s" % ("kernel", "wgs", "calls", "read MB", "write MB", "total MB", "avg us", "TB/s"))
    rows = []
    for key in fk:
        if key not in wk:
            continue
        rd = 2.0 * statistics.median(fk[key]) * 1024.0
        wr = statistics.median(wk[key]) * 1024.0
        us = statistics.mean(dur[key]) / 1e3 if key in dur else 0.0
        rows.append((key, len(fk[key]), rd, wr, us))
    for (name, g), n, rd, wr, us in sorted(rows, key=lambda r: -(r[2] + r[3]) * r[1]):
        print("%-34s %8d %7d %10.2f %10.2f %10.2f %9.2f %8.2f" % (name[:34], g, n, rd / 1e6, wr / 1e6, (rd + wr) / 1e6, us,
                                                                 (rd + wr) / us / 1e6 if us > 0 else 0.0))


if __name__ == "__main__":
    main()
