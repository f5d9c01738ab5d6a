"""Aggregates a rocprofv3 kernel trace (…_kernel_trace.csv) by (kernel, grid size): launches, total and average
duration, plus the idle time between consecutive dispatches -- which level of the multigrid cycle the time goes to.
usage: kernel_trace_by_grid.py trace.csv [out.txt]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
agg = defaultdict(lambda: [0, 0])
ev = []
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("femshell::", "")
    grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    agg[(name, grid // max(wg, 1))][0] += 1
    agg[(name, grid // max(wg, 1))][1] += e - s
    ev.append((s, e))
ev.sort()
busy = sum(e - s for s, e in ev)
gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
small = sum(g for g in gaps if 0 < g < 50000)
print("dispatches %d, kernel time %.3f ms, idle between dispatches (gaps < 50 us) %.3f ms in %d gaps, median gap %.2f us"
      % (len(ev), busy / 1e6, small / 1e6, sum(1 for g in gaps if 0 < g < 50000), sorted(gaps)[len(gaps) // 2] / 1e3), file=out)
print("%-34s %8s %8s %10s %9s %6s" % ("kernel", "wgs", "calls", "total ms", "avg us", "%"), file=out)
for (name, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-34s %8d %8d %10.3f %9.2f %6.2f" % (name[:34], g, n, t / 1e6, t / n / 1e3, 100.0 * t / busy), file=out)
