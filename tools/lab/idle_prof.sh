#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/idle_prof
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/idle_prof -o f -- python3 tools/lab/idle_probe.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/idle_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_assemble" in r["Kernel_Name"] or "k_rhs" in r["Kernel_Name"] or "k_item" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[28:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-18s gap before %9.1f us   duration %8.1f us" % (r["Kernel_Name"].split("(")[0][-18:], (s - prev_end) / 1e3 if prev_end else 0.0, (e - s) / 1e3))
    prev_end = e
PY
rm -rf gpurun_out/idle_prof
