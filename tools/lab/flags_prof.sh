#!/bin/bash
# duration of k_item_flags per call under rocprofv3 (lab): tools/lab/flags_prof.sh
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/flags_prof
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/flags_prof -o f -- python3 tools/lab/flags_probe.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/flags_prof/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_item_flags" in r["Kernel_Name"]:
        print("k_item_flags %.1f us grid %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X")))
PY
rm -rf gpurun_out/flags_prof
