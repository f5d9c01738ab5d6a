#!/bin/bash
# kernel trace of one multigrid setup + solve of the 4M-triangle panel: which kernels the setup spends its GPU time in
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r06_setup_trace
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o trace -- python3 tools/amg_probe.py panel 1414 > $out/probe.log 2>&1
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/kernel_trace.csv
python3 - <<'PY'
import csv, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_setup_trace"
rows = list(csv.DictReader(open(out + "/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the setup: from the first k_patch_sigma / k_histogram to the first k_pcg_init
names = [r["Kernel_Name"] for r in rows]
def first(sub, start=0):
    for i in range(start, len(names)):
        if sub in names[i]:
            return i
    return -1
a = first("k_patch_sigma")
b = first("k_pcg_init", a)
t0 = int(rows[a]["Start_Timestamp"])
with open(out + "/setup_timeline.txt", "w") as f:
    agg = {}
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("femshell::", "").replace("void ", "").split("(")[0][:70]
        k = agg.setdefault(nm, [0, 0.0, s])
        k[0] += 1
        k[1] += (e - s) / 1e6
    f.write("setup window %.2f ms, %d dispatches\n" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e6, b - a))
    for nm, (cnt, ms, s) in sorted(agg.items(), key=lambda kv: kv[1][2]):
        f.write("%8.2f ms first at  %5d x %8.3f ms  %s\n" % ((s - t0) / 1e6, cnt, ms, nm))
print(open(out + "/setup_timeline.txt").read())
PY
rm -rf $out/prof $out/kernel_trace.csv
