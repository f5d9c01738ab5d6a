#!/bin/bash
# per-kernel times of the dense inverse under rocprofv3 --kernel-trace --stats, one configuration per argument
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r05_dense_trace.txt
: > $O
n=0
for v in "$@"; do
  n=$((n+1))
  echo "== $v" >> $O
  rm -rf gpurun_out/dtrace_$n
  # (the environment goes in through the shell, the program after -- is python itself)
  export $v
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dtrace_$n -o t -- python3 tools/lab/dense_probe.py 105 2 > gpurun_out/dtrace_$n.log 2>&1 || { grep -v "^    @" gpurun_out/dtrace_$n.log | tail -5; exit 1; }
  for k in $v; do unset ${k%%=*}; done
  grep "dense inverse" gpurun_out/dtrace_$n.log >> $O
  python3 - gpurun_out/dtrace_$n >> $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dense" in r["Name"] and "gemv" not in r["Name"]:
            name = r["Name"].replace("(anonymous namespace)::", "").replace("femshell::", "").split("(")[0].replace("void ", "")
            print("   %-34s calls %4s  avg %8.1f us  total %8.3f ms" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -rf gpurun_out/dtrace_$n
done
cat $O
