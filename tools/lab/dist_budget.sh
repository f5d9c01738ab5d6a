#!/bin/bash
# tools/lab/dist_budget_probe.py under rocprofv3, by level, for N = 8, 4, 2   (gpurun_out/r05_dist_budget_N*.txt)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out
for N in 8 4 2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/db_$N -o a -- python3 tools/lab/dist_budget_probe.py $N > $out/r05_dist_budget_N$N.json 2> $out/db_$N.err
  python3 tools/kernel_trace_by_grid.py $(find $out/db_$N -name "*kernel_trace.csv" | head -1) $out/r05_dist_budget_N$N.txt
  rm -rf $out/db_$N
  tail -1 $out/r05_dist_budget_N$N.json | cut -c1-300
  head -24 $out/r05_dist_budget_N$N.txt
done
