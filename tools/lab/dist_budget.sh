#!/bin/bash
# tools/lab/dist_budget_probe.py, the per-GPU strip of an N-rank run on a one-rank RCCL communicator: once without the profiler (the
# figure that counts: <tag>_dist_budget_N<N>_unprofiled.json) and once under rocprofv3 --kernel-trace for the kernels by level
# (<tag>_dist_budget_N<N>.json / .txt; a quarter of an iteration of 150 launches is the tracer there).   usage: dist_budget.sh [tag] [N ...]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out
tag=${1:-r06}
shift
Ns=${@:-8 4 2}
for N in $Ns; do
  timeout 300 python3 tools/lab/dist_budget_probe.py $N > $out/${tag}_dist_budget_N${N}_unprofiled.json 2> $out/db_$N.err
  timeout 300 python3 tools/lab/dist_budget_probe.py $N >> $out/${tag}_dist_budget_N${N}_unprofiled.json 2>> $out/db_$N.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/db_$N -o a -- python3 tools/lab/dist_budget_probe.py $N > $out/${tag}_dist_budget_N$N.json 2>> $out/db_$N.err
  python3 tools/kernel_trace_by_grid.py $(find $out/db_$N -name "*kernel_trace.csv" | head -1) $out/${tag}_dist_budget_N$N.txt
  rm -rf $out/db_$N
  grep -h '"ms_per_iteration"' $out/${tag}_dist_budget_N${N}_unprofiled.json $out/${tag}_dist_budget_N$N.json | sed -E 's/.*"iterations": ([0-9]+).*"ms_per_iteration": ([0-9.]+).*/N '$N': \1 iterations, \2 ms per iteration/'
  head -40 $out/${tag}_dist_budget_N$N.txt
done
