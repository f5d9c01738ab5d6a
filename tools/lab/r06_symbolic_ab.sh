#!/bin/bash
# Round 6, VERDICT r5 item 3: the patterns of the coarsening steps in HBM (default) against the host's lists
# (FEMSHELL_AMG_SYMBOLIC=host), 4M-triangle panel and cylinder: laps of the setup, iterations, time.  One gpurun call.
out=gpurun_out/r06_symbolic
mkdir -p $out
[ -x tools/lab/pagefault_probe ] || g++ -O2 -pthread tools/lab/pagefault_probe.cpp -o tools/lab/pagefault_probe
tools/lab/pagefault_probe 16 > $out/pagefault.txt 2>&1
tools/lab/pagefault_probe 1 >> $out/pagefault.txt 2>&1
cat /sys/kernel/mm/transparent_hugepage/enabled >> $out/pagefault.txt 2>&1
for where in device host device host; do
  for mesh in panel cylinder; do
    echo "== $mesh 1414 FEMSHELL_AMG_SYMBOLIC=$where" >> $out/laps.txt
    FEMSHELL_AMG_SYMBOLIC=$where FEMSHELL_AMG_VERBOSE=1 python tools/amg_probe.py $mesh 1414 >> $out/laps.txt 2>&1 || exit 1
  done
done
grep -E "^==|pc_setup_seconds" $out/laps.txt | sed -E 's/.*"iterations": ([0-9]+).*"solve_seconds": ([0-9.]+).*"pc_setup_seconds": ([0-9.]+).*/   iterations \1 solve \2 s setup \3 s/' > $out/summary.txt
cat $out/summary.txt
