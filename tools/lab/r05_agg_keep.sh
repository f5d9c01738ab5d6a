#!/bin/bash
# the aggregation on a graph filtered to the `keep` closest neighbours per node (FEMSHELL_AMG_AGG_KEEP) across mesh sizes
mkdir -p gpurun_out
O=gpurun_out/r05_agg_keep.txt
: > $O
for k in 0 12 10; do
  echo "== FEMSHELL_AMG_AGG_KEEP=$k" >> $O
  FEMSHELL_AMG_AGG_KEEP=$k timeout -k 10 600 python tools/lab/thickness_probe.py 1300 3 1414 3 1600 3 2200 3 3200 3 >> $O 2>&1 || exit 1
  FEMSHELL_AMG_AGG_KEEP=$k timeout -k 10 300 python tools/lab/levels_probe.py panel 1414 1400 >> $O 2>&1 || exit 1
  FEMSHELL_AMG_AGG_KEEP=$k timeout -k 10 300 python tools/amg_probe.py roof 354 2>&1 | grep "second solve" | cut -c1-60 >> $O
done
cat $O
