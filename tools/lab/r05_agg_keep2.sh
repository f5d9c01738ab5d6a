#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_agg_keep2.txt
: > $O
for k in 0 12; do
  echo "== FEMSHELL_AMG_AGG_KEEP=$k" >> $O
  FEMSHELL_AMG_AGG_KEEP=$k timeout -k 10 900 python tools/lab/thickness_probe.py 700 3 900 3 1200 3 1500 3 1800 3 2000 3 2400 3 2828 3 4000 3 >> $O 2>&1 || exit 1
  FEMSHELL_AMG_AGG_KEEP=$k timeout -k 10 300 python tools/lab/levels_probe.py panel 1000 1400 >> $O 2>&1 || exit 1
  FEMSHELL_AMG_AGG_KEEP=$k timeout -k 10 300 python tools/lab/levels_probe.py panel 2000 1400 >> $O 2>&1 || exit 1
  FEMSHELL_AMG_AGG_KEEP=$k timeout -k 10 300 python tools/lab/levels_probe.py panel 3200 1400 >> $O 2>&1 || exit 1
  FEMSHELL_AMG_AGG_KEEP=$k timeout -k 10 300 python tools/lab/levels_probe.py panel 4000 1400 >> $O 2>&1 || exit 1
done
cat $O
