#!/bin/bash
# how many significant bits does the smoother's copy of the level operators need?  iterations / time per setting, one solve each
for kind in panel cylinder; do
  for sig in 24 20 18 17 16 14 12; do
    echo "== $kind sigbits $sig"
    FEMSHELL_AMG_SMOOTH_SIGBITS=$sig FEMSHELL_AMG_FUSE=3 python3 tools/lab/solve_time_probe.py $kind 1414 2>&1 | head -1
  done
done
echo "== roof 354"
for sig in 24 17 16 14; do FEMSHELL_AMG_SMOOTH_SIGBITS=$sig FEMSHELL_AMG_FUSE=3 python3 tools/lab/solve_time_probe.py roof 354 2>&1 | head -1; done
