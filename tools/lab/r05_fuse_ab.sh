#!/bin/bash
# A/B of FEMSHELL_AMG_FUSE on one box: three solves per setting on the 4M-triangle panel / cylinder (tools/lab/solve_time_probe.py)
set -u
out=gpurun_out/r05b
mkdir -p $out
for round in 1 2; do
  for fuse in ${FUSE_LIST:-0 -1}; do
    echo "== FEMSHELL_AMG_FUSE=$fuse round $round"
    FEMSHELL_AMG_FUSE=$fuse python3 tools/lab/solve_time_probe.py ${1:-panel} 1414
  done
done
