#!/bin/bash
# Round 6, item 2 (c): the K cycle on level 1 only (FEMSHELL_AMG_K_LEVELS=1: the levels below are visited once per visit of their
# parent) against the K cycle on every level -- iterations and solve time on the three north-star meshes, and the 1/8 strip.
mkdir -p gpurun_out
out=gpurun_out/r06_klevels.txt
: > $out
for mesh in "panel 1414" "cylinder 1414" "roof 354"; do
  for k in "" 1; do
    echo "== $mesh K_LEVELS=${k:-all}" >> $out
    FEMSHELL_AMG_K_LEVELS=$k python3 tools/amg_probe.py $mesh 2>&1 | grep -E '^\{|second solve' | python3 -c "
import sys, json
for ln in sys.stdin:
    ln = ln[ln.index('{'):]
    d = json.loads(ln)
    print('   iterations %d  solve %.4f s  setup %.3f s  err_est %.2e' % (d['iterations'], d['solve_seconds'], d['pc_setup_seconds'], d['error_estimate']))" >> $out
  done
done
for k in "" 1; do
  echo "== strip N=8 K_LEVELS=${k:-all}" >> $out
  FEMSHELL_AMG_K_LEVELS=$k python3 tools/lab/dist_budget_probe.py 8 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   iterations %d  ms_per_iteration %.3f' % (d['iterations'], d['ms_per_iteration']))" >> $out
done
cat $out
