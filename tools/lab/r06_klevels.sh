#!/bin/bash
# Round 6, item 2 (c): which levels need the K cycle's Krylov steps?  FEMSHELL_AMG_K_MASK: bit l = level l wrapped (default all).
# 2 = level 1 only (levels below visited once per visit of their parent), 4 = level 2 only (level 1 visited once per iteration).
# Iterations and solve time on the three north-star meshes, and the 1/8 strip.   -> gpurun_out/r06_klevels.txt
mkdir -p gpurun_out
out=gpurun_out/r06_klevels.txt
: > $out
for mesh in "panel 1414" "cylinder 1414"; do
  for k in "" 2 4; do
    echo "== $mesh K_MASK=${k:-all}" >> $out
    FEMSHELL_AMG_K_MASK=$k python3 tools/amg_probe.py $mesh 2>&1 | grep -E 'second solve' | python3 -c "
import sys, json
for ln in sys.stdin:
    ln = ln[ln.index('{'):]
    d = json.loads(ln)
    print('   iterations %d  solve %.4f s  err_est %.2e' % (d['iterations'], d['solve_seconds'], d['error_estimate']))" >> $out
  done
done
for k in "" 2 4; do
  echo "== strip N=8 K_MASK=${k:-all}" >> $out
  FEMSHELL_AMG_K_MASK=$k python3 tools/lab/dist_budget_probe.py 8 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   iterations %d  ms_per_iteration %.3f' % (d['iterations'], d['ms_per_iteration']))" >> $out
done
cat $out
