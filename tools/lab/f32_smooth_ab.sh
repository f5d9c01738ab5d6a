#!/bin/bash
# A/B of single-precision smoothing products (FEMSHELL_AMG_SMOOTH_F32: 0 off, 2 level 0 only, 1 levels >= 4096 nodes) on the BASELINE-size meshes
for w in "panel 1414" "cylinder 1414" "roof 354"; do
  for f in 0 2 1 0 1; do
    echo "== $w  FEMSHELL_AMG_SMOOTH_F32=$f"
    FEMSHELL_AMG_SMOOTH_F32=$f python3 tools/amg_probe.py $w 2>&1 | grep "second solve" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l.split(':', 1)[1])
    print('   iterations %d  solve %.4f s  true_rel_residual %.3e  error_estimate %.2e' % (d['iterations'], d['solve_seconds'], d['true_rel_residual'], d['error_estimate']))"
  done
done
