#!/bin/bash
# A/B on one box: residual in front of the restriction as an increment (FEMSHELL_AMG_RESIDUAL_INCREMENT) and the coarsest
# inverse stored in single precision (FEMSHELL_AMG_DENSE_F32)
for w in "panel 1414" "cylinder 1414"; do
  for cfg in "1 0" "0 0" "1 1" "0 0" "1 0" "1 1"; do
    set -- $cfg
    echo "== $w  RESIDUAL_INCREMENT=$1 DENSE_F32=$2"
    FEMSHELL_AMG_RESIDUAL_INCREMENT=$1 FEMSHELL_AMG_DENSE_F32=$2 python3 tools/amg_probe.py $w 2>&1 | grep "second solve" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l.split(':', 1)[1])
    print('   iterations %d  solve %.4f s  error_estimate %.2e' % (d['iterations'], d['solve_seconds'], d['error_estimate']))"
  done
done
