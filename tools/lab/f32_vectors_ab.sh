#!/bin/bash
# A/B of what the smoothing products keep in single precision besides the operator's values (FEMSHELL_AMG_VEC_F32: 0 nothing,
# 1 their results -- direct part and transposed products --, 2 their input, the Chebyshev direction, as well) on the
# BASELINE-size meshes, alternating on one box.  Output: gpurun_out/f32_vectors_ab.txt
out=gpurun_out/f32_vectors_ab.txt
mkdir -p gpurun_out
: > $out
for w in "panel 1414" "cylinder 1414" "roof 354"; do
  for f in 0 1 2 0 2; do
    echo "== $w  FEMSHELL_AMG_VEC_F32=$f" >> $out
    FEMSHELL_AMG_VEC_F32=$f python3 tools/amg_probe.py $w 2>&1 | grep "second solve" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l.split(':', 1)[1])
    print('   iterations %d  solve %.4f s  true_rel_residual %.3e  error_estimate %.2e' % (d['iterations'], d['solve_seconds'], d['true_rel_residual'], d['error_estimate']))" >> $out || exit 1
  done
done
