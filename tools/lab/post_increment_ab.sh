#!/bin/bash
# A/B of the residual the post-smoothing starts from: b - A x as an FP64 product with the iterate (FEMSHELL_AMG_POST_INCREMENT=0)
# against (restricted residual) - A (P x_c) on the smoother's copy of the operator (1, default), on the BASELINE-size meshes,
# alternating on one box.  Output: gpurun_out/post_increment_ab.txt
out=gpurun_out/post_increment_ab.txt
mkdir -p gpurun_out
: > $out
for w in "panel 1414" "cylinder 1414" "roof 354"; do
  for f in 0 1 0 1; do
    echo "== $w  FEMSHELL_AMG_POST_INCREMENT=$f" >> $out
    FEMSHELL_AMG_POST_INCREMENT=$f python3 tools/amg_probe.py $w 2>&1 | grep "second solve" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l.split(':', 1)[1])
    print('   iterations %d  solve %.4f s  true_rel_residual %.3e  error_estimate %.2e' % (d['iterations'], d['solve_seconds'], d['true_rel_residual'], d['error_estimate']))" >> $out || exit 1
  done
done
