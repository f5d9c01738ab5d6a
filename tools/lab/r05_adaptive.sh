#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_adaptive.txt
: > $O
for v in 1 0 1 0; do
  for k in panel cylinder; do
    echo "== FEMSHELL_REFINE_ADAPTIVE=$v $k" >> $O
    FEMSHELL_REFINE_ADAPTIVE=$v timeout -k 10 200 python tools/amg_probe.py $k 1414 2>&1 | grep -E "wall_s|second solve" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[l.index('{'):])
    print('   iterations %d  solve %.4f s  setup %.3f s  error_estimate %.2e  correction_rel %.2e  reduction %.2e' % (d['iterations'], d['solve_seconds'], d['pc_setup_seconds'], d['error_estimate'], d['refine_correction_rel'], d['refine_residual_reduction']))" >> $O
  done
done
cat $O
timeout -k 10 400 python -m pytest tests/test_gpu_amg.py -x -q --timeout 200 2>&1 | tail -3
