#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r05_cg_single.txt
: > $O
for v in 0 1 0 1; do
  echo "== FEMSHELL_CG_SINGLE_REDUCTION=$v" >> $O
  FEMSHELL_CG_SINGLE_REDUCTION=$v timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --profile 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('   cg it/s %.1f  ms/iter %.4f  assembly %.3f ms' % (d['cg_iters_per_s'], d['cg_ms_per_iter'], d['ms_per_step']))" >> $O
done
cat $O
