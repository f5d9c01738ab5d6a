#!/bin/bash
# Round 6, item 3: the chunked aggregation (FEMSHELL_AMG_AGG_CHUNK, default 131072 rows per piece) against the whole-graph one (0):
# iterations, setup and solve seconds on the four north-star meshes; then the laps of one setup of the panel.
mkdir -p gpurun_out
out=gpurun_out/r06_chunked_agg.txt
: > $out
for ch in 0 131072 65536; do
  for mesh in "panel 1414" "cylinder 1414" "roof 354"; do
    echo "== chunk $ch  $mesh" >> $out
    FEMSHELL_AMG_AGG_CHUNK=$ch python3 tools/amg_probe.py $mesh 2>&1 | grep -E '^\{' | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    print('   iterations %d  solve %.4f s  setup %.3f s  err_est %.2e' % (d['iterations'], d['solve_seconds'], d['pc_setup_seconds'], d['error_estimate']))" >> $out
  done
  echo "== chunk $ch  flap 500x1000" >> $out
  FEMSHELL_AMG_AGG_CHUNK=$ch python3 tools/flap_amg_probe.py 500x1000 2>&1 | tail -1 >> $out
done
FEMSHELL_AMG_VERBOSE=1 python3 tools/amg_probe.py panel 1414 2>&1 | grep "amg setup" > gpurun_out/r06_setup_laps.txt
FEMSHELL_HOST_POOL=0 FEMSHELL_AMG_VERBOSE=1 python3 tools/amg_probe.py panel 1414 2>&1 | grep -E "amg setup|^\{" > gpurun_out/r06_setup_laps_nopool.txt
cat $out; cat gpurun_out/r06_setup_laps.txt; grep "^{" gpurun_out/r06_setup_laps_nopool.txt | cut -c1-200
