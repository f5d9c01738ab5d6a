#!/bin/bash
# the pivot inverse of the coarsest inverse inside the trailing update's launch (FEMSHELL_AMG_DENSE_LOOKAHEAD=1) against a launch of its own (default)
for f in 1 0 1 0; do
  echo "== FEMSHELL_AMG_DENSE_LOOKAHEAD=$f"
  FEMSHELL_AMG_DENSE_LOOKAHEAD=$f python3 tools/amg_probe.py panel 1414 2>&1 | grep -E "dense inverse|second solve" | cut -c1-330
done
