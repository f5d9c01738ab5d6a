#!/bin/bash
# the coarsest inverse (7386 dofs): pivot inverse inside the trailing update's launch (FEMSHELL_AMG_DENSE_LOOKAHEAD) and the
# panel kernel's rows split over two workgroups (FEMSHELL_AMG_DENSE_PANEL_SPLIT), alternating on one box
for cfg in "1 1" "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $cfg
  echo "== LOOKAHEAD=$1 PANEL_SPLIT=$2"
  FEMSHELL_AMG_DENSE_LOOKAHEAD=$1 FEMSHELL_AMG_DENSE_PANEL_SPLIT=$2 python3 tools/amg_probe.py panel 1414 2>&1 | grep -E "dense inverse" | cut -c1-200
done
