#!/bin/bash
# where the one-time costs of the first solve go (4M-triangle panel): plan phases and multigrid setup laps
FEMSHELL_PLAN_VERBOSE=1 FEMSHELL_AMG_VERBOSE=1 python3 tools/amg_probe.py panel 1414 2>&1 | cut -c1-260
