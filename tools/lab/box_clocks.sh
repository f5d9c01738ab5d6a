#!/bin/bash
# memory / fabric clocks of the box next to the per-kernel times of the CG iteration (boxes of the pool differ by 10 % on
# the symmetric SpMV while copying at the same rate)
python3 tools/cg_kernels_probe.py 2>&1 | tail -1
rocm-smi --showclocks 2>&1 | grep -i "mclk\|sclk\|fclk"
rocm-smi --showmemuse 2>&1 | grep -i "partition\|activity" | head -3
rocm-smi --showcomputepartition --showmemorypartition 2>&1 | grep -i "partition" | head -4
