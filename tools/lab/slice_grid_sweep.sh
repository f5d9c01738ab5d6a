#!/bin/bash
# FEMSHELL_SLICE_GRID (workgroups of the slice-walking kernels) on the 4M-triangle panel: last line of cg_kernels_probe per value
for g in 65536 32768 16384 8192 4096 2048; do
  echo -n "grid $g: "; FEMSHELL_SLICE_GRID=$g python3 tools/cg_kernels_probe.py 2>&1 | tail -1
done
