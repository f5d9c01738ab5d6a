#!/bin/bash
# Round 6: arenas of 512 MB (requests below 256 MB... the round's first setting was 96 MB) against 2048 MB (requests below 1 GB),
# multigrid setup of the 4M-triangle panel: hipMalloc calls and their time as FEMSHELL_AMG_VERBOSE prints them, setup seconds.
out=gpurun_out/r06_arena
mkdir -p $out
: > $out/log.txt
for mb in 512 2048 512 2048; do
  echo "== FEMSHELL_POOL_ARENA_MB=$mb" >> $out/log.txt
  FEMSHELL_POOL_ARENA_MB=$mb FEMSHELL_AMG_VERBOSE=1 python tools/amg_probe.py panel 1414 2>&1 | grep -E "device allocator|pc_setup_seconds\": 0\.[0-9]" | sed -E 's/.*"solve_seconds": ([0-9.]+).*"pc_setup_seconds": ([0-9.]+).*/   solve \1 s setup \2 s/' >> $out/log.txt
done
cat $out/log.txt
