#!/bin/bash
# set_mesh and multigrid setup of the 4M-triangle panel against the number of host threads (the box shows 256 cores to
# sched_getaffinity under a CPU quota of 16: cpu.max).  Output: gpurun_out/host_threads.txt
out=gpurun_out/host_threads.txt
mkdir -p gpurun_out
{ echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cpu.stat before:"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null; } > $out
for t in default 8 16 32 64; do
  echo "=== FEMSHELL_HOST_THREADS=$t" >> $out
  if [ $t = default ]; then unset FEMSHELL_HOST_THREADS; else export FEMSHELL_HOST_THREADS=$t; fi
  python tools/lab/setup_phases_probe.py 2>&1 | grep -E "^==|femshell plan|set_mesh\]|graph \+ aggregation|level 0 patterns" >> $out || exit 1
done
{ echo "cpu.stat after:"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null; } >> $out
