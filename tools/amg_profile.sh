#!/bin/bash
# rocprofv3 kernel statistics of the multigrid-preconditioned solve of the 4M-triangle panel (setup + two solves):
#   tools/amg_profile.sh [tag]   ->  gpurun_out/<tag>_amg_kernel_stats.csv
set -u
tag=${1:-r02}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_amg_stats -o a -- python3 tools/amg_probe.py panel 1414 > $out/${tag}_amg_probe.txt 2> $out/${tag}_amg_stats.err
cp $out/${tag}_amg_stats/a_kernel_stats.csv $out/${tag}_amg_kernel_stats.csv 2> /dev/null
rm -rf $out/${tag}_amg_stats
head -12 $out/${tag}_amg_kernel_stats.csv | cut -c1-60
