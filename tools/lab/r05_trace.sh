#!/bin/bash
# kernel trace of the multigrid solve of the 4M-triangle panel, by (kernel, grid) = by level:  tools/lab/r05_trace.sh <tag> [env...]
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for kv in "$@"; do export "$kv"; done
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 tools/amg_probe.py ${PROBE_KIND:-panel} ${PROBE_N:-1414} > $out/probe.txt 2> $out/trace.err
python3 tools/kernel_trace_by_grid.py $(find $out/trace -name "*kernel_trace.csv" | head -1) $out/kernels_by_level.txt
rm -rf $out/trace
head -3 $out/probe.txt | cut -c1-400
head -45 $out/kernels_by_level.txt
