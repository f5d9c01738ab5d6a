#!/bin/bash
# round 5, first GPU call: HBM traffic of the multigrid solve's kernels by level, and the box's streaming rates
set -u
out=gpurun_out/r05a
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
hipcc -O3 --offload-arch=gfx950 tools/lab/hbm_rw.hip -o tools/lab/hbm_rw && tools/lab/hbm_rw > $out/hbm_rw.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 tools/amg_probe.py panel 1414 > $out/probe_trace.txt 2> $out/trace.err
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 tools/amg_probe.py panel 1414 > $out/probe_fetch.txt 2> $out/fetch.err
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 tools/amg_probe.py panel 1414 > $out/probe_write.txt 2> $out/write.err
python3 tools/pmc_by_grid.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1) $(find $out/trace -name "*kernel_trace.csv" | head -1) > $out/amg_traffic_by_level.txt
python3 tools/kernel_trace_by_grid.py $(find $out/trace -name "*kernel_trace.csv" | head -1) $out/amg_kernels_by_level.txt
rm -rf $out/fetch $out/write $out/trace
head -40 $out/amg_traffic_by_level.txt
cat $out/hbm_rw.txt
