#!/bin/bash
# matrix-core counters of the coarsest inverse (round 5): three rocprofv3 passes over the 4M-triangle probe
set -u
out=gpurun_out/r05_mfma
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 tools/amg_probe.py panel 1414 > $out/probe.txt 2> $out/trace.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/busy -o b -- python3 tools/amg_probe.py panel 1414 > /dev/null 2> $out/busy.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $out/mops -o m -- python3 tools/amg_probe.py panel 1414 > /dev/null 2> $out/mops.err
python3 tools/pmc_mfma.py $(find $out/busy -name "*counter_collection.csv" | head -1) $(find $out/mops -name "*counter_collection.csv" | head -1) $(find $out/trace -name "*kernel_trace.csv" | head -1) $out/pmc_mfma.json
rm -rf $out/trace $out/busy $out/mops
grep "dense inverse" $out/probe.txt
