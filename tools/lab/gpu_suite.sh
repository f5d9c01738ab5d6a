#!/bin/bash
# the whole GPU test-suite in one process, progress into gpurun_out/gpu_suite.txt
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --timeout 400 > gpurun_out/gpu_suite.txt 2>&1
rc=$?
echo "exit $rc" >> gpurun_out/gpu_suite.txt
tail -8 gpurun_out/gpu_suite.txt
exit $rc
