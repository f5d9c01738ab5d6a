#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_amg.py -x -q --timeout 200 -k "dense" > gpurun_out/r05_dense_tests.txt 2>&1; echo "tests exit $?" >> gpurun_out/r05_dense_tests.txt
tail -4 gpurun_out/r05_dense_tests.txt
bash tools/lab/r05_dense_trace.sh "FEMSHELL_AMG_DENSE_PIPE=0" "FEMSHELL_AMG_DENSE_PIPE=1" "FEMSHELL_AMG_DENSE_PIPE=0 FEMSHELL_AMG_DENSE_LOOKAHEAD=0" 2>&1 | grep -E "^==|dense inverse|update<|pivot|panels"
