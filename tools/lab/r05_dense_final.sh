#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_amg.py -x -q --timeout 200 > gpurun_out/r05_dense_tests.txt 2>&1; echo "tests exit $?" >> gpurun_out/r05_dense_tests.txt
tail -3 gpurun_out/r05_dense_tests.txt
bash tools/lab/r05_dense_trace.sh "FEMSHELL_AMG_DENSE_TILE=64" "FEMSHELL_AMG_DENSE_LOOKAHEAD=0" 2>&1 | grep -E "^==|dense inverse|update|pivot|panels"
bash tools/lab/r05_dense.sh "FEMSHELL_AMG_DENSE_TILE=64" "FEMSHELL_AMG_DENSE_TILE=128" 2>&1 | grep -E "^==|dense inverse"
