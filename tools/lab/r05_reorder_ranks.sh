#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_multirank_gpu.py -q --timeout 400 -s -k "renumbered or randomly" > gpurun_out/r05_reorder_ranks.txt 2>&1
echo "exit $?" >> gpurun_out/r05_reorder_ranks.txt
tail -40 gpurun_out/r05_reorder_ranks.txt
