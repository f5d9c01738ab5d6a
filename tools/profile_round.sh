#!/bin/bash
# Collects what profiles/ holds for a round, on the GPU box:  tools/profile_round.sh r02
#   gpurun_out/<tag>_bench.json            bench.py default run (parity probe + CPU baseline included)
#   gpurun_out/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of bench.py --profile
#   gpurun_out/<tag>_pmc_hbm_traffic.json  FETCH_SIZE / WRITE_SIZE, one counter per pass (tools/pmc_summary.py)
# Every profiler pass runs under its own timeout; no TA_* counters (they hang rocprofv3 on this pool).
set -u
tag=${1:-r04}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- python3 bench.py --steps 10 --warmup 2 --profile > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
cp $out/${tag}_stats/s_kernel_stats.csv $out/${tag}_kernel_stats.csv 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --profile > /dev/null 2> $out/${tag}_pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --profile > /dev/null 2> $out/${tag}_pmc_write.err
python3 tools/pmc_summary.py $out/${tag}_pmc_fetch/f_counter_collection.csv $out/${tag}_pmc_write/w_counter_collection.csv $out/${tag}_pmc_hbm_traffic.json
# the default bench run comes after the counter passes and sees their summary (bench.py attaches `traffic` only from a
# profile taken with the same kernel sources; on the box the copy under profiles/ is scratch -- commit it from gpurun_out/)
cp $out/${tag}_pmc_hbm_traffic.json profiles/${tag}_pmc_hbm_traffic.json 2> /dev/null
timeout 900 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
# FP64 work of the kernels (the assembly kernel is VALU-bound with symmetric storage): instruction counters, one pass each
for cnt in SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64; do
  timeout 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $out/${tag}_pmc_$cnt -o c -- python3 bench.py --steps 3 --warmup 1 --profile > /dev/null 2> $out/${tag}_pmc_$cnt.err
done
python3 tools/pmc_flops.py $out/${tag}_pmc_SQ_INSTS_VALU_FMA_F64/c_counter_collection.csv $out/${tag}_pmc_SQ_INSTS_VALU_MUL_F64/c_counter_collection.csv $out/${tag}_pmc_SQ_INSTS_VALU_ADD_F64/c_counter_collection.csv $out/${tag}_pmc_fp64.json
rm -rf $out/${tag}_pmc_SQ_INSTS_VALU_FMA_F64 $out/${tag}_pmc_SQ_INSTS_VALU_MUL_F64 $out/${tag}_pmc_SQ_INSTS_VALU_ADD_F64
# keep the merged-back directory small: the raw counter tables are large
rm -rf $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_stats
tail -1 $out/${tag}_bench.json
# configs[1] at full size as a run of its own (configs[2], the cylinder, is part of the default bench line)
timeout 400 python3 bench.py --workload roof --no-cpu-baseline --no-full-parity --jacobi-probe-iters 0 > $out/${tag}_bench_roof.json 2> $out/${tag}_bench_roof.err
# the multigrid solve: kernel statistics and the same trace by (kernel, grid size) = by level of the hierarchy
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_amg_stats -o a -- python3 tools/amg_probe.py panel 1414 > $out/${tag}_amg_probe.txt 2> $out/${tag}_amg_stats.err
cp $(find $out/${tag}_amg_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_amg_kernel_stats.csv 2> /dev/null
python3 tools/kernel_trace_by_grid.py $(find $out/${tag}_amg_stats -name "*kernel_trace.csv" | head -1) $out/${tag}_amg_kernels_by_level.txt
rm -rf $out/${tag}_amg_stats
