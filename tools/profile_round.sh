#!/bin/bash
# Collects what profiles/ holds for a round, on the GPU box:  tools/profile_round.sh r05
#   gpurun_out/<tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of bench.py --profile (assembly + block-Jacobi CG kernels)
#   gpurun_out/<tag>_pmc_hbm_traffic.json    FETCH_SIZE / WRITE_SIZE of the same, one counter per pass (tools/pmc_summary.py)
#   gpurun_out/<tag>_pmc_fp64.json           FP64 instruction counters of the same (tools/pmc_flops.py)
#   gpurun_out/<tag>_amg_kernel_stats.csv, _amg_kernels_by_level.txt, _amg_by_level.json, _amg_traffic_by_level.txt
#                                            the multigrid solve of the 4M-triangle panel: trace by level, traffic by level
#   gpurun_out/<tag>_pmc_mfma.json           matrix-core counters of the coarsest inverse (tools/pmc_mfma.py)
#   gpurun_out/<tag>_bench.json, _bench_roof.json   bench.py default run (reads the summaries above from profiles/) and configs[1]
#   gpurun_out/<tag>_bench_detail.json, _bench_roof_detail.json   their detail records (bench_detail.json of each run)
# Every profiler pass runs under its own timeout; no TA_* counters (they hang rocprofv3 on this pool); counters in passes of
# their own, never together with the runtime / hip / hsa trace domains.
set -u
tag=${1:-r05}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
one() { find "$1" -name "$2" | head -1; }
# ---- assembly + block-Jacobi CG kernels (bench.py --profile)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o s -- python3 bench.py --steps 40 --warmup 5 --profile > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
cp $out/${tag}_stats/s_kernel_stats.csv $out/${tag}_kernel_stats.csv 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --profile > /dev/null 2> $out/${tag}_pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --profile > /dev/null 2> $out/${tag}_pmc_write.err
python3 tools/pmc_summary.py $out/${tag}_pmc_fetch/f_counter_collection.csv $out/${tag}_pmc_write/w_counter_collection.csv $out/${tag}_pmc_hbm_traffic.json
for cnt in SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64; do
  timeout 300 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $out/${tag}_pmc_$cnt -o c -- python3 bench.py --steps 3 --warmup 1 --profile > /dev/null 2> $out/${tag}_pmc_$cnt.err
done
python3 tools/pmc_flops.py $out/${tag}_pmc_SQ_INSTS_VALU_FMA_F64/c_counter_collection.csv $out/${tag}_pmc_SQ_INSTS_VALU_MUL_F64/c_counter_collection.csv $out/${tag}_pmc_SQ_INSTS_VALU_ADD_F64/c_counter_collection.csv $out/${tag}_pmc_fp64.json
rm -rf $out/${tag}_pmc_SQ_INSTS_VALU_FMA_F64 $out/${tag}_pmc_SQ_INSTS_VALU_MUL_F64 $out/${tag}_pmc_SQ_INSTS_VALU_ADD_F64 $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_stats
# ---- the multigrid solve: kernel statistics, the trace by (kernel, grid size) = by level of the hierarchy, per-level
# milliseconds per outer iteration, HBM traffic by level, matrix-core counters of the coarsest inverse
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_amg_stats -o a -- python3 tools/amg_probe.py panel 1414 > $out/${tag}_amg_probe.txt 2> $out/${tag}_amg_stats.err
cp $(one $out/${tag}_amg_stats "*kernel_stats.csv") $out/${tag}_amg_kernel_stats.csv 2> /dev/null
python3 tools/kernel_trace_by_grid.py $(one $out/${tag}_amg_stats "*kernel_trace.csv") $out/${tag}_amg_kernels_by_level.txt
python3 tools/amg_level_times.py $out/${tag}_amg_kernels_by_level.txt $out/${tag}_amg_probe.txt $out/${tag}_amg_by_level.json
FEMSHELL_AMG_DENSE_LOOKAHEAD=0 timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_amg_fetch -o f -- python3 tools/amg_probe.py panel 1414 > /dev/null 2> $out/${tag}_amg_fetch.err
FEMSHELL_AMG_DENSE_LOOKAHEAD=0 timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_amg_write -o w -- python3 tools/amg_probe.py panel 1414 > /dev/null 2> $out/${tag}_amg_write.err
python3 tools/pmc_by_grid.py $(one $out/${tag}_amg_fetch "*counter_collection.csv") $(one $out/${tag}_amg_write "*counter_collection.csv") $(one $out/${tag}_amg_stats "*kernel_trace.csv") > $out/${tag}_amg_traffic_by_level.txt
# (counter passes serialise launches: the look-ahead of the dense inverse, which waits across two streams, is switched off in them --
#  in ALL counter passes over the multigrid solve, or each would first run into its bounded waits)
FEMSHELL_AMG_DENSE_LOOKAHEAD=0 timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_mfma_busy -o b -- python3 tools/amg_probe.py panel 1414 > /dev/null 2> $out/${tag}_mfma_busy.err
FEMSHELL_AMG_DENSE_LOOKAHEAD=0 timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $out/${tag}_mfma_mops -o m -- python3 tools/amg_probe.py panel 1414 > /dev/null 2> $out/${tag}_mfma_mops.err
python3 tools/pmc_mfma.py $(one $out/${tag}_mfma_busy "*counter_collection.csv") $(one $out/${tag}_mfma_mops "*counter_collection.csv") $(one $out/${tag}_amg_stats "*kernel_trace.csv") $out/${tag}_pmc_mfma.json $out/${tag}_amg_probe.txt
rm -rf $out/${tag}_amg_stats $out/${tag}_amg_fetch $out/${tag}_amg_write $out/${tag}_mfma_busy $out/${tag}_mfma_mops
# ---- the default bench run comes last and sees the summaries (bench.py attaches `traffic` and the per-level times only from a
# profile taken with the same kernel sources; on the box the copies under profiles/ are scratch -- commit them from gpurun_out/)
for f in pmc_hbm_traffic.json pmc_fp64.json amg_by_level.json pmc_mfma.json; do cp $out/${tag}_$f profiles/${tag}_$f 2> /dev/null; done
timeout 1100 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cp bench_detail.json $out/${tag}_bench_detail.json 2> /dev/null  # (the detail record of THIS run: the next one writes the same file)
tail -c 600 $out/${tag}_bench.err
# configs[1] at full size as a run of its own (configs[2], the cylinder, is part of the default bench line)
timeout 400 python3 bench.py --workload roof --no-cpu-baseline --no-full-parity --jacobi-probe-iters 0 > $out/${tag}_bench_roof.json 2> $out/${tag}_bench_roof.err
cp bench_detail.json $out/${tag}_bench_roof_detail.json 2> /dev/null
python3 tools/bench_summary.py < $out/${tag}_bench.json 2> /dev/null | head -12
