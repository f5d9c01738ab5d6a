cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out; tag=r04a
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_amg_stats -o a -- python3 tools/amg_probe.py panel 1414 > $out/${tag}_amg_probe.txt 2> $out/${tag}_amg_stats.err
cp $(find $out/${tag}_amg_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_amg_kernel_stats.csv 2> /dev/null
python3 tools/kernel_trace_by_grid.py $(find $out/${tag}_amg_stats -name "*kernel_trace.csv" | head -1) $out/${tag}_amg_kernels_by_level.txt
rm -rf $out/${tag}_amg_stats
head -60 $out/${tag}_amg_kernels_by_level.txt
