#!/bin/bash
# Round 6: the 4M-triangle panel in the caller's row-major numbering against the library's Morton / Cuthill-McKee renumbering
# (FEMSHELL_REORDER): slices become compact patches, more transposed products stay inside a slice (LDS instead of HBM).
mkdir -p gpurun_out
out=gpurun_out/r06_reorder_ab.txt
: > $out
q="--no-cpu-baseline --no-full-parity --no-fullsize-parity --jacobi-probe-iters 0"
for lap in 1 2; do
  for v in "" morton rcm; do
    echo "lap $lap reorder=${v:-none}" >> $out
    FEMSHELL_REORDER=$v FEMSHELL_BENCH_DETAIL_DIR=/tmp/rr python3 bench.py --gpus 1 --steps 20 --warmup 5 $q | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
dd=json.load(open('/tmp/rr/bench_detail.json'))
print('  ms_per_step %.4f  kernel %s %.4f ms  cg %.1f it/s  spmv %.4f ms  update %.4f ms  tts %.3f s / %d' % (d['ms_per_step'], d['roofline']['kernel'], d['roofline']['ms_per_launch'], d['cg_iters_per_s'], dd['roofline_cg_spmv']['ms_per_launch'], dd['roofline_cg_update']['ms_per_launch'], d['time_to_solution_s'], d['time_to_solution_iterations']))" >> $out
  done
done
cat $out
