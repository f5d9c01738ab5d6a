#!/bin/bash
# A/B on one box: the driver's bench command with only the W warm-up steps of the contract against the same with round 5's
# extra 30 assemblies + 200 CG iterations of device warm-up in front (FEMSHELL_BENCH_EXTRA_WARMUP=1); alternating, 3 laps.
# Writes gpurun_out/r06_warmup_ab.txt (summarised in profiles/r06_warmup_ab.txt).
set -e
mkdir -p gpurun_out
out=gpurun_out/r06_warmup_ab.txt
: > $out
q="--no-cpu-baseline --no-full-parity --no-fullsize-parity --jacobi-probe-iters 0"
for lap in 1 2 3; do
  for extra in 0 1; do
    echo "lap $lap extra_warmup=$extra" >> $out
    FEMSHELL_BENCH_EXTRA_WARMUP=$extra FEMSHELL_BENCH_DETAIL_DIR=/tmp python3 bench.py --gpus 1 --steps 20 --warmup 5 $q | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  value %.4g el/s  ms_per_step %.4f  kernel %.4f ms  cg %.1f it/s  tts %.3f s / %d' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['cg_iters_per_s'], d['time_to_solution_s'], d['time_to_solution_iterations']))" >> $out
  done
done
cat $out
