#!/bin/bash
# What one synchronous femshell_assemble costs beyond its kernel (round 6): the status word in mapped host memory (no copy), no
# event pair around the launch, and the runtime's completion signals polled instead of interrupt-driven (HSA_ENABLE_INTERRUPT=0),
# against round 5's step.  Alternating on one box.  -> gpurun_out/r06_sync_step_ab.txt
mkdir -p gpurun_out
out=gpurun_out/r06_sync_step_ab.txt
: > $out
q="--no-cpu-baseline --no-full-parity --no-fullsize-parity --jacobi-probe-iters 0 --profile"
for lap in 1 2 3; do
  for v in "1 1 1" "0 1 1" "1 0 1" "1 1 0"; do
    set -- $v
    echo "lap $lap mapped=$1 events=$2 hsa_interrupt=$3" >> $out
    HSA_ENABLE_INTERRUPT=$3 FEMSHELL_STATUS_MAPPED=$1 FEMSHELL_ASM_EVENTS=$2 FEMSHELL_BENCH_DETAIL_DIR=/tmp python3 bench.py --gpus 1 --steps 20 --warmup 5 $q | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ms_per_step %.4f  kernel %.4f ms  value %.4g  cg %.1f it/s' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['value'], d['cg_iters_per_s']))" >> $out
  done
done
cat $out
