#!/bin/bash
# What one synchronous femshell_assemble costs beyond its kernel (round 6; the signal variant was built, measured and removed):
# in mapped host memory that a one-lane kernel writes behind the assembly (FEMSHELL_ASM_SIGNAL=1, the default) against the stream
# synchronisation of round 5 (=0); with it, the status word in mapped host memory / a copy (FEMSHELL_STATUS_MAPPED), the event pair
# around the launch (FEMSHELL_ASM_EVENTS), the runtime's completion signals polled instead of interrupt-driven (HSA_ENABLE_INTERRUPT=0).
# -> gpurun_out/r06_sync_step_ab.txt
mkdir -p gpurun_out
out=gpurun_out/r06_sync_step_ab.txt
: > $out
q="--no-cpu-baseline --no-full-parity --no-fullsize-parity --jacobi-probe-iters 0 --profile"
for lap in 1 2 3; do
  for v in "1 1 1 1" "0 1 1 1" "0 0 1 1" "0 1 0 1" "0 1 1 0"; do
    set -- $v
    echo "lap $lap signal=$1 mapped=$2 events=$3 hsa_interrupt=$4" >> $out
    HSA_ENABLE_INTERRUPT=$4 FEMSHELL_ASM_SIGNAL=$1 FEMSHELL_STATUS_MAPPED=$2 FEMSHELL_ASM_EVENTS=$3 FEMSHELL_BENCH_DETAIL_DIR=/tmp python3 bench.py --gpus 1 --steps 20 --warmup 5 $q | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ms_per_step %.4f  kernel %.4f ms  value %.4g  cg %.1f it/s' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['value'], d['cg_iters_per_s']))" >> $out
  done
done
cat $out
