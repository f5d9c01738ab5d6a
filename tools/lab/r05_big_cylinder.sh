#!/bin/bash
# does the smoother's single-precision copy still do on curved shells beyond the north-star size?  16M- and 32M-triangle pinched
# cylinders, default against FEMSHELL_AMG_SMOOTH_F32=0 (all FP64)
mkdir -p gpurun_out
O=gpurun_out/r05_big_cylinder.txt
: > $O
for n in 2828 4000; do
  for v in 1 0; do
    echo "== cylinder $n x $n, FEMSHELL_AMG_SMOOTH_F32=$v" >> $O
    FEMSHELL_AMG_SMOOTH_F32=$v timeout -k 10 400 python tools/amg_probe.py cylinder $n 2>&1 | grep -E "wall_s|n_nodes" | cut -c1-900 >> $O || exit 1
  done
done
cat $O | cut -c1-400
