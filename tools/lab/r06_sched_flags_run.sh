#!/bin/bash
# times the assembly kernel of the libraries r06_sched_flags_build.sh made, alternately with the default build (on the GPU box's copy)
cd fem-shell_amd
cp libfemshell.so libfemshell_default.so
for round in 1 2 3; do
  for tag in default maxclause bias0 relaxed; do
    [ -f libfemshell_$tag.so ] || continue
    cp libfemshell_$tag.so libfemshell.so
    echo -n "$tag: "
    (cd .. && ASM_WARMUP=100 ASM_REPS=100 CG=0 python tools/asm_only.py 2>/dev/null | grep ASM)
  done
done
cp libfemshell_default.so libfemshell.so
