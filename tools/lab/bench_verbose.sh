#!/bin/bash
# the driver's bench command with the setup laps of every multigrid hierarchy on stderr
mkdir -p gpurun_out
FEMSHELL_AMG_VERBOSE=1 timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_verbose2.json 2> gpurun_out/bench_verbose2.err
python3 - <<'PY'
import json
b = json.loads(open("gpurun_out/bench_verbose2.json").read().strip().split("\n")[-1])
print("panel setup", b["time_to_solution"]["pc_setup_seconds"], "cylinder", b["config2_pinched_cylinder_4M"]["time_to_solution"])
PY
awk '/amg setup/ {v=$(NF-1); if (v+0 > 0.04) print NR": "$0}' gpurun_out/bench_verbose2.err | head -20
