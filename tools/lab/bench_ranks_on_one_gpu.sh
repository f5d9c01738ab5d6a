#!/bin/bash
# Rehearsal of the driver's N > 1 bench line at full size on ONE GPU: N ranks share device 0 and talk through the test
# transport (tests/helpers/fake_rccl) -- checks the path, its memory and its wall time, not scaling.   [N]
set -u
N=${1:-2}
HERE="$(cd "$(dirname "$0")/../.." && pwd)"
make -C "$HERE/tests/helpers/fake_rccl" -s
export FEMSHELL_RCCL_LIB="$HERE/tests/helpers/fake_rccl/libfake_rccl.so" FEMSHELL_BENCH_SAME_DEVICE=1 MASTER_ADDR=127.0.0.1
S=$(date +%s)
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 "$HERE/bench.py" --gpus $N --steps 20 --warmup 3
echo "wall $(( $(date +%s) - S )) s for N=$N"
