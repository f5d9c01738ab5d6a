#!/bin/bash
# BASELINE configs[4] at full size on one GPU: the perpendicular flap 0.1 x 1 (dead axis y), 500 x 1000 squares -> 1,000,000
# tri3, E=1e6 nu=0.3 t=0.1, bottom edge id 20, other edges id 2, forces from the in-process dummy fluid; a few time steps of
# the serial-implicit coupling with the multigrid-preconditioned solve (K and the hierarchy are built once).
#   tools/coupled_flap_full.sh [steps] [nx] [nz]
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
HOST="$HERE/fem-shell_amd/host"
steps=${1:-3}; nx=${2:-500}; nz=${3:-1000}
make -C "$HOST" -s
tmp=$(mktemp -d)
"$HOST/meshGen" t $nx $nz 0 0 0.1 1 2,20,2,2 1 0 1 y "$tmp/flap" > /dev/null
ls -la "$tmp/flap.xda" | awk '{print "mesh file:", $5, "bytes"}'
t0=$(date +%s%N)
"$HOST/FEM-shell-precice" -nu 0.3 -e 1e6 -t 0.1 -mesh "$tmp/flap.xda" \
    -config "$HERE/tests/golden/coupling/inprocess_config.xml" -dt 0.01 -axis y -steps $steps -fluid edge -pc_type gamg -ksp_rtol 1e-10 2>&1 | grep -v "^tip\[[1-9][0-9]" | tail -25
echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms (mesh reading, symbolic phase, assembly, multigrid setup, coupling loop)"
rm -rf "$tmp"
