#!/bin/bash
# Twin of the reference's src/fem-shell/run_examples.sh (tests A-G), running the MI355X
# FEM-shell on the same example meshes with the same parameters (fixtures in tests/golden/meshes).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
MESHES="${MESHES:-$HERE/../../tests/golden/meshes}"
OUT="${OUT:-example-out}"
[ -x "$HERE/FEM-shell" ] || make -C "$HERE" -s
mkdir -p "$OUT"
run() { echo "Test $1: "; shift; "$HERE/FEM-shell" "$@"; }
run A -nu 0.25 -e 30000 -t 1.0 -mesh "$MESHES/test_A_uv_t.xda" -out "$OUT/test_A_uv_t"
run B -nu 0.25 -e 30000 -t 1.0 -mesh "$MESHES/test_B_uv_q.xda" -out "$OUT/test_B_uv_q"
run C -nu 0.3 -e 10.92 -t 1.0 -mesh "$MESHES/test_C_w_tA16.xda" -out "$OUT/test_C_w_tA16"
run D -nu 0.3 -e 1e7 -t 0.5 -mesh "$MESHES/test_D_w_q_uni16.xda" -out "$OUT/test_D_w_q_uni16"
run E -nu 0.25 -e 10000 -t 0.25 -mesh "$MESHES/test_E_uvw_t.xda" -out "$OUT/test_E_uvw_t"
run F -nu 0.3 -e 1.7472e7 -t 0.01 -mesh "$MESHES/test_F_032_ss_uni.xda" -out "$OUT/test_F_032_ss_uni"
run G -nu 0.3 -e 1e7 -t 0.5 -mesh "$MESHES/test_G_mpi_64_q.xda" -out "$OUT/test_G_mpi_64_q"
echo "....all examples finished!"
