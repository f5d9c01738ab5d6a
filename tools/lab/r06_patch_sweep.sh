# sweep of the patch smoother's two parameters on the 20k-point Delaunay shell numbered along x (all FP64 preconditioner)
for mc in 4 6 8 10; do
  for tau in 0.9 0.8 0.7 0.6 0.5; do
    echo -n "max $mc tau $tau: "
    NUMBERING=${NUMBERING:-x} FEMSHELL_AMG_SMOOTH_F32=0 FEMSHELL_AMG_PATCH_MAX=$mc TAUS=$tau timeout -k 10 200 python3 tools/lab/r06_patch_probe.py 20000 3 1500 2>&1 | tail -1 | sed -e 's/patch {.*clusters.: \([0-9]*\), .nodes_in_clusters.: \([0-9]*\).*/clusters \1 nodes \2/' | cut -c1-200
  done
done
