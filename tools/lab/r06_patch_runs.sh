# the patch smoother with candidate defaults on several shells / numberings (all FP64 preconditioner)
for par in "0.7 8" "0.8 8" "0.7 6" "0.6 8"; do
  set -- $par
  for cfg in "NUMBERING=gen N=20000 S=3" "NUMBERING=x N=20000 S=3" "NUMBERING=gen FEMSHELL_REORDER=morton N=20000 S=3" "NUMBERING=gen FEMSHELL_REORDER=rcm N=20000 S=3" "NUMBERING=gen N=3000 S=2" "NUMBERING=x N=3000 S=2" "NUMBERING=x N=50000 S=5" "NUMBERING=x N=700 S=1"; do
    echo -n "tau $1 max $2 | $cfg: "
    env $cfg FEMSHELL_AMG_SMOOTH_F32=0 FEMSHELL_AMG_PATCH_MAX=$2 TAUS=$1 bash -c 'timeout -k 10 200 python3 tools/lab/r06_patch_probe.py $N $S 1500' 2>&1 | tail -1 | sed -e 's/patch {.*clusters.: \([0-9]*\), .nodes_in_clusters.: \([0-9]*\).*/clusters \1 nodes \2/' | cut -c10-170
  done
done
