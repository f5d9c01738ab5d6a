#!/bin/bash
# Issue / wait / occupancy counters of the assembly kernel, one rocprofv3 pass per counter group (never with the trace domains):
#   tools/pmc_asm.sh <tag>     -> gpurun_out/<tag>_summary.txt       
set -u
tag=${1:-asm}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export CG=0
: > $out/${tag}_summary.txt
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_WAVES_EQ_64 SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/${tag}_p$i -o c -- python3 tools/asm_only.py > $out/${tag}_p$i.log 2>&1 || { echo "group $i ($grp) failed: $(tail -2 $out/${tag}_p$i.log)" >> $out/${tag}_summary.txt; }
  python3 - $out/${tag}_p$i <<'PY' >> $out/${tag}_summary.txt
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_assemble" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        v = sorted(v.values())
        print("%-28s launches %d  median %.5g" % (k, len(v), v[len(v) // 2]))
PY
  rm -rf $out/${tag}_p$i
done
cat $out/${tag}_summary.txt
