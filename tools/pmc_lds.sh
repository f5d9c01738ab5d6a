#!/bin/bash
# LDS counters of the assembly kernel (one rocprofv3 pass per counter group):  tools/pmc_lds.sh <tag>
set -u
tag=${1:-lds}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export CG=0
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/${tag}_p$i -o c -- python3 tools/asm_only.py > $out/${tag}_p$i.log 2>&1
  python3 - $out/${tag}_p$i <<'PY' >> $out/${tag}_summary.txt
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_assemble" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-28s launches %d  median %.4g" % (k, len(v), sorted(v)[len(v) // 2]))
PY
  rm -rf $out/${tag}_p$i
done
cat $out/${tag}_summary.txt
