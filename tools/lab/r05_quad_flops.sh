#!/bin/bash
# FP64 instruction counters of the QUAD4 assembly kernel on the 2000 x 2000 quad panel (one rocprofv3 pass; counters only).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r05_quad_flops
rm -rf $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d $O -o c -- python3 tools/quad_probe.py 2000 > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
python3 - $O <<'PY' > gpurun_out/r05_quad_flops.txt
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_assemble" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    rows = list(acc.values())
    r = rows[len(rows) // 2]
    fl = 64.0 * (2 * r["SQ_INSTS_VALU_FMA_F64"] + r["SQ_INSTS_VALU_MUL_F64"] + r["SQ_INSTS_VALU_ADD_F64"])
    print("launches %d; one launch: FMA %.4g MUL %.4g ADD %.4g wave instructions = %.2f GFLOP = %.1f kflop per QUAD4 (4.0 M)" % (
        len(rows), r["SQ_INSTS_VALU_FMA_F64"], r["SQ_INSTS_VALU_MUL_F64"], r["SQ_INSTS_VALU_ADD_F64"], fl / 1e9, fl / 4.0e6 / 1e3))
PY
rm -rf $O
cat gpurun_out/r05_quad_flops.txt; grep -E "QUAD4|cg:" $O.log
