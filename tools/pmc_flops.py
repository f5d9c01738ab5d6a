#!/usr/bin/env python3
"""FP64 work per kernel launch from three rocprofv3 counter passes (SQ_INSTS_VALU_FMA_F64 / MUL_F64 / ADD_F64; wave-level
instruction counts, 64 lanes each, an FMA = 2 flops):  pmc_flops.py <fma.csv> <mul.csv> <add.csv> <out.json>"""
import csv
import json
import statistics
import sys


def per_kernel(path):
    rows = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            name = name[:name.index("(")] if "(" in name else name
            if name.startswith("void "):
                name = name[5:]
            rows.setdefault(name, {}).setdefault(r["Dispatch_Id"], 0.0)
            rows[name][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: statistics.median(v.values()) for k, v in rows.items()}


def main():
    fma, mul, add = (per_kernel(p) for p in sys.argv[1:4])
    out = {}
    for k in fma:
        if not k.startswith("femshell::"):
            continue
        out[k] = {"wave_insts_fma_f64": fma[k], "wave_insts_mul_f64": mul.get(k, 0.0), "wave_insts_add_f64": add.get(k, 0.0),
                  "flops_per_launch": 64.0 * (2.0 * fma[k] + mul.get(k, 0.0) + add.get(k, 0.0))}
        print("%-50s %.3f GFLOP per launch" % (k, out[k]["flops_per_launch"] / 1e9))
    with open(sys.argv[4], "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
