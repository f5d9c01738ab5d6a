#!/usr/bin/env python3
"""Matrix-core counters of the dense inverse of the coarsest operator (csrc/amg_dense.hip), from rocprofv3 counter passes over
tools/amg_probe.py:   pmc_mfma.py <busy_pass.csv> <mops_pass.csv> <kernel_trace.csv> <out.json> [<probe.txt of the trace pass>]
The counter passes run with FEMSHELL_AMG_DENSE_LOOKAHEAD=0 (rocprofv3 serialises launches while it collects counters; the look-ahead's
launch on the second stream would wait for a launch that cannot start): same kernels, same flops, kernel names without the
template arguments.  Time base of the whole inverse: the wall time the library measured in the trace pass (probe.txt; the pivot
launches overlap the updates there, so kernel durations do not add up to it); per kernel: durations of the trace pass.
  pass 1: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE      pass 2: --pmc SQ_INSTS_VALU_MFMA_MOPS_F64
Derived, per kernel and over all kernels of the inverse:
  flops_by_counter      = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 (the counter's unit)  -- against the flops the host counts as issued
  tflops_by_counter     = flops_by_counter / kernel time of the UNPROFILED trace pass
  mfma_busy_fraction    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of SIMD-cycles a matrix instruction
                          was executing (GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3, MI355X_MICROARCH.md)"""
import csv
import json
import re
import sys
from collections import defaultdict


def clean(name):
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "").replace("femshell::", "")
    return re.sub(r"<.*", "", name) if name.startswith("k_dense_") and not name.startswith("k_dense_gemv") else name


def counters(path):
    per = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = clean(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
    return per, {k: len(v) for k, v in launches.items()}


def durations(path):
    per = defaultdict(float)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            per[clean(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return per


def main():
    busy, n1 = counters(sys.argv[1])
    mops, n2 = counters(sys.argv[2])
    dur = durations(sys.argv[3])
    out = {"kernels": {}, "peak_fp64_matrix_tflops": 78.6}
    tot = defaultdict(float)
    for k in sorted(busy):
        if not k.startswith("k_dense_") or k.startswith("k_dense_gemv"):  # (the products with the inverse run in the solve, not here)
            continue
        b, m = busy[k], mops.get(k, {})
        flops = m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0
        gui = b.get("GRBM_GUI_ACTIVE", 0.0)
        e = {"launches": n1[k], "seconds_unprofiled_trace": dur.get(k, 0.0),
             "SQ_VALU_MFMA_BUSY_CYCLES": b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), "SQ_BUSY_CYCLES": b.get("SQ_BUSY_CYCLES", 0.0),
             "GRBM_GUI_ACTIVE": gui, "SQ_INSTS_VALU_MFMA_MOPS_F64": m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0),
             "flops_by_counter": flops,
             "tflops_by_counter": flops / dur[k] / 1e12 if dur.get(k) else None,
             "mfma_busy_fraction": b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui / 8.0) if gui else None}
        out["kernels"][k] = e
        for key in ("seconds_unprofiled_trace", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "flops_by_counter"):
            tot[key] += e[key]
    wall = None
    if len(sys.argv) > 5:
        m = re.search(r"dense inverse of the coarsest operator on the matrix cores: n = \d+, ([0-9.]+) ms", open(sys.argv[5]).read())
        wall = float(m.group(1)) * 1e-3 if m else None
    if wall:  # (one inverse per probe run)
        tot["seconds_unprofiled_trace"] = wall
    out["all_kernels_of_the_inverse"] = {
        "time_base": "wall time of the inverse as the library measured it in the trace pass" if wall else "sum of kernel durations",
        "seconds_unprofiled_trace": tot["seconds_unprofiled_trace"], "flops_by_counter": tot["flops_by_counter"],
        "tflops_by_counter": tot["flops_by_counter"] / tot["seconds_unprofiled_trace"] / 1e12 if tot["seconds_unprofiled_trace"] else None,
        "frac_of_78.6_tflops_by_counter": tot["flops_by_counter"] / tot["seconds_unprofiled_trace"] / 1e12 / 78.6 if tot["seconds_unprofiled_trace"] else None,
        "mfma_busy_fraction": tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * tot["GRBM_GUI_ACTIVE"] / 8.0) if tot["GRBM_GUI_ACTIVE"] else None}
    with open(sys.argv[4], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["all_kernels_of_the_inverse"]))
    for k, e in out["kernels"].items():
        print("%-28s %4d launches  %.3f ms  %.1f TFLOP/s by counter  MFMA busy %.3f" % (k, e["launches"], 1e3 * e["seconds_unprofiled_trace"],
              e["tflops_by_counter"] or 0.0, e["mfma_busy_fraction"] or 0.0))


if __name__ == "__main__":
    main()
