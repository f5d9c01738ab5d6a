"""One-screen summary of a bench run: the compact stdout line on stdin, the detail record beside it if present.
usage: python3 bench.py ... | python3 tools/bench_summary.py [bench_detail.json]"""
import json
import os
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("elements/s %.4g  ms/step %.4f  kernel %s %.4f ms  frac %.3f  cg it/s %.1f  tts %s s / %s it  parity_max_rel %s" % (
    d["value"], d["ms_per_step"], r.get("kernel"), r.get("ms_per_launch", float("nan")), r.get("frac", float("nan")), d["cg_iters_per_s"],
    d.get("time_to_solution_s"), d.get("time_to_solution_iterations"), d.get("parity_max_rel")))
path = sys.argv[1] if len(sys.argv) > 1 else d.get("detail_file", "bench_detail.json")
if os.path.exists(path):
    dd = json.load(open(path))
    for k in ("roofline_cg_spmv", "roofline_cg_update", "roofline_cg_iteration"):
        if k in dd:
            print(k, "%.4f ms  frac %.3f  bytes %.4g" % (dd[k]["ms_per_launch"], dd[k]["frac"], dd[k]["algorithmic_bytes_per_launch"]))
    t = dd.get("time_to_solution")
    if t:
        print("multigrid: %d iterations %.4f s, setup %.3f s, %.2f GB/iteration = %.3f of HBM" % (
            t["iterations"], t["solve_seconds"], t["pc_setup_seconds"], t["algorithmic_gb_per_iteration"], t["roofline_amg_iteration"]["frac"]))
