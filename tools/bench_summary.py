import sys, json
d = json.loads(sys.stdin.read())
print("elements/s %.3g  ms/step %.3f  cg it/s %.1f  (cg ms %.3f)" % (d["value"], d["ms_per_step"], d["cg_iters_per_s"], d["cg_ms_per_iter"]))
for k in ("roofline", "roofline_cg_spmv", "roofline_cg_update", "roofline_cg_iteration"):
    print(k, "%.3f ms  frac %.3f  bytes %.3g" % (d[k]["ms_per_launch"], d[k]["frac"], d[k]["algorithmic_bytes_per_launch"]))
if "time_to_solution" in d:
    t = d["time_to_solution"]; print("amg", t["iterations"], t["solve_seconds"], t["pc_setup_seconds"])
if "parity" in d: print(d["parity"])
