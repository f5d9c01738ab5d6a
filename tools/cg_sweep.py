import importlib, sys, os
sys.path.insert(0, '.')
from bench import panel_mesh
pkg = importlib.import_module("fem-shell_amd")
m = panel_mesh(int(os.environ.get("NX", "1414")))
fs = pkg.FemShell(0.3, 1e7, 0.5)
fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
fs.assemble()
out = ["grid=%s" % os.environ.get("FEMSHELL_SLICE_GRID")]
for k, name in [(pkg.KERNEL_SPMV, "spmv"), (pkg.KERNEL_CG_UPDATE, "update"), (pkg.KERNEL_CG_DIRECTION, "direction")]:
    ms, by = fs.time_kernel(k, 20)
    out.append("%s %.4f ms %.0f GB/s" % (name, ms, by/ms/1e6))
_, info = fs.solve(rtol=0.0, max_it=300, fetch=False)
out.append("cg %.4f ms/iter" % (1e3*info["solve_seconds"]/info["iterations"]))
print(" | ".join(out))
