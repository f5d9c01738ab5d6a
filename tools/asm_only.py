import importlib, sys, os
sys.path.insert(0, '.')
from bench import panel_mesh
pkg = importlib.import_module("fem-shell_amd")
nx = int(os.environ.get("NX", "1414"))
m = panel_mesh(nx)
fs = pkg.FemShell(0.3, 1e7, 0.5)
fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
for _ in range(int(os.environ.get("ASM_WARMUP", "3"))):
    fs.assemble()
ms, by = fs.time_kernel(pkg.KERNEL_ASSEMBLE, int(os.environ.get("ASM_REPS", "5")))
print("ASM: %.3f ms  %.1f GB/s  %.1f Melem/s" % (ms, by/ms/1e6, len(m.tri)/ms/1e3))
ms_s, by_s = fs.time_kernel(pkg.KERNEL_SPMV, 20)
print("SPMV: %.4f ms %.0f GB/s" % (ms_s, by_s / ms_s / 1e6))
if os.environ.get("CG", "1") == "1":
    _, info = fs.solve(rtol=0.0, max_it=300, fetch=False)
    print("CG: %.4f ms/iter" % (1e3 * info["solve_seconds"] / info["iterations"]))
