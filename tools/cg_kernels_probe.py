"""Per-kernel times of the CG iteration on the 4M-triangle panel through femshell_time_kernel (HIP events on the library's
stream), several repetitions in one process:  python tools/cg_kernels_probe.py [nx]"""
import importlib, sys
sys.path.insert(0, ".")
from bench import panel_mesh
pkg = importlib.import_module("fem-shell_amd")
m = panel_mesh(int(sys.argv[1]) if len(sys.argv) > 1 else 1414)
fs = pkg.FemShell(0.3, 1e7, 0.5)
fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
for _ in range(20):
    fs.assemble()
fs.solve(rtol=0.0, max_it=200, fetch=False)
for rep in range(3):
    a, _ = fs.time_kernel(pkg.KERNEL_ASSEMBLE, 10)
    s, _ = fs.time_kernel(pkg.KERNEL_SPMV, 20)
    u, _ = fs.time_kernel(pkg.KERNEL_CG_UPDATE, 20)
    d, _ = fs.time_kernel(pkg.KERNEL_CG_DIRECTION, 20)
    _, info = fs.solve(rtol=0.0, max_it=400, fetch=False)
    print("assemble %.4f  spmv %.4f  update %.4f  direction %.4f  | CG %.4f ms/iter" % (a, s, u, d, 1e3 * info["solve_seconds"] / info["iterations"]), flush=True)
