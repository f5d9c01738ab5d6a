"""A/B on one box, one process: the six words below the diagonal of the diagonal blocks written (FEMSHELL_DIAG_UPPER=0) or not."""
import importlib, os, sys
sys.path.insert(0, '.')
from bench import panel_mesh
pkg = importlib.import_module("fem-shell_amd")
m = panel_mesh(int(sys.argv[1]) if len(sys.argv) > 1 else 1414)
ctx = {}
for flag in ("1", "0"):
    os.environ["FEMSHELL_DIAG_UPPER"] = flag
    fs = pkg.FemShell(0.3, 1e7, 0.5)
    fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
    fs.assemble()
    ctx[flag] = fs
for rep in range(5):
    for flag in ("1", "0"):
        ms, by = ctx[flag].time_kernel(pkg.KERNEL_ASSEMBLE, 20)
        sp, _ = ctx[flag].time_kernel(pkg.KERNEL_SPMV, 20)
        print("round %d diag_upper=%s: assembly %.4f ms (%.3f GB algorithmic), spmv %.4f ms" % (rep, flag, ms, by / 1e9, sp), flush=True)
