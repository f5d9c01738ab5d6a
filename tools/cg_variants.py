"""ms per CG iteration of the classic and the single-reduction recurrence on one GPU, for a few mesh sizes
(profiling aid): python tools/cg_variants.py"""
import importlib, os, sys
sys.path.insert(0, '.')
from bench import panel_mesh
pkg = importlib.import_module("fem-shell_amd")
for nx in (64, 354, 707, 1414):
    m = panel_mesh(nx)
    fs = pkg.FemShell(0.3, 1e7, 0.5)
    fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
    fs.assemble()
    row = ["%dx%d squares, %d tri3" % (nx, nx, len(m.tri))]
    for var in ("0", "1"):
        os.environ["FEMSHELL_CG_SINGLE_REDUCTION"] = var
        fs.solve(rtol=0.0, max_it=50, fetch=False)
        _, info = fs.solve(rtol=0.0, max_it=400, fetch=False)
        row.append("%s %.4f ms/iter" % ("single-reduction" if var == "1" else "classic", 1e3 * info["solve_seconds"] / info["iterations"]))
    print(" | ".join(row))
    fs.close()
