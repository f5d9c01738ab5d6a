import importlib, sys
import numpy as np
sys.path.insert(0, '.')
from tests.helpers import fullsize
pkg = importlib.import_module("fem-shell_amd")
for kind in ("panel", "cylinder"):
    m, mat = fullsize.workload(kind, 1414)
    fs = pkg.FemShell(*mat)
    fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
    for passes in (1, 0):
        fs.set_preconditioner("amg", refine_passes=passes)
        u, info = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
        h = fs.residual_history()
        # where does the history cross 1e-8 / 1e-10 (first phase), and what does the pass start from
        first = int(np.argmax(h < (1e-8) ** 2)) if np.any(h < 1e-16) else -1
        print(kind, "passes", passes, {k: info[k] for k in ("iterations", "solve_seconds", "rel_residual", "true_rel_residual", "refine_passes_done", "refine_correction_rel", "refine_residual_reduction", "error_estimate")}, "history crosses 1e-8 at", first, flush=True)
    fs.close()
