"""Block-Jacobi CG alone on a BASELINE-size system, run to convergence (or the iteration limit): the data point the
multigrid preconditioner is measured against.  usage: jacobi_full_solve.py panel|cylinder|roof N [rtol] [max_it]"""
import importlib
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from tests.helpers import meshes  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
kind, n = sys.argv[1], int(sys.argv[2])
rtol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-10
max_it = int(sys.argv[4]) if len(sys.argv) > 4 else 2000000
if kind == "panel":
    m = meshes.structured(n, n, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
    mat = (0.3, 1e7, 0.5)
elif kind == "roof":
    m = meshes.scordelis_lo(n)
    mat = m.material
else:
    m = meshes.pinched_cylinder(n, n)
    mat = m.material
fs = pkg.FemShell(*mat, device=0)
fs.set_mesh(m.xyz, m.tri)
fs.set_dirichlet(m.dirichlet_mask())
fs.set_loads(m.loads)
t0 = time.time()
u, info = fs.solve(rtol=rtol, max_it=max_it, fetch=False)
h = fs.residual_history(1 << 22)
out = {"mesh": "%s %d (%d tri3)" % (kind, n, len(m.tri)), "rtol": rtol, "iterations": info["iterations"], "converged": info["converged"],
       "solve_seconds": info["solve_seconds"], "wall_seconds": time.time() - t0, "rel_residual": info["rel_residual"],
       "true_rel_residual": info["true_rel_residual"],
       "history_every_50000": [float(v) for v in h[::50000]], "max_rel_residual_seen": float(h.max()) if len(h) else None}
print(json.dumps(out))
