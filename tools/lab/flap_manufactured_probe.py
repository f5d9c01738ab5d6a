"""Why does the rounding-correction solve of the manufactured flap break down?  python tools/lab/flap_manufactured_probe.py nx nz"""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import meshes, manufactured
pkg = importlib.import_module("fem-shell_amd")
nx, nz = int(sys.argv[1]), int(sys.argv[2])
m = meshes.structured(nx, nz, 0, 0, 0.1, 1.0, kind="t", ul_lr=True, bcids=(2, 20, 2, 2), factor=1.0, loading=0, dead_axis="y")
fs = pkg.FemShell(0.3, 1e6, 0.1, device=0)
fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(np.zeros((m.n_nodes, 6))); fs.assemble()
from tests.helpers import fullsize
for kind in ("flap",):
    out = fullsize.manufactured_solve(fs, m, kind, rtol=1e-10, passes=(1,))
    r = out["runs"][1]
    print(kind, "b norm %.3e u* norm %.3e" % (out["b_norm"], out["u_star_norm"]), out.get("rounding_of_b"),
          "iterations", r["iterations"], "converged", r["converged"], "err %.2e" % r["rel_err_vs_manufactured"], "estimate %.2e" % r["error_estimate"], flush=True)
