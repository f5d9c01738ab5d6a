"""Multigrid solve of the coupled example's flap (0.1 x 1 in the x-z plane, bottom edge clamped, E=1e6 nu=0.3 t=0.1) at
growing sizes:  python tools/flap_amg_probe.py"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import meshes
pkg = importlib.import_module("fem-shell_amd")
sizes = [(10, 100), (50, 100), (100, 200), (250, 500), (500, 1000)]
if len(sys.argv) > 1:
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for nx, nz in sizes:
    m = meshes.structured(nx, nz, 0, 0, 0.1, 1.0, kind="t", ul_lr=True, bcids=(2, 20, 2, 2), factor=1.0, loading=0, dead_axis="y")
    loads = np.zeros((m.n_nodes, 6))
    left = np.where(np.abs(m.xyz[:, 0]) < 1e-12)[0]
    loads[left, 0] = 1.0
    fs = pkg.FemShell(0.3, 1e6, 0.1)
    fs.set_mesh(m.xyz, m.tri)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(loads)
    fs.set_preconditioner("amg")
    try:
        u, info = fs.solve(rtol=1e-10, max_it=3000)
        print(nx, nz, len(m.tri), "its", info["iterations"], "conv", info["converged"], "levels", info["amg_levels"],
              "solve %.3f s setup %.3f s" % (info["solve_seconds"], info["pc_setup_seconds"]), [l["n_nodes"] if isinstance(l, dict) and "n_nodes" in l else l for l in fs.amg_levels()][:8], flush=True)
    except pkg.FemShellError as ex:
        print(nx, nz, len(m.tri), "ERROR", ex, flush=True)
    fs.close()
