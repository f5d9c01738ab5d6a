"""The random-point Delaunay shell with hull slivers (element quality down to 1e-4) under stronger smoothing:
python tools/amg_sliver_probe.py"""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_gpu_parity import delaunay_shell
pkg = importlib.import_module("fem-shell_amd")
xyz, tri = delaunay_shell(20000, 3)
n = len(xyz)
dmask = np.zeros(n, dtype=np.uint8)
dmask[xyz[:, 0] < 0.15] = 0x3F
loads = np.zeros((n, 6))
loads[:, 2] = 1.0
for opts in ({}, {"smoother_degree": 4, "coarse_degree": 4}, {"smoother_degree": 6, "coarse_degree": 6}, {"smoother_degree": 4, "coarse_degree": 4, "eig_ratio": 100.0},
             {"smoother_degree": 8, "coarse_degree": 8, "eig_ratio": 100.0}):
    fs = pkg.FemShell(0.3, 7.0e4, 0.03)
    fs.set_mesh(xyz, tri)
    fs.set_dirichlet(dmask)
    fs.set_loads(loads)
    fs.set_preconditioner("amg", **opts)
    try:
        u, info = fs.solve(rtol=1e-10, max_it=3000)
        print(opts, "its", info["iterations"], "conv", info["converged"], "%.2f s" % info["solve_seconds"], flush=True)
    except pkg.FemShellError as ex:
        print(opts, "ERROR", ex, flush=True)
    fs.close()
