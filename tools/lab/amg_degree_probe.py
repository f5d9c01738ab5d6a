"""Smoother degrees 2/4 (default) against 3/4 and eig_ratio 20 / 30 on the three BASELINE meshes (lab probe)."""
import importlib, sys
sys.path.insert(0, ".")
from tests.helpers import fullsize, meshes
pkg = importlib.import_module("fem-shell_amd")
for kind, n in (("panel", 1414), ("cylinder", 1414), ("roof", 354)):
    if kind == "roof":
        m = meshes.scordelis_lo(n); mat = m.material
    else:
        m, mat = fullsize.workload(kind, n)
    fs = pkg.FemShell(*mat)
    fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads); fs.assemble()
    for sd, cd, ratio in ((2, 4, 30.0), (3, 4, 30.0), (3, 4, 20.0), (2, 4, 20.0), (3, 3, 30.0)):
        fs.set_preconditioner("amg", smoother_degree=sd, coarse_degree=cd, eig_ratio=ratio)
        fs.solve(rtol=1e-10, max_it=1500, fetch=False)
        u, info = fs.solve(rtol=1e-10, max_it=1500, fetch=False)
        print("%-8s fine %d coarse %d ratio %g: %4d its %.3f s, estimate %.1e" % (kind, sd, cd, ratio, info["iterations"], info["solve_seconds"], info["error_estimate"]), flush=True)
    fs.close()
