import importlib, sys, json
sys.path.insert(0, ".")
from tests.helpers import fullsize, meshes
pkg = importlib.import_module("fem-shell_amd")
for kind, n in (("panel", 1414), ("cylinder", 1414), ("roof", 354)):
    if kind == "roof":
        m = meshes.scordelis_lo(n); mat = m.material
    else:
        m, mat = fullsize.workload(kind, n)
    fs = pkg.FemShell(*mat, device=0)
    fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
    for sd, cd in ((2, 4), (2, 3), (2, 5), (3, 3), (1, 4)):
        fs.set_preconditioner("amg", smoother_degree=sd, coarse_degree=cd)
        fs.solve(rtol=1e-10, max_it=3000, fetch=False)
        _, info = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
        print(kind, "degrees", sd, cd, "iterations", info["iterations"], "solve %.3f s" % info["solve_seconds"], flush=True)
    fs.close()
