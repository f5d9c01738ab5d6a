"""Round 6: smoother degrees of the fine / coarse levels and the eigenvalue ratio once more on the final tree (default 3 / 4 / 30):
iterations and solve seconds of the second solve (hierarchy reused) on the three BASELINE meshes."""
import importlib, sys
sys.path.insert(0, ".")
from tests.helpers import fullsize, meshes
pkg = importlib.import_module("fem-shell_amd")
SWEEP = ((3, 4, 30.0), (2, 4, 30.0), (2, 3, 30.0), (2, 2, 30.0), (1, 4, 30.0), (2, 5, 30.0), (3, 3, 30.0), (3, 5, 30.0), (4, 4, 30.0), (4, 5, 30.0),
         (3, 4, 20.0), (3, 4, 40.0), (4, 4, 40.0))  # (fine degree, coarse degree, lambda_max / lambda_min of the smoother); default first
for kind, n in (("panel", 1414), ("cylinder", 1414), ("roof", 354)):
    if kind == "roof":
        m = meshes.scordelis_lo(n); mat = m.material
    else:
        m, mat = fullsize.workload(kind, n)
    fs = pkg.FemShell(*mat)
    fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads); fs.assemble()
    for sd, cd, ratio in SWEEP:
        fs.set_preconditioner("amg", smoother_degree=sd, coarse_degree=cd, eig_ratio=ratio)
        fs.solve(rtol=1e-10, max_it=1500, fetch=False)
        u, info = fs.solve(rtol=1e-10, max_it=1500, fetch=False)
        print("%-8s fine %d coarse %d ratio %g: %4d its %.3f s (%.2f ms per iteration), estimate %.1e %s" % (
            kind, sd, cd, ratio, info["iterations"], info["solve_seconds"], 1e3 * info["solve_seconds"] / max(info["iterations"], 1),
            info["error_estimate"], "" if info["converged"] == 1 else "NOT CONVERGED"), flush=True)
    fs.close()
