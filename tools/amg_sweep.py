"""Sweep of the multigrid options on one mesh: iterations and solve time per combination.
usage: amg_sweep.py panel|roof|cylinder N"""
import importlib
import itertools
import sys

sys.path.insert(0, ".")
from tests.helpers import meshes  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
kind, n = sys.argv[1], int(sys.argv[2])
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
fs.set_mesh(m.xyz, m.tri, m.quad)
fs.set_dirichlet(m.dirichlet_mask())
fs.set_loads(m.loads)
fs.assemble()
print("%s %d: %d tri" % (kind, n, len(m.tri)), flush=True)
for cyc, sd, cd, ratio in itertools.product(("K", "V"), (1, 2, 3), (2, 3, 4, 6), (30.0,)):
    fs.set_preconditioner("amg", cycle=cyc, smoother_degree=sd, coarse_degree=cd, eig_ratio=ratio, refine_passes=1)
    try:
        u, info = fs.solve(rtol=1e-10, max_it=1500, fetch=False)
        print("cycle %s fine %d coarse %d ratio %g: %4d its %.3f s (%.2f ms/it) conv %d setup %.2f s" % (
            cyc, sd, cd, ratio, info["iterations"], info["solve_seconds"], 1e3 * info["solve_seconds"] / max(1, info["iterations"]),
            info["converged"], info["pc_setup_seconds"]), flush=True)
    except pkg.FemShellError as e:
        print("cycle %s fine %d coarse %d ratio %g: %s" % (cyc, sd, cd, ratio, e), flush=True)
for ratio in (10.0, 20.0, 50.0, 100.0):
    fs.set_preconditioner("amg", eig_ratio=ratio, refine_passes=1)
    u, info = fs.solve(rtol=1e-10, max_it=1500, fetch=False)
    print("default K 2/4 ratio %g: %4d its %.3f s" % (ratio, info["iterations"], info["solve_seconds"]), flush=True)
