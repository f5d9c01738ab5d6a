"""Time to solution of the multigrid-preconditioned solve on a BASELINE-size mesh, three solves on one hierarchy (for A/B runs
of two library builds, tools/lab/ab_lib.sh).  usage: solve_time_probe.py panel|cylinder|roof N"""
import importlib
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
fs.set_preconditioner("amg")
for rep in range(3):
    u, info = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
    print("   %s %d: %d iterations, solve %.4f s, setup %.3f s" % (kind, n, info["iterations"], info["solve_seconds"], info["pc_setup_seconds"]), flush=True)
fs.close()
