"""Iteration counts of the multigrid solve against where the hierarchy ends: levels_probe.py panel|cylinder N coarsest_nodes [...]"""
import importlib, json, sys
sys.path.insert(0, ".")
from tests.helpers import meshes  # noqa: E402
pkg = importlib.import_module("fem-shell_amd")
kind, n = sys.argv[1], int(sys.argv[2])
if kind == "panel":
    m = meshes.structured(n, n, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2); mat = (0.3, 1e7, 0.5)
else:
    m = meshes.pinched_cylinder(n, n); mat = m.material
fs = pkg.FemShell(*mat, device=0)
fs.set_mesh(m.xyz, m.tri, m.quad); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads); fs.assemble()
for cn in [int(a) for a in sys.argv[3:]]:
    fs.set_preconditioner("amg", coarsest_nodes=cn)
    u, info = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
    print("%s %d coarsest_nodes %d: levels %s iterations %d solve %.3f s setup %.3f s" % (kind, n, cn, [l["n_nodes"] for l in fs.amg_levels()],
          info["iterations"], info["solve_seconds"], info["pc_setup_seconds"]), flush=True)
