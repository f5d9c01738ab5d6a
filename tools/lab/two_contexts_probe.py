"""Three 4M-triangle contexts in one process, one after the other (as bench.py has them): setup laps of each.
python tools/lab/two_contexts_probe.py [n=1414]"""
import importlib
import sys
import time

sys.path.insert(0, ".")
from tests.helpers import meshes  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1414
keep = []
for kind in ("panel", "cylinder", "panel"):
    if kind == "panel":
        m = meshes.structured(n, n, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
        mat = (0.3, 1e7, 0.5)
    else:
        m = meshes.pinched_cylinder(n, n)
        mat = m.material
    fs = pkg.FemShell(*mat, device=0)
    fs.set_mesh(m.xyz, m.tri, m.quad)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    fs.assemble()
    fs.set_preconditioner("amg")
    print("==== %s" % kind, file=sys.stderr, flush=True)
    t0 = time.time()
    u, info = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
    print("%s: %d iterations, setup %.3f s, solve %.3f s, wall %.3f s, dense inverse %.2f ms" % (
        kind, info["iterations"], info["pc_setup_seconds"], info["solve_seconds"], time.time() - t0, fs.amg_dense_stats()["ms"]), flush=True)
    keep.append(fs)
