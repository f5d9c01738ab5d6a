"""The dense inverse of a coarsest operator of about 7.4k dofs without the 4M-triangle problem around it (profiling aid):
python tools/lab/dense_probe.py [squares_per_side=105] [repeats=3]"""
import importlib
import sys

sys.path.insert(0, ".")
from tests.helpers import meshes  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 105
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = meshes.structured(nx, nx, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
for k in range(reps):
    fs = pkg.FemShell(0.3, 1e7, 0.5)
    fs.set_mesh(m.xyz, m.tri, m.quad)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    fs.set_preconditioner("amg")
    u, info = fs.solve(rtol=1e-10, max_it=300)
    st = fs.amg_dense_stats()
    lv = [l["n_nodes"] for l in fs.amg_levels()]
    print("levels %s  dense inverse: %d dofs  %.3f ms  %.1f TFLOP/s issued  dropped %d   solve: %d iterations conv %d" % (
        lv, st["n"], st["ms"], st["mfma_flops_issued"] / st["ms"] / 1e9 if st["ms"] > 0 else 0.0, st["dropped_directions"],
        info["iterations"], info["converged"]), flush=True)
    fs.close()
