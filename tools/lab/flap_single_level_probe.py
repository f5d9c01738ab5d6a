"""Why does the multigrid default break down on the 2,000-triangle coupled flap (one level: the dense inverse alone)?
python tools/lab/flap_single_level_probe.py [nx nz]"""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import meshes
pkg = importlib.import_module("fem-shell_amd")
nx, nz = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10, 100)
m = meshes.structured(nx, nz, 0, 0, 0.1, 1.0, kind="t", ul_lr=True, bcids=(2, 20, 2, 2), factor=1.0, loading=0, dead_axis="y")
loads = np.zeros((len(m.xyz), 6)); loads[np.abs(m.xyz[:, 0]) < 1e-12, 0] = 1.0
for dense_min, coarsest in ((250, 1400), (100000, 1400), (250, 300), (250, 100)):
    os.environ["FEMSHELL_AMG_DENSE_DEVICE_MIN"] = str(dense_min)
    fs = pkg.FemShell(0.3, 1e6, 0.1, device=0)
    fs.set_mesh(m.xyz, m.tri, m.quad); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(loads); fs.assemble()
    fs.set_preconditioner("amg", coarsest_nodes=coarsest)
    try:
        u, info = fs.solve(rtol=1e-12, max_it=500)
        print("dense_device_min %d coarsest %d:" % (dense_min, coarsest), json.dumps({k: info[k] for k in ("iterations", "converged", "rel_residual", "amg_levels", "error_estimate")}), fs.amg_dense_stats())
    except Exception as e:
        print("dense_device_min %d coarsest %d: FAILED %s" % (dense_min, coarsest, e), fs.residual_history()[:20])
    fs.close()
