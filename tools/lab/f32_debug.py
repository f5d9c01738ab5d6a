import importlib, os, sys
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import meshes
pkg = importlib.import_module("fem-shell_amd")
m = meshes.scordelis_lo(48)
res = {}
for mode in ("0", "3", "2", "1"):
    os.environ["FEMSHELL_AMG_SMOOTH_F32"] = mode
    fs = pkg.FemShell(*m.material)
    fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads); fs.assemble()
    fs.set_preconditioner("amg", coarsest_nodes=60)
    u, info = fs.solve(rtol=1e-12, max_it=500)
    res[mode] = u
    print(mode, info["iterations"], info["amg_levels"], fs.residual_history()[:4], "diff vs 0: %.3e" % np.abs(u - res["0"]).max())
    fs.close()
