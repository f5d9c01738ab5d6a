"""Round 6: a hub of high valence -- which coarsening steps build their patterns in HBM, which take the host's lists (rows beyond the
lane sets of csrc/amg_symbolic.hip), where the 255-per-row limit of both paths ends it.  usage: fan_probe.py VALENCE ..."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, ".")
os.environ.setdefault("FEMSHELL_AMG_PATCH_TAU", "0")
os.environ.setdefault("FEMSHELL_AMG_DEVICE_MIN", "100")
from tests.test_gpu_amg import _fan_mesh  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
for v in [int(a) for a in sys.argv[1:]]:
    xyz, tri = _fan_mesh(v)
    n = len(xyz)
    dmask = np.zeros(n, dtype=np.uint8)
    dmask[np.hypot(xyz[:, 0], xyz[:, 1]) > 5.5] = 0x3F
    loads = np.zeros((n, 6))
    loads[:, 2] = 1.0
    fs = pkg.FemShell(0.3, 1e7, 0.2, device=0)
    fs.set_mesh(xyz, tri)
    fs.set_dirichlet(dmask)
    fs.set_loads(loads)
    fs.set_preconditioner("amg", coarsest_nodes=60)
    try:
        u, info = fs.solve(rtol=1e-10, max_it=2000)
        print(v, fs.amg_symbolic_info(), info["iterations"], info["converged"], [l["n_nodes"] for l in fs.amg_levels()])
    except Exception as e:  # noqa: BLE001
        print(v, "error:", e)
    fs.close()
