import importlib, sys, os
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("fem-shell_amd")
from tests.test_gpu_parity import delaunay_shell
xyz, tri = delaunay_shell(2500, 5, jittered=True)
n = len(xyz)
rng = np.random.default_rng(3)
fixed = np.flatnonzero(xyz[:, 0] < 0.2).astype(np.int32)
loaded = rng.choice(n, 40, replace=False).astype(np.int32)
f6 = rng.normal(size=(40, 6))
def run(flags, pc, env):
    for k, v in env.items():
        os.environ[k] = v
    try:
        fs = pkg.FemShell(0.3, 7.0e4, 0.03, flags=flags)
        fs.set_mesh(xyz, tri)
        fs.set_dirichlet(np.full(len(fixed), 0x3F, np.uint8), node_ids=fixed)
        fs.set_loads(f6, node_ids=loaded)
        fs.set_preconditioner(pc)
        u, info = fs.solve(rtol=1e-12, max_it=2000)
        print(hex(flags), pc, env, info["iterations"], info["converged"], info["amg_levels"], flush=True)
        fs.close()
    except Exception as e:
        print(hex(flags), pc, env, "FAILED", e, flush=True)
    for k in env:
        del os.environ[k]
for fl in (pkg.REORDER_MORTON, pkg.REORDER_RCM):
    flags = pkg.REF_DEFAULT | fl
    run(flags, "jacobi", {})
    run(flags, "amg", {"FEMSHELL_AMG_SETUP": "host"})
    run(flags, "amg", {"FEMSHELL_AMG_GALERKIN": "valu"})
    run(flags, "amg", {"FEMSHELL_AMG_VERBOSE": "1"})
