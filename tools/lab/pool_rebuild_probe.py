"""Does the device pool settle when a hierarchy is rebuilt again and again?  4M-triangle panel, the multigrid setup repeated after a
change of the constraint set; HBM in use (torch.cuda.mem_get_info) and the setup's seconds after each.  (round 6: carved blocks,
deferred releases -- csrc/context.hpp DevPool)"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
from tests.helpers import fullsize  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1414
m, mat = fullsize.workload("panel", n)
fs = pkg.FemShell(*mat, device=0)
fs.set_mesh(m.xyz, m.tri)
fs.set_loads(m.loads)
dm0 = m.dirichlet_mask()
for k in range(8):
    dm = dm0.copy()
    dm[m.n_nodes // 2 + k] = 0x3F
    fs.set_dirichlet(dm)
    fs.set_preconditioner("amg")
    u, info = fs.solve(rtol=1e-10, max_it=400, fetch=False)
    free, total = torch.cuda.mem_get_info(0)
    print("rebuild %d: setup %.3f s, %d iterations, solve %.3f s, HBM in use %.2f GB" % (k, info["pc_setup_seconds"], info["iterations"], info["solve_seconds"], (total - free) / 1e9), flush=True)
fs.close()
free, total = torch.cuda.mem_get_info(0)
print("after close: HBM in use %.2f GB" % ((total - free) / 1e9))
