"""The flat panel beyond the north-star size on one GPU: symbolic phase, assembly rate, memory, multigrid solve.
usage: big_panel_probe.py NX   (NX = 4000: 32,000,000 triangles, 96 M dofs)"""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
from tests.helpers import meshes  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
m = meshes.structured(n, n, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
fs = pkg.FemShell(0.3, 1e7, 0.5, device=0)
t0 = time.time()
fs.set_mesh(m.xyz, m.tri, None)
t1 = time.time()
fs.set_dirichlet(m.dirichlet_mask())
fs.set_loads(m.loads)
for _ in range(5):
    fs.assemble()
ms, nbytes = fs.time_kernel(pkg.KERNEL_ASSEMBLE, 10)
print("panel %d x %d: %d triangles, %d dofs; set_mesh %.2f s; assembly %.3f ms = %.2f G elements/s, %.2f GB algorithmic = %.0f GB/s" % (
    n, n, len(m.tri), 6 * m.n_nodes, t1 - t0, ms, len(m.tri) / ms / 1e6, nbytes / 1e9, nbytes / ms / 1e6), flush=True)
s_ms, s_bytes = fs.time_kernel(pkg.KERNEL_SPMV, 10)
print("symmetric SpMV %.3f ms = %.0f GB/s" % (s_ms, s_bytes / s_ms / 1e6), flush=True)
fs.set_preconditioner("amg")
u, info = fs.solve(rtol=1e-10, max_it=3000, fetch=False)
free, total = torch.cuda.mem_get_info(0)
print("multigrid: %d levels, setup %.2f s, %d iterations, solve %.2f s, error estimate %.1e; HBM in use %.1f of %.0f GB" % (
    info["amg_levels"], info["pc_setup_seconds"], info["iterations"], info["solve_seconds"], info["error_estimate"], (total - free) / 1e9, total / 1e9), flush=True)
fs.close()
