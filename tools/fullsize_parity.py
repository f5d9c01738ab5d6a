"""Converged full-size solve of BASELINE configs[1] (Scordelis-Lo roof, 354x354 squares = 250,632 tri3) on the GPU
against the oracle's refined direct solve: solver term (same matrix) and total (oracle-assembled matrix).
usage: fullsize_parity.py [n=354] [jacobi]"""
import importlib
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from tests.helpers import meshes, oracle  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 354
with_jacobi = len(sys.argv) > 2
m = meshes.scordelis_lo(n)
fs = pkg.FemShell(*m.material, device=0)
fs.set_mesh(m.xyz, m.tri, m.quad)
fs.set_dirichlet(m.dirichlet_mask())
fs.set_loads(m.loads)
fs.set_preconditioner("amg")
out = {"mesh": "Scordelis-Lo %dx%d (%d tri3)" % (n, n, len(m.tri))}
for rtol in (1e-10, 1e-12, 1e-14):
    t0 = time.time()
    u, info = fs.solve(rtol=rtol, max_it=2000)
    out["amg_rtol_%g" % rtol] = {"iterations": info["iterations"], "solve_s": info["solve_seconds"], "wall_s": time.time() - t0,
                                 "true_rel_residual": info["true_rel_residual"], "u": u.copy()}
rg, cg, vg, Fg = fs.export_bsr()
t0 = time.time()
ug, hist = oracle.refined_solve(rg, cg, vg, Fg, sweeps=5, return_history=True)
out["direct_s"] = time.time() - t0
out["refinement_residuals"] = hist
for k in list(out):
    if k.startswith("amg_rtol"):
        u = out[k].pop("u")
        out[k]["rel_err_vs_direct_same_matrix"] = float(np.linalg.norm(u.ravel() - ug) / np.linalg.norm(ug))
        out[k]["max_err_over_max_u"] = float(np.abs(u.ravel() - ug).max() / np.abs(ug).max())
if with_jacobi:
    fs.set_preconditioner("jacobi")
    t0 = time.time()
    u, info = fs.solve(rtol=1e-12, max_it=2000000)
    out["jacobi_rtol_1e-12"] = {"iterations": info["iterations"], "solve_s": info["solve_seconds"],
                                "true_rel_residual": info["true_rel_residual"],
                                "rel_err_vs_direct_same_matrix": float(np.linalg.norm(u.ravel() - ug) / np.linalg.norm(ug))}
mat = oracle.material(*m.material)
t0 = time.time()
r0, c0, v0, F0 = oracle.assemble(m.xyz, m.tri, m.quad, mat, m.dirichlet_mask(), m.loads)
out["matrix_rel_diff"] = float(np.abs(vg - v0).max() / np.abs(v0).max())
u0 = oracle.refined_solve(r0, c0, v0, F0, sweeps=5)
out["oracle_total_s"] = time.time() - t0
out["sensitivity_two_assemblies"] = float(np.linalg.norm(ug - u0) / np.linalg.norm(u0))
print(json.dumps(out, indent=1))
