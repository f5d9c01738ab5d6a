"""||K||_F and trace of the library's K on the mesh tools/lab/scatter_lab builds (no Dirichlet dofs), to hold the lab's two
scatter kernels to:  python tools/lab/scatter_check.py NX"""
import importlib
import sys

import numpy as np

sys.path.insert(0, ".")
pkg = importlib.import_module("fem-shell_amd")
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n1 = nx + 1
j, i = np.meshgrid(np.arange(n1), np.arange(n1), indexing="ij")
xyz = np.stack([10.0 * i.ravel() / nx, 10.0 * j.ravel() / nx, np.zeros(n1 * n1)], axis=1)
jj, ii = np.meshgrid(np.arange(nx), np.arange(nx), indexing="ij")
a = (jj * n1 + ii).ravel()
b, c, d = a + 1, a + n1, a + n1 + 1
tri = np.empty((2 * nx * nx, 3), dtype=np.int32)
tri[0::2] = np.stack([a, b, d], axis=1)
tri[1::2] = np.stack([a, d, c], axis=1)
fs = pkg.FemShell(0.3, 1e7, 0.5, device=0)
fs.set_mesh(xyz, tri, None)
fs.set_dirichlet(np.zeros(n1 * n1, dtype=np.uint8))
fs.assemble()
rp, ci, vals, F = fs.export_bsr()
vals = np.asarray(vals).reshape(-1, 6, 6)
rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
diag = vals[rows == np.asarray(ci)]
print("library: %d blocks, ||K||_F = %.15e  trace = %.15e" % (len(ci), np.sqrt((vals.astype(np.longdouble) ** 2).sum()), np.trace(diag, axis1=1, axis2=2).sum()))
