import importlib, sys
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("fem-shell_amd")
from tests.test_gpu_parity import delaunay_shell
from tests.helpers import oracle
from importlib import import_module
b = import_module("fem-shell_amd.binding")
xyz, tri = delaunay_shell(2500, 5)
n = len(xyz)
fixed = np.flatnonzero(xyz[:, 0] < 0.2)
for kind in ("morton", "rcm"):
    perm = pkg.reorder_host(kind, xyz, tri)
    iperm = np.empty_like(perm); iperm[perm] = np.arange(n, dtype=np.int32)
    X = xyz[perm]; T = iperm[tri].astype(np.int32)
    dm = np.zeros(n, np.uint8); dm[iperm[fixed]] = 0x3F
    r, c, v, F = oracle.assemble(X, T, np.zeros((0, 4), np.int32), oracle.material(0.3, 7e4, 0.03), dm, np.zeros((n, 6)))
    B = b.amg_host_rbm(X, dm)
    for lvl in range(2):
        out = b.amg_host_coarsen(r, c, v, B, 2.0)
        agg = out["agg"]; na = agg.max() + 1
        pc = out["P_cols"]; 
        rw = np.bincount(pc, minlength=na)
        pw = np.diff(out["P_rowptr"])
        aw = np.diff(out["Ac_rowptr"])
        print(kind, lvl, "n", len(r) - 1, "na", na, "agg size max", np.bincount(agg).max(), "P row max", pw.max(), "R row max", rw.max(), "Ac row max", aw.max())
        r, c, v, B = out["Ac_rowptr"].astype(np.int32), out["Ac_cols"], out["Ac_vals"], out["Bc"]
