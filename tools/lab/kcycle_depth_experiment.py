"""K cycle on every coarse level but the last one before the coarsest (where the cycle is a two-grid method with an exact
coarse solve already), in the numpy restatement.  python tools/lab/kcycle_depth_experiment.py panel|roof|cyl NX [coarsest]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from tests.helpers import oracle, meshes
import amg_oracle as ao

SKIP = {"n": 2}
VISITS = {}
def cycle(levels, li, b, kcycle):
    L = levels[li]
    VISITS[li] = VISITS.get(li, 0) + 1
    if li == len(levels) - 1:
        return L.dense_inv @ b
    x = ao.smooth(L, b, None)
    bc = L.R @ (b - L.A @ x)
    if kcycle and li + SKIP["n"] < len(levels):
        xc = ao.kcycle_solve(levels, li + 1, bc)
    else:
        xc = cycle(levels, li + 1, bc, kcycle)
    x = x + L.P @ xc
    return ao.smooth(L, b, x)
ao.cycle = cycle
which, NX = sys.argv[1], int(sys.argv[2])
cn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if which == "panel":
    m = meshes.structured(NX, NX, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2); mat = (0.3, 1e7, 0.5)
elif which == "roof":
    m = meshes.scordelis_lo(NX); mat = m.material
else:
    m = meshes.pinched_cylinder(NX, NX); mat = m.material
r, c, v, F = oracle.assemble(m.xyz, m.tri, m.quad, oracle.material(*mat), dirichlet=m.dirichlet_mask(), loads=m.loads)
A = oracle.to_scipy(r, c, v).tobsr((6, 6))
levels = ao.setup(A, m.xyz, m.dirichlet_mask(), coarsest_nodes=cn, tri=m.tri)
print(which, NX, "levels", [L.n for L in levels], flush=True)
for skip in (2, 3):
    SKIP["n"] = skip
    VISITS.clear()
    u, hist = ao.solve(A, F.ravel(), levels, kcycle=True, rtol=1e-10, max_it=600, refine_passes=0)
    print("K cycle where li + %d < levels: %d iterations, visits per iteration %s" % (skip, len(hist), [round(VISITS.get(l, 0) / len(hist), 1) for l in range(len(levels))]), flush=True)
